#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_r03a.log; tail -15 $OUT/gpu_tests_r03a.log
cp $OUT/parity_report.json $OUT/parity_report_r03a.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/bench_r03a.json 2> $OUT/bench_r03a.err; cut -c1-300 $OUT/bench_r03a.json; tail -2 $OUT/bench_r03a.err
python -c "
import json; d=json.load(open('$OUT/bench_r03a.json')); print(json.dumps(d['roofline_hbm'])); print(d['b256']); print(d['cpu_baseline']['value'], d['cpu_baseline']['best_value'], d['cpu_baseline']['best_threads'], d['cpu_baseline']['thread_sweep_train_images_per_s'])"
