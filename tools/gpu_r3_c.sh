#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python tools/debug_j2o.py > $OUT/r03_debug_j2o.txt 2>$OUT/r3c.err; cat $OUT/r03_debug_j2o.txt
python -m pytest tests/test_head_gpu.py -m gpu -q --tb=short 2>&1 | tail -25 > $OUT/r3c_head.log; tail -12 $OUT/r3c_head.log
python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -s -k "yardstick or failing_bucket or trainer_end" 2>&1 | grep -v "^E        +" | tail -40 > $OUT/r3c_nets.log; tail -25 $OUT/r3c_nets.log
python -m pytest tests/test_dp_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $OUT/r3c_dp.log; tail -6 $OUT/r3c_dp.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 2>>$OUT/r3c.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NHWC', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'], json.dumps(d['roofline_hbm']))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --coord-weight 1 2>>$OUT/r3c.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NHWC cw1', d['value'], d['ms_per_step'], json.dumps(d['roofline_hbm']))"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r3c -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 > $OUT/prof_r3c.log 2>&1 )
find $OUT/prof_r3c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_r3c.csv
rm -rf $OUT/prof_r3c
grep -E "nhwc|finish|zero_f64" $OUT/kernel_stats_r3c.csv | cut -c1-200
tail -3 $OUT/r3c.err
