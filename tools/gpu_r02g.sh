#!/bin/bash
TAG=r02g
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json
tools/gpu_tests.sh $TAG
python bench.py --steps 20 --warmup 5 --no-split-mode --no-cpu-baseline --per-layer $OUT/per_layer_$TAG.txt > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -3 $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']); print(d['forward']); print(d['config3'])"
python bench.py --steps 10 --warmup 3 --no-split-mode --no-extras --no-cpu-baseline --no-parity --net hourglass_1 > $OUT/bench_${TAG}_hg1.json 2>> $OUT/bench_$TAG.err; cut -c1-200 $OUT/bench_${TAG}_hg1.json
