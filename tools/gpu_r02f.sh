#!/bin/bash
TAG=r02f
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json
tools/gpu_tests.sh $TAG
python bench.py --steps 20 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -3 $OUT/bench_$TAG.err; cut -c1-300 $OUT/bench_$TAG.json
AWR_TUNE_CACHE= AWR_WGRAD_ALGO=2 python bench.py --steps 20 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline --no-parity > $OUT/bench_${TAG}_algo2.json 2>> $OUT/bench_$TAG.err; cut -c1-300 $OUT/bench_${TAG}_algo2.json
python bench.py --steps 10 --warmup 3 --no-split-mode --no-extras --no-cpu-baseline --no-parity --net hourglass_1 > $OUT/bench_${TAG}_hg1.json 2>> $OUT/bench_$TAG.err; cut -c1-300 $OUT/bench_${TAG}_hg1.json
AWR_TUNE_CACHE= AWR_WGRAD_ALGO=2 python bench.py --steps 10 --warmup 3 --no-split-mode --no-extras --no-cpu-baseline --no-parity --net hourglass_1 > $OUT/bench_${TAG}_hg1_algo2.json 2>> $OUT/bench_$TAG.err; cut -c1-300 $OUT/bench_${TAG}_hg1_algo2.json
