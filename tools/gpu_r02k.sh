#!/bin/bash
TAG=r02k
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_$TAG.log; tail -40 $OUT/gpu_tests_$TAG.log
python bench.py --steps 20 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -5 $OUT/bench_$TAG.err; cut -c1-300 $OUT/bench_$TAG.json
