#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel-trace summary.  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json     # the profiler runs below reuse the bench run's tile choices (no tuning launches in the traces)
python -m pytest tests -m gpu -q --tb=short -s 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_$TAG.log
tail -3 $OUT/gpu_tests_$TAG.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cat $OUT/bench_$TAG.json
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-split-mode > $OUT/prof_$TAG.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-split-mode --wgrad-streams 0 > $OUT/prof_serial_$TAG.log 2>&1 )
ls -R $OUT/prof_$TAG | head -20
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_split_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --gemm-products 6 --wgrad-streams 0 > $OUT/prof_split_$TAG.log 2>&1 )
