#!/bin/bash
TAG=r02e
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wave_per_tap or stem" 2>&1 | tail -30
timeout 900 python tools/microbench_gemm.py wgrad 2>&1 | tee $OUT/microbench_wgrad_$TAG.txt
python bench.py --steps 20 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -3 $OUT/bench_$TAG.err; cut -c1-400 $OUT/bench_$TAG.json
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-split-mode --no-extras --wgrad-streams 0 > $OUT/prof_serial_$TAG.log 2>&1 )
find $OUT/prof_serial_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial_$TAG.csv
grep -E "stem" $OUT/kernel_stats_serial_$TAG.csv | awk -F'","' '{print $1, $2, $4}' | cut -c1-200
