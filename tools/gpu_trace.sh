#!/bin/bash
# Kernel timeline of the default train step (two side streams): rocprofv3 --kernel-trace, CSV kept -> tools/timeline.py
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_trace_$TAG.json
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
python bench.py --steps 10 --warmup 3 $COMMON "$@" > $OUT/trace_bench_$TAG.json 2> $OUT/trace_bench_$TAG.err; cut -c1-200 $OUT/trace_bench_$TAG.json
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $COMMON "$@" > $OUT/trace_$TAG.log 2>&1 )
find $OUT/trace_$TAG -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_$TAG.csv
rm -rf $OUT/trace_$TAG
python tools/timeline.py $OUT/kernel_trace_$TAG.csv > $OUT/timeline_$TAG.txt; head -60 $OUT/timeline_$TAG.txt
