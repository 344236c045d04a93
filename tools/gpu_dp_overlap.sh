#!/bin/bash
# 2 ranks on the one GPU, rocprofv3 kernel + memory-copy trace of both, analysis of rank 0 -> gpurun_out/dp_overlap_<tag>.txt
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/dp_trace_$TAG -o trace_%pid% -- \
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $GRAFT_REPO_ROOT/tools/dp_overlap.py worker 32 8 > $OUT/dp_trace_$TAG.log 2>&1 )
grep -h "buckets" $OUT/dp_trace_$TAG.log > $OUT/dp_overlap_$TAG.txt
find $OUT/dp_trace_$TAG -name "*kernel_trace.csv" -size +50k | sort | head -4
K=$(find $OUT/dp_trace_$TAG -name "*kernel_trace.csv" -size +50k | sort | head -1)
M=${K/kernel_trace/memory_copy_trace}
ls -la $K $M
python tools/dp_overlap.py analyse $K $M >> $OUT/dp_overlap_$TAG.txt
head -3 $M
rm -rf $OUT/dp_trace_$TAG
cat $OUT/dp_overlap_$TAG.txt; tail -5 $OUT/dp_trace_$TAG.log
