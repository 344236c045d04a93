#!/bin/bash
TAG=r02i
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline --no-parity"
show() { python -c "
import json,sys; d=json.loads(open('$1').read()); print('$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['gemm_seconds_per_step'])"; }
for rep in 1 2; do
$B > $OUT/b_${TAG}_base.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_base.json base
AWR_LIB_PATH=$GRAFT_REPO_ROOT/awr-adaptive-weighting-regression_amd/lib/libawr_hip_prefetch.so $B > $OUT/b_${TAG}_pf.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_pf.json prefetch
AWR_LAZY_MAXC=64 $B > $OUT/b_${TAG}_c64.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_c64.json lazy_maxc64
AWR_LAZY_MAXC=0 $B > $OUT/b_${TAG}_c0.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_c0.json lazy_maxc0
done
python bench.py --steps 10 --warmup 3 --no-split-mode --no-extras --no-cpu-baseline --no-parity --net hourglass_1 --per-layer $OUT/per_layer_${TAG}_hg1_train.txt > $OUT/b_${TAG}_hg1.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_hg1.json hg1_train
python bench.py --mode infer --batch 128 --steps 10 --warmup 3 --net hourglass_1 --per-layer $OUT/per_layer_${TAG}_hg1_infer.txt 2>> $OUT/bench_$TAG.err | cut -c1-300
