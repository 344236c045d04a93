#!/bin/bash
TAG=r02j
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json
tools/gpu_tests.sh $TAG
B="python bench.py --steps 30 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline --no-parity"
show() { python -c "
import json,sys; d=json.loads(open('$1').read()); print('$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['gemm_seconds_per_step'])"; }
$B > $OUT/b_${TAG}_base.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_base.json base
$B --deterministic > $OUT/b_${TAG}_det.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_det.json deterministic
python bench.py --mode infer --batch 128 --steps 20 --warmup 3 --net hourglass_1 2>> $OUT/bench_$TAG.err | cut -c1-330
python bench.py --mode infer --batch 128 --steps 20 --warmup 3 --net resnet_18 2>> $OUT/bench_$TAG.err | cut -c1-330
