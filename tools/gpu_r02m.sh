#!/bin/bash
TAG=r02m
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
tools/gpu_tests.sh $TAG
for b in 64 128; do
python bench.py --mode infer --batch $b --steps 20 --warmup 3 --net hourglass_1 --per-layer $OUT/per_layer_${TAG}_hg1_infer.txt 2>> $OUT/bench_$TAG.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('infer hourglass_1 b$b', d['value'], d['ms_per_step'], d['mfma_frac'])"
done
python bench.py --steps 10 --warmup 3 --no-split-mode --no-extras --no-cpu-baseline --no-parity --net hourglass_1 2>> $OUT/bench_$TAG.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train hourglass_1 b64', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'])"
python tools/check_hg2_256.py 128 2>&1 | tail -1
