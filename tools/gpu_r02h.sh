#!/bin/bash
TAG=r02h
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "resnet_18 or dropin or inference" 2>&1 | tail -5
for rep in 1 2; do
for lazy in 0 1; do
AWR_LAZY_ALL=$lazy python bench.py --steps 30 --warmup 5 --no-split-mode --no-extras --no-cpu-baseline --no-parity --per-layer $OUT/per_layer_${TAG}_lazy$lazy.txt > $OUT/bench_${TAG}_lazy$lazy.json 2>> $OUT/bench_$TAG.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_${TAG}_lazy$lazy.json').read()); print('lazy_all=$lazy', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['gemm_seconds_per_step'])"
done; done
