#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py -m gpu -q --tb=short -k "batchnorm or nhwc" 2>&1 | tail -8
python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -k "resnet_50" 2>&1 | grep -v "^E        +" | tail -25
python bench.py --steps 5 --warmup 2 --net resnet_50 --batch 32 --no-cpu-baseline --no-split-mode --no-extras 2>&1 | cut -c1-400
