#!/bin/bash
TAG=r02d
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wave_per_tap or forward_dgrad_wgrad" 2>&1 | tail -30
timeout 900 python tools/microbench_gemm.py wgrad 2>&1 | tee $OUT/microbench_wgrad_$TAG.txt
