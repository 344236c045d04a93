#!/bin/bash
# Round 4, first GPU pass of the LDS-DMA GEMM staging (AWR_DMA=1/2/3 vs 0): operator parity, isolated layers, whole step.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4a; mkdir -p $OUT
for m in 1 2 3; do
  echo "== AWR_DMA=$m operator parity" | tee -a $OUT/tests.log
  AWR_DMA=$m timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "conv or stem_im2col or affine or chunking" 2>&1 | tail -5 | tee -a $OUT/tests.log
done
for m in 0 1 2 3; do
  AWR_DMA=$m timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v Warning | tee $OUT/fwdset_dma$m.txt
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2; do
  for m in 0 1 2 3; do
    AWR_DMA=$m python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AWR_DMA=$m', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
  done
done
