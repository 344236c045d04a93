#!/bin/bash
# Round 4 study: weight gradients as fewer, longer workgroups (AWR_WGRAD_CAP = workgroup target of the kernel-row kernel, 3x that for the per-tap kernels) --
# a smaller share of every CU for longer, beside the data-gradient chain and its memory-bound passes.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do
  for cap in 0 512 384 256; do
    AWR_WGRAD_CAP=$cap python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 cap=$cap', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
    AWR_WGRAD_CAP=$cap python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 cap=$cap', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
  done
done
