#!/bin/bash
# Round 4, third GPU pass: fragment-side input affine in the LDS-DMA GEMM, LDS-DMA weight gradient (stage depth 16 / 32 pixels) -- parity, then A/Bs.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c; mkdir -p $OUT
for kp in 0 16 32; do
  echo "== AWR_WGRAD_KP=$kp operator parity" | tee -a $OUT/tests.log
  AWR_WGRAD_KP=$kp timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -4 | tee -a $OUT/tests.log
done
timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" | tail -12 | tee -a $OUT/tests.log
AWR_WGRAD_DMA=0 timeout 600 python tools/microbench_gemm.py wgradset 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/wgradset_dma0.txt
for kp in 16 32; do
  AWR_WGRAD_KP=$kp timeout 600 python tools/microbench_gemm.py wgradset 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/wgradset_kp$kp.txt
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { # label, env...
  lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2; do
  run "dma0" AWR_DMA=0
  run "gemm-dma,wgrad-reg" AWR_WGRAD_DMA=0
  run "wgrad-kp-auto" AWR_WGRAD_KP=0
  run "wgrad-kp16" AWR_WGRAD_KP=16
  run "wgrad-kp32" AWR_WGRAD_KP=32
done
