#!/bin/bash
# Round 4: why faster weight-gradient kernels (isolated +4-9 %) made the step slower (tools/gpu_r4_p.sh: +0.3 % ResNet18, +1 % Hourglass-1): resident
# waves of the kernel-row kernel capped at four (variants/row44), the round-3 kernel choice on the big lazy layers (AWR_KEEP_TAPS), per-tap LDS-DMA kernels.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4q; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3; do
  run "head" AWR_LIB_PATH=variants/head/libawr_hip.so
  run "new" AWR_X=0
  run "new+keep-taps" AWR_KEEP_TAPS=1
  run "row44" AWR_LIB_PATH=variants/row44/libawr_hip.so
  run "row44+keep-taps" AWR_LIB_PATH=variants/row44/libawr_hip.so AWR_KEEP_TAPS=1
  run "new+wgrad-dma" AWR_WGRAD_DMA=1
  run "row44+wgrad-dma" AWR_LIB_PATH=variants/row44/libawr_hip.so AWR_WGRAD_DMA=1
done
