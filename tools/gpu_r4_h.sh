#!/bin/bash
# Round 4, kernel-row weight gradient (algo 3): parity, isolated launches against the per-tap kernel (register / DMA staging, incl. the 128x128 tile), whole step.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wgrad or every_tile or forward_dgrad" 2>&1 | grep -v "^E        +" | tail -15 | tee $OUT/ops.log
AWR_WGRAD_DMA=0 timeout 900 python tools/microbench_gemm.py wgradset 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/wgradset_reg.txt
AWR_WGRAD_DMA=1 AWR_WGRAD_KP=16 timeout 900 python tools/microbench_gemm.py wgradset 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/wgradset_dma16.txt
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2; do
  run "per-tap" AWR_X=0
  run "kernel-row" AWR_WGRAD_ROW=1
done
AWR_WGRAD_ROW=1 python bench.py $C --wgrad-streams 0 --per-layer $OUT/per_layer_f32_row.txt > /dev/null 2>&1
