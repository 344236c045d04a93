#!/bin/bash
# Round 4: pipelined loops without exits inside the trip (kernel-row weight gradient 123 -> 81 registers, per-tap LDS-DMA weight gradient 162 -> 100 /
# 96 -> 69 / 50 -> 36, forward GEMM 104 -> 94), the kernel-row kernel's fused BatchNorm loader as an in-LDS pass one stage ahead (three stage buffers):
# parity, isolated launches (in-tree vs variants/head, fence4, gemmexits), whole-step A/B.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wgrad or gemm or conv or staging" 2>&1 | tail -3 | tee $OUT/ops.log
for lib in "" variants/head/libawr_hip.so variants/fence4/libawr_hip.so; do
  AWR_LIB_PATH=$lib timeout 900 python tools/microbench_gemm.py rowset 2>&1 | tee -a $OUT/rowset.txt
done
for v in 0 1; do AWR_WGRAD_DMA=$v timeout 900 python tools/microbench_gemm.py wgradset 2>&1 | tee -a $OUT/wgradset.txt; done
for lib in "" variants/gemmexits/libawr_hip.so; do
  echo "lib=$lib" | tee -a $OUT/fwdset.txt; AWR_LIB_PATH=$lib timeout 900 python tools/microbench_gemm.py fwdset 2>&1 | tee -a $OUT/fwdset.txt
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3; do
  run "new" AWR_X=0
  run "head" AWR_LIB_PATH=variants/head/libawr_hip.so
  run "new+wgrad-dma" AWR_WGRAD_DMA=1
  run "gemm-loop-exits" AWR_LIB_PATH=variants/gemmexits/libawr_hip.so
done
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden or bitwise or side_streams" 2>&1 | grep -v "^E        +" | tail -4 | tee -a $OUT/ops.log
