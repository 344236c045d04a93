#!/bin/bash
# Round 4: final kernels vs the convolution kernels of commit fef8340 (before the second pass over the weight-gradient kernels) under the SAME host code
# (variants/oldconv = today's objects + the old awr_conv.o): ResNet18, Hourglass-1, config 5 on one box; the two new tests.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_net_abi_gpu.py tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "c_abi_only or tuning_cache" 2>&1 | grep -v "^E        +" | tail -6 | tee $OUT/ops.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3 4; do
  run "old-kernels" AWR_LIB_PATH=variants/oldconv/libawr_hip.so
  run "final" AWR_X=0
done
for i in 1 2; do
  for lib in variants/oldconv/libawr_hip.so ""; do
    AWR_LIB_PATH=$lib python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 lib=$lib |" | tee -a $OUT/bench_ab.txt
  done
done
