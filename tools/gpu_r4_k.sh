#!/bin/bash
# Round 4: padding folded into the clamp (fused input affine), unmasked 1x1 variant: parity + same-box numbers for the affine column and the steps.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" | tail -6 | tee $OUT/ops.log
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden and (resnet_18 or hourglass_1)" 2>&1 | grep -v "^E        +" | tail -4 | tee -a $OUT/ops.log
timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep "layer1\|layer2\|hg 1x1 256->128 @64\|hg 3x3\|plain" | tee $OUT/fwdset_aff.txt
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do
  python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench.txt
  python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench.txt
done
python bench.py $C --mode infer --net hourglass_1 --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3', d['value'], d['ms_per_step'], d['mfma_frac'])" | tee -a $OUT/bench.txt
