#!/bin/bash
# Round 4: where the fragment-side input affine costs (probe builds: 1 = no padding mask, 2 = coefficients from registers, 3 = both; wrong results, timing only);
# row kernel with the branch-free padding.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wgrad" 2>&1 | tail -3 | tee $OUT/ops.log
for v in default aff1 aff2 aff3; do
  LIBV=""; [ $v != default ] && LIBV="AWR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libawr_$v.so"
  echo "== $v" | tee -a $OUT/fwdset_aff.txt
  env $LIBV timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep "layer1\|layer2\|hg 1x1 256->128 @64\|hg 3x3\|plain" | tee -a $OUT/fwdset_aff.txt
done
timeout 600 python tools/microbench_gemm.py wgradset 2>&1 | grep "3x3 \|hg 3x3" | grep -v s2 | sed 's/(1, 1)\/2048.*row\/512/ ... row\/512/' | tee $OUT/wgradset_row.txt
