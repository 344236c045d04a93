#!/bin/bash
# fp32 weight-gradient kernel: K-slice depth study (AWR_WBK = 32 / 64 pixels per slice)
set -e
cd $GRAFT_REPO_ROOT
PK=awr-adaptive-weighting-regression_amd
mkdir -p gpurun_out/probe
for v in 32 64; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DAWR_WBK=$v -Iinclude -c $PK/csrc/awr_conv.hip -o gpurun_out/probe/conv_w$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/probe/libawr_w$v.so $PK/lib/awr_head.o $PK/lib/awr_elem.o gpurun_out/probe/conv_w$v.o
  echo "== AWR_WBK=$v"
  AWR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_out/probe/libawr_w$v.so python tools/microbench_gemm.py layers 2>/dev/null | grep -E "^layer|^deconv|\(1, 1\)|\(2, 1\)" | sed 's/fwd.*| wgrad/wgrad/'
done
