#!/bin/bash
# Bottleneck study of conv_gemm_kernel: build probe variants (AWR_PROBE=1..4, see awr_conv.hip) and run the layer microbenchmark.
set -e
cd $GRAFT_REPO_ROOT
PK=awr-adaptive-weighting-regression_amd
mkdir -p gpurun_out/probe
for v in ${PROBES:-0 1 2 3 4}; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DAWR_PROBE=$v -c $PK/csrc/awr_conv.hip -o gpurun_out/probe/conv_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/probe/libawr_probe_$v.so $PK/lib/awr_head.o $PK/lib/awr_elem.o gpurun_out/probe/conv_$v.o
  echo "== AWR_PROBE=$v"
  AWR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_out/probe/libawr_probe_$v.so python tools/microbench_gemm.py tiles 2>/dev/null
done
