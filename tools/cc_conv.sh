#!/bin/bash
# compile one csrc/*.hip to /tmp/t with register remarks + ISA (development helper): tools/cc_conv.sh awr_conv [extra flags]
f=${1:-awr_conv}; shift
mkdir -p /tmp/t
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -c /root/repo/awr-adaptive-weighting-regression_amd/csrc/$f.hip -o /tmp/t/$f.o -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/t/$f.remarks.txt
rc=$?
grep -v "remark:" /tmp/t/$f.remarks.txt | head -40
exit $rc
