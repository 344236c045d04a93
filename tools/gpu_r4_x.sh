#!/bin/bash
# Round 4 study: pseudo-random start delay of a launch's first generation of workgroups (variants/stagger, AWR_STAGGER = delay unit of ~1024 cycles, 0..7 units
# per workgroup) against the lock step of equally long workgroups: isolated forward launches, then the Hourglass-1 step.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4x; mkdir -p $OUT
for s in 0 1 2 4 8; do
  echo "AWR_STAGGER=$s" | tee -a $OUT/fwdset.txt
  AWR_STAGGER=$s AWR_LIB_PATH=variants/stagger/libawr_hip.so timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -E "hg |head |deconv 256->256 @32|layer1" | tee -a $OUT/fwdset.txt
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2; do
  for s in 0 2 4; do
    AWR_STAGGER=$s AWR_LIB_PATH=variants/stagger/libawr_hip.so python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 stagger=$s', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
  done
done
