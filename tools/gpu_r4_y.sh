#!/bin/bash
# Round 4: one more resident wave per SIMD for the GEMM instantiations that sit 2-4 registers above a register step (variants/occA: reduction epilogues of the 64x128 /
# 128x64 tiles 82-83 -> 80 = six waves, EM 4 99 -> 96 = five, no spills; occB: + the operand-prefetch forms 98-100 -> 96 with 4-8 spilled) vs the in-tree build (HEAD).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4y; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3 4; do
  run "head" AWR_X=0
  run "occA" AWR_LIB_PATH=variants/occA/libawr_hip.so
  run "occB" AWR_LIB_PATH=variants/occB/libawr_hip.so
done
