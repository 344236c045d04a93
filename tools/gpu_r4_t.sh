#!/bin/bash
# Round 4: several tiles per workgroup in the LDS-DMA forward / data-gradient GEMM (next tile's first stage requested before the epilogue): parity,
# isolated launches per tile count, whole steps (AWR_GEMM_TPW=1 = one tile per workgroup everywhere, unset = automatic).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward" 2>&1 | tail -3 | tee $OUT/ops.log
timeout 900 python tools/microbench_gemm.py tpwset 2>&1 | grep -v amdgpu.ids | tee $OUT/tpwset.txt
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --batch 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b256 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3; do
  run "one-tile" AWR_GEMM_TPW=1
  run "auto" AWR_X=0
done
for i in 1 2; do
  for v in 1 0; do
    AWR_GEMM_TPW=$v python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 AWR_GEMM_TPW=$v |" | tee -a $OUT/bench_ab.txt
  done
done
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden or bitwise" 2>&1 | grep -v "^E        +" | tail -3 | tee -a $OUT/ops.log
