#!/bin/bash
# Round 4, fifth GPU pass: BatchNorm backward off the critical chain (data gradients evaluate it from (g, y)); fused-pair batch chunks fixed;
# blocked accumulation parity reports after the initialiser fix.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" | tail -15 | tee $OUT/ops.log
for acc in 0 1; do
  rm -f gpurun_out/parity_report.json
  AWR_ACCUM=$acc timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^E        +" | tail -25 > $OUT/nets_accum$acc.log; tail -6 $OUT/nets_accum$acc.log
  cp gpurun_out/parity_report.json $OUT/parity_report_accum$acc.json
done
timeout 1200 python -m pytest tests/test_full_size_gpu.py -m gpu -q --tb=short -x -s 2>&1 | grep -v "^E        +" | tail -25 | tee $OUT/full_size.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2; do
  run "classic-bn-bwd" AWR_NO_LAZY_BNB=1
  run "lazy-bn-bwd" AWR_X=1
  run "UPPER-BOUND-no-apply(wrong results)" AWR_NO_LAZY_BNB=1 AWR_EXP_NO_BN_BWD_APPLY=1
done
python bench.py $C --per-layer $OUT/per_layer_f32.txt > /dev/null 2>&1
python bench.py $C --net hourglass_1 --per-layer $OUT/per_layer_hg1_train.txt > /dev/null 2>&1
