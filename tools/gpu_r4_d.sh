#!/bin/bash
# Round 4, fourth GPU pass: config-5 batch-128 test on its own (full traceback), blocked accumulation (AWR_ACCUM=1): parity report + cost,
# per-layer tables with the fragment-side input affine.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -q --tb=long -x -s -k config5 > $OUT/config5_b128.log 2>&1; tail -30 $OUT/config5_b128.log | cut -c1-400
for acc in 0 1; do
  rm -f gpurun_out/parity_report.json
  AWR_ACCUM=$acc timeout 1200 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^E        +" | tail -15 > $OUT/nets_accum$acc.log; tail -4 $OUT/nets_accum$acc.log
  cp gpurun_out/parity_report.json $OUT/parity_report_accum$acc.json
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2; do
  for acc in 0 1; do
    AWR_ACCUM=$acc python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 AWR_ACCUM=$acc', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
    AWR_ACCUM=$acc python bench.py $C --mode infer --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 infer b128 AWR_ACCUM=$acc', d['value'], d['ms_per_step'], d['mfma_frac'])" | tee -a $OUT/bench_ab.txt
    AWR_ACCUM=$acc python bench.py $C --mode infer --net hourglass_1 --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 AWR_ACCUM=$acc', d['value'], d['ms_per_step'], d['mfma_frac'])" | tee -a $OUT/bench_ab.txt
  done
done
python bench.py $C --per-layer $OUT/per_layer_f32.txt > /dev/null 2>&1
python bench.py $C --net hourglass_1 --per-layer $OUT/per_layer_hg1_train.txt > /dev/null 2>&1
