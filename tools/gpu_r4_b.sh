#!/bin/bash
# Round 4, second GPU pass: full -m gpu suite with the LDS-DMA staging as the default, then same-box A/Bs against AWR_DMA=0.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" > $OUT/gpu_tests.log; tail -8 $OUT/gpu_tests.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2; do
  for m in 0 2; do
    AWR_DMA=$m python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 AWR_DMA=$m', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels'])" | tee -a $OUT/bench_ab.txt
    AWR_DMA=$m python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 AWR_DMA=$m', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
    AWR_DMA=$m python bench.py $C --mode infer --net hourglass_1 --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 AWR_DMA=$m', d['value'], d['ms_per_step'], d['mfma_frac'])" | tee -a $OUT/bench_ab.txt
    AWR_DMA=$m python bench.py $C --mode infer --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 infer b128 AWR_DMA=$m', d['value'], d['ms_per_step'], d['mfma_frac'])" | tee -a $OUT/bench_ab.txt
  done
done
python bench.py $C --per-layer $OUT/per_layer_f32.txt > /dev/null 2>&1
python bench.py $C --net hourglass_1 --per-layer $OUT/per_layer_hg1_train.txt > /dev/null 2>&1
