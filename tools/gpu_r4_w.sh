#!/bin/bash
# Round 4: hipGraph replay vs eager issue with the final kernels, train step and inference, batch 4 / 16 / 64.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4w; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 --no-hourglass-train"
for b in 4 16 64; do
  for g in "" "--graph"; do
    for i in 1 2; do
      python bench.py $C --batch $b $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train b=$b graph=${g:-no}', d['value'], d['ms_per_step'])" | tee -a $OUT/graph.txt
      python bench.py $C --batch $b $g --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer b=$b graph=${g:-no}', d['value'], d['ms_per_step'])" | tee -a $OUT/graph.txt
    done
  done
done
