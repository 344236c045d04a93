#!/bin/bash
# Round 4: number of weight-gradient side streams re-measured with the final kernels (1 | 2 | 3).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4u; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do
  for n in 1 2 3; do
    python bench.py $C --wgrad-streams $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 streams=$n', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_ab.txt
    python bench.py $C --wgrad-streams $n --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 streams=$n', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_ab.txt
  done
done
