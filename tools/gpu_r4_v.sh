#!/bin/bash
# Round 4: phases of a strided data gradient handed out longest-first (default) vs in parity order (AWR_PHASE_ORDER=1): the three strided 3x3 data gradients
# of ResNet18 in the serial per-layer pass, whole step, parity.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4v; mkdir -p $OUT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do
  for v in parity longest; do
    if [ $v = parity ]; then export AWR_PHASE_ORDER=1; else unset AWR_PHASE_ORDER; fi
    python bench.py $C --per-layer $OUT/pl_$v.txt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $v', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
    grep -E "dgrad:layer[234].0.conv1 " $OUT/pl_$v.txt | sed "s/^/$v /" | tee -a $OUT/bench_ab.txt
  done
done
unset AWR_PHASE_ORDER
timeout 1200 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden" 2>&1 | grep -v "^E        +" | tail -3 | tee $OUT/ops.log
