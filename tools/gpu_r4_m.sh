#!/bin/bash
# Round 4: Hourglass residuals with bn2's output written out (plain input for the 3x3 conv2) vs the loader affine: parity, Hourglass-1 / config 5 steps.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4m; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "hourglass" 2>&1 | grep -v "^E        +" | tail -5 | tee $OUT/nets.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do
  for m in lazy written; do
    E="AWR_X=0"; [ $m = lazy ] && E="AWR_HG_LAZY_BN2=1"
    env $E python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 bn2-$m', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
  done
done
for m in lazy written; do
  E="AWR_X=0"; [ $m = lazy ] && E="AWR_HG_LAZY_BN2=1"
  env $E python tools/check_hg2_256.py 128 2>&1 | grep "HG-2" | sed "s/^/bn2-$m /" | tee -a $OUT/bench_ab.txt
done
python bench.py $C --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_hg1.txt > /dev/null 2>&1
