#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras"
run() { n=$1; shift; python bench.py $C "$@" > $OUT/b_r02s_$n.json 2>$OUT/b_r02s_$n.err; python - <<P
import json
try:
    d=json.load(open("$OUT/b_r02s_$n.json")); print("$n", d["value"], d["ms_per_step"])
except Exception as e: print("$n failed", e)
P
}
PREV=$GRAFT_REPO_ROOT/gpurun_prev/libawr_prev.so
for i in 1 2; do
run hg_train_new$i --net hourglass_1 --steps 20 --warmup 6
AWR_LIB_PATH=$PREV run hg_train_old$i --net hourglass_1 --steps 20 --warmup 6
done
for i in 1 2; do
run hg_infer_new$i --net hourglass_1 --mode infer --batch 128 --steps 20 --warmup 6
AWR_LIB_PATH=$PREV run hg_infer_old$i --net hourglass_1 --mode infer --batch 128 --steps 20 --warmup 6
done
for i in 1 2; do
run r18_new$i --steps 20 --warmup 6
AWR_LIB_PATH=$PREV run r18_old$i --steps 20 --warmup 6
done
