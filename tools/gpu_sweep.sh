#!/bin/bash
# One-box sweep of the configurations quoted in profiles/rNN_summary.md (train batch sizes, deterministic / serial modes, Hourglass, low-batch inference, split-operand mode)
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { n=$1; shift; python bench.py $C "$@" > $OUT/b_sweep_$n.json 2>$OUT/b_sweep_$n.err; python - <<P
import json
try:
    d=json.load(open("$OUT/b_sweep_$n.json")); print("$n", d["value"], d["ms_per_step"], d.get("roofline",{}).get("step_mfma_frac"), d.get("roofline",{}).get("frac"))
except Exception as e: print("$n failed", e)
P
}
timeout 900 python tools/check_hg2_256.py 128 2>&1 | tail -2
run b64 --steps 30 --warmup 8
run b256 --steps 10 --warmup 3 --batch 256
run b64_det --steps 20 --warmup 5 --deterministic
run b64_serial --steps 20 --warmup 5 --wgrad-streams 0
run b16 --steps 50 --warmup 10 --batch 16
run b4 --steps 50 --warmup 10 --batch 4
run hg1 --steps 20 --warmup 5 --net hourglass_1
run hg1_i64 --mode infer --net hourglass_1 --batch 64 --steps 20 --warmup 5
run hg1_i4 --mode infer --net hourglass_1 --batch 4 --steps 50 --warmup 5
run r18_i4 --mode infer --batch 4 --steps 50 --warmup 5
run x6 --steps 20 --warmup 5 --gemm-products 6
run x6_i128 --mode infer --batch 128 --steps 20 --warmup 5 --gemm-products 6
