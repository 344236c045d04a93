#!/bin/bash
# Round 4 checkpoint on ONE box: accumulation probe, full -m gpu suite, smoke, the default bench line (timed: the driver runs exactly this).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E        +" > $OUT/gpu_tests.log ) 2>&1 | grep real; grep -E "passed|failed" $OUT/gpu_tests.log | tail -4
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
cut -c1-300 $OUT/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4g/bench.json"))
for k in ("b256", "hg1_train_b64", "config5", "config3", "split_mode", "cpu_baseline", "joint_err_mm_vs_oracle"):
    print(k, json.dumps(d.get(k))[:400])
print("forward", json.dumps(d.get("forward"))[:500])
PY
bash tools/gpu_profiles.sh r04 > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
bash tools/gpu_trace.sh r04 > /dev/null 2>&1; head -3 gpurun_out/timeline_r04.txt
