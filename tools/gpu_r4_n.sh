#!/bin/bash
# Round 4: stem_pool with the two-plane activation tile: parity, LDS conflict counter, step.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "stem" 2>&1 | tail -3 | tee $OUT/ops.log
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short -x -k "resnet_18 or config2 or config4 or config1" 2>&1 | grep -v "^E        +" | tail -4 | tee -a $OUT/ops.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
for i in 1 2 3; do python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels'])" | tee -a $OUT/bench.txt; done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $C --wgrad-streams 0 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4n/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(f[0])):
    if row["Counter_Name"] == "SQ_LDS_BANK_CONFLICT" and "stem" in row["Kernel_Name"]:
        a = agg[row["Kernel_Name"].split("(")[0]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k, (n, v) in agg.items():
    print("%-60s launches %d  LDS bank conflict cycles per launch %.0f" % (k[:60], n, v / n))
PY
rm -rf $OUT/pmc
