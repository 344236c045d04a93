#!/bin/bash
# Round 4: stem kernels (patch pitch 49 in stem_pool, swizzled scatter tile in stem_bwd): parity + LDS conflict counters; weight gradients issued behind
# their data gradient (AWR_WGRAD_LATE=1) vs beside it.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "stem" 2>&1 | tail -3 | tee $OUT/ops.log
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "resnet_18 or bitwise or side_streams" 2>&1 | grep -v "^E        +" | tail -4 | tee -a $OUT/ops.log
AWR_WGRAD_LATE=1 timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden and (resnet_18 or hourglass_1)" 2>&1 | grep -v "^E        +" | tail -3 | tee -a $OUT/ops.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['stem kernels (fused direct 5x5 conv+BN+ReLU+pool, fwd+bwd incl. recomputation)'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3; do
  run "wgrad-beside" AWR_X=0
  run "wgrad-late" AWR_WGRAD_LATE=1
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $C --wgrad-streams 0 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r4o/stem_conflicts.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r4o/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(f[0])):
    if row["Counter_Name"] == "SQ_LDS_BANK_CONFLICT" and "stem" in row["Kernel_Name"]:
        a = agg[row["Kernel_Name"].split("(")[0]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k, (n, v) in agg.items():
    print("%-60s launches %d  LDS bank conflict cycles per launch %.0f" % (k[:60], n, v / n))
PY
rm -rf $OUT/pmc
python bench.py $C --wgrad-streams 0 --per-layer $OUT/per_layer_f32.txt > /dev/null 2>&1; grep stem $OUT/per_layer_f32.txt
