#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P="python $GRAFT_REPO_ROOT/tools/pmc_wino_wgrad.py"
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmcww_$n -o pmc -- $P > $OUT/pmcww_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass b SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA
pass c SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES
python - <<'P'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for n in "abc":
    for f in glob.glob(out + "/pmcww_%s/**/*counter_collection.csv" % n, recursive=True):
        seen = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wino_wgrad" not in k: continue
            k = k[k.index("wino_wgrad"):k.index("(")] if "(" in k else k
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if n == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c in sorted(d): print("   %-28s %16.0f per launch" % (c, d[c] / n))
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print("   -> wait_inst_any/wave_cycles %.3f  wait_any %.3f  mfma_busy/busy %.3f  valu %.3f lds %.3f vmem %.3f  avg VMEM latency (level/insts) %.0f cycles  avg waves %.1f" % (
        d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(d.get("SQ_BUSY_CYCLES", 1), 1),
        d.get("SQ_ACTIVE_INST_VALU", 0) / wc, d.get("SQ_ACTIVE_INST_LDS", 0) / wc, d.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        d.get("SQ_INST_LEVEL_VMEM", 0) / max(d.get("SQ_INSTS_VMEM_RD", 0) + d.get("SQ_INSTS_VMEM_WR", 0), 1), d.get("SQ_LEVEL_WAVES", 0) / max(d.get("GRBM_GUI_ACTIVE", 1), 1)))
P
rm -rf $OUT/pmcww_a $OUT/pmcww_b $OUT/pmcww_c
