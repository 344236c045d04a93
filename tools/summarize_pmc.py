"""Per-kernel HBM traffic and SQ counters from the rocprofv3 --pmc passes of tools/gpu_pmc.sh.

    python tools/summarize_pmc.py gpurun_out r01c > profiles/r01_hbm_traffic_pmc.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, section HBM), hence `fetch_MB_per_launch_x2`.  Kernel names are cut at the argument list."""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    fetch = load(os.path.join(out_dir, "pmc_fetch_%s" % tag, "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(out_dir, "pmc_write_%s" % tag, "pmc_counter_collection.csv"), "WRITE_SIZE")
    sq = {c: load(os.path.join(out_dir, "pmc_sq_%s" % tag, "pmc_counter_collection.csv"), c)
          for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")}
    res = {}
    for name, (n, kb) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        if not name.startswith("awr::"):
            continue
        ent = {"launches": n, "fetch_MB_per_launch_x2": round(2.0 * kb / n / 1024.0, 3)}
        if name in write:
            ent["write_MB_per_launch"] = round(write[name][1] / write[name][0] / 1024.0, 3)
        gui = sq["GRBM_GUI_ACTIVE"].get(name)
        if gui and gui[1] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs (256 CUs x 4)
            ent["mfma_busy_frac"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"][name][1] / (gui[1] / 8.0 * 1024.0), 4)
            wave = sq["SQ_WAVE_CYCLES"][name][1]
            if wave > 0:
                ent["wait_inst_any_frac_of_wave_cycles"] = round(sq["SQ_WAIT_INST_ANY"][name][1] / wave, 4)
                ent["wait_any_frac_of_wave_cycles"] = round(sq["SQ_WAIT_ANY"][name][1] / wave, 4)
            ent["lds_bank_conflict_cycles_per_launch"] = round(sq["SQ_LDS_BANK_CONFLICT"][name][1] / sq["SQ_LDS_BANK_CONFLICT"][name][0], 1)
        res[name] = ent
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
