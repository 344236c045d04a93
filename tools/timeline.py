#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of `bench.py` (train mode).

Splits the trace into steps at `adam_kernel`, and for the last steps reports: step span, the time at least one kernel was
running (union), the time >= 2 kernels ran concurrently, idle time, the per-queue busy time, and the largest idle gaps with
the kernels on either side -- i.e. how much of the step the GPU spent waiting for a launch.
Usage: tools/timeline.py kernel_trace.csv [n_steps]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("awr::", "")[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    # only the optimiser launches that close a real step: bench.py's serial event passes at the end replay the plan op by op (one long
    # single-queue segment) and launch the optimiser on its own; a real step has the modal launch count of the segments
    from collections import Counter
    counts = [ends[k] - ends[k - 1] for k in range(1, len(ends))]
    modal = Counter(c for c in counts if c > 50).most_common(1)
    if modal:
        keep = [k for k in range(1, len(ends)) if counts[k - 1] == modal[0][0]]
        segs = [(ends[k - 1], ends[k]) for k in keep]
    else:
        segs = []
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if len(segs) < nsteps:
        print("not enough steps in trace (%d adam launches, %d full steps)" % (len(ends), len(segs)))
        return
    for s in range(len(segs) - nsteps, len(segs)):
        seg = rows[segs[s][0] + 1: segs[s][1] + 1]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        # sweep
        ev = []
        for a, b, _, _ in seg:
            ev.append((a, 1))
            ev.append((b, -1))
        ev.sort()
        depth, last, busy, multi = 0, t0, 0, 0
        for t, d in ev:
            if depth >= 1:
                busy += t - last
            if depth >= 2:
                multi += t - last
            depth += d
            last = t
        span = t1 - t0
        print("step %d: span %.3f ms, >=1 kernel running %.3f ms, >=2 concurrently %.3f ms, idle %.3f ms (%d launches)" %
              (s, span / 1e6, busy / 1e6, multi / 1e6, (span - busy) / 1e6, len(seg)))
        perq = {}
        for a, b, _, q in seg:
            perq[q] = perq.get(q, 0) + b - a
        print("   busy per queue: " + ", ".join("q%s %.3f ms" % (q, v / 1e6) for q, v in sorted(perq.items())))
        if s == len(segs) - 1:
            # idle gaps: times with depth 0
            gaps = []
            cur_end, prev = seg[0][1], seg[0]
            for r in seg[1:]:
                if r[0] > cur_end:
                    gaps.append((r[0] - cur_end, prev[2], r[2], (cur_end - t0) / 1e6))
                if r[1] > cur_end:
                    cur_end, prev = r[1], r
            gaps.sort(reverse=True)
            print("   %d idle gaps, total %.3f ms; histogram (us): " % (len(gaps), sum(g[0] for g in gaps) / 1e6) +
                  ", ".join("%s:%d" % (lab, sum(1 for g in gaps if lo <= g[0] / 1e3 < hi)) for lab, lo, hi in
                            (("<2", 0, 2), ("2-5", 2, 5), ("5-10", 5, 10), ("10-20", 10, 20), (">=20", 20, 1e9))))
            for g in gaps[:25]:
                print("   gap %6.1f us at %7.3f ms  after %-44s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
            # main-queue chain: which kernels sit alone (depth 1) the longest, by name
            alone = {}
            ev2 = sorted([(a, 1, i) for i, (a, b, _, _) in enumerate(seg)] + [(b, -1, i) for i, (a, b, _, _) in enumerate(seg)])
            live, last = set(), t0
            for t, d, i in ev2:
                if len(live) == 1:
                    n = seg[next(iter(live))][2]
                    alone[n] = alone.get(n, 0) + t - last
                if d > 0:
                    live.add(i)
                else:
                    live.discard(i)
                last = t
            print("   time running ALONE, by kernel (top 15):")
            for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:15]:
                print("      %8.1f us  %s" % (v / 1e3, n))


if __name__ == "__main__":
    main()
