#!/bin/bash
# Kernel timeline of the inference path: tools/gpu_trace_infer.sh <tag> <bench args>
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode infer --steps 6 --warmup 3 "$@" > $OUT/trace_$TAG.log 2>&1 )
find $OUT/trace_$TAG -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_$TAG.csv
rm -rf $OUT/trace_$TAG
python - <<P
import csv,re
rows=[]
for r in csv.DictReader(open("$OUT/kernel_trace_$TAG.csv")):
    n=re.sub(r"\(.*$","",re.sub(r"^void ","",r["Kernel_Name"])).replace("awr::","")[:44]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,r["Queue_Id"],r["Grid_Size_X"],r["Workgroup_Size_X"]))
rows.sort()
heads=[i for i,r in enumerate(rows) if r[2].startswith("head_fwd")]
seg=rows[heads[-2]+1:heads[-1]+1]
t0=seg[0][0]
print("one inference call: %d launches, span %.1f us, sum of kernel durations %.1f us"%(len(seg),(seg[-1][1]-t0)/1e3,sum(r[1]-r[0] for r in seg)/1e3))
for r in seg: print("%8.1f %7.1f q%s %-44s grid %s/%s"%((r[0]-t0)/1e3,(r[1]-r[0])/1e3,r[3],r[2],r[4],r[5]))
P
