#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
D=$OUT/r03_diag2.txt; rm -f $D
python tools/diag_parity.py --net hourglass_1 --cw 0 --top 12 --out $D > /dev/null 2> $OUT/diag2.err
python tools/diag_parity.py --net resnet_18 --cw 0 --top 8 --out $D > /dev/null 2>> $OUT/diag2.err
tail -3 $OUT/diag2.err
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r3b -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 > $OUT/prof_r3b.log 2>&1 )
find $OUT/prof_r3b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_r3b.csv
rm -rf $OUT/prof_r3b
head -30 $OUT/kernel_stats_r3b.csv | cut -c1-160
