#!/bin/bash
# Round-3 first GPU session: parity localiser runs (Hourglass-1 cw0 gradient question, ResNet18 train-mode map error) + baseline bench of this box.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
D=$OUT/r03_diag_parity.txt; rm -f $D
python tools/diag_parity.py --net hourglass_1 --cw 0 --out $D > /dev/null 2> $OUT/diag.err
AWR_NO_DUAL=1 python tools/diag_parity.py --net hourglass_1 --cw 0 --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net hourglass_1 --cw 0 --streams 0 --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net hourglass_1 --cw 0 --det --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net hourglass_1 --cw 0 --seed 24 --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net hourglass_1 --cw 0 --seed 25 --wseed 10 --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net hourglass_1 --cw 1 --top 10 --out $D > /dev/null 2>> $OUT/diag.err
python tools/diag_parity.py --net resnet_18 --cw 0 --out $D > /dev/null 2>> $OUT/diag.err
tail -5 $OUT/diag.err
grep -c . $D
python bench.py --steps 20 --warmup 5 > $OUT/bench_r03_base.json 2> $OUT/bench_r03_base.err
cut -c1-400 $OUT/bench_r03_base.json
