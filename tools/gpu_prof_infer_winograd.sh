TAG=r06; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
prof() { local name=$1; shift; ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$name -o trace -- "$@" > $OUT/prof_${TAG}_$name.log 2>&1 ); find $OUT/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_${TAG}_$name.csv; rm -rf $OUT/prof_${TAG}_$name; }
AWR_WINOGRAD=1 prof infer_r18_b128_winograd python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3
AWR_WINOGRAD=1 prof infer_hg1_b128_winograd python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3 --net hourglass_1
AWR_WINOGRAD=1 python $GRAFT_REPO_ROOT/bench.py --mode infer --net hourglass_1 --batch 128 --steps 10 --warmup 3 --per-layer $OUT/per_layer_${TAG}_hg1_infer_b128_winograd.txt > /dev/null 2>&1
head -4 $OUT/kernel_stats_${TAG}_infer_hg1_b128_winograd.csv | cut -c1-150
