#!/bin/bash
# Round profiles: rocprofv3 kernel-trace stats of the bench commands (train, serial train, inference, Hourglass) + separate PMC passes.
# Usage: tools/gpu_profiles.sh <tag>   -> gpurun_out/prof_<tag>_*/, kernel_stats_<tag>_*.csv, pmc_summary_<tag>.json
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_prof_$TAG.json
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 --no-accurate-mode --no-data-path --no-winograd"
python bench.py --steps 20 --warmup 5 --no-split-mode > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; cut -c1-250 $OUT/bench_$TAG.json
python bench.py --steps 5 --warmup 2 $COMMON --wgrad-streams 0 --per-layer $OUT/per_layer_${TAG}_f32.txt > /dev/null 2>> $OUT/bench_$TAG.err
python bench.py --steps 5 --warmup 2 $COMMON --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_${TAG}_hg1_train.txt > /dev/null 2>> $OUT/bench_$TAG.err
python bench.py --mode infer --net hourglass_1 --batch 128 --steps 10 --warmup 3 $COMMON --per-layer $OUT/per_layer_${TAG}_hg1_infer_b128.txt > /dev/null 2>> $OUT/bench_$TAG.err
# the opt-in Winograd mode "forward+wgrad" (AWR_WINOGRAD=3), serial replay per launch + its kernel statistics
AWR_WINOGRAD=3 python bench.py --steps 5 --warmup 2 $COMMON --wgrad-streams 0 --per-layer $OUT/per_layer_${TAG}_f32_winograd.txt > /dev/null 2>> $OUT/bench_$TAG.err
AWR_WINOGRAD=3 python bench.py --steps 5 --warmup 2 $COMMON --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_${TAG}_hg1_train_winograd.txt > /dev/null 2>> $OUT/bench_$TAG.err
prof() {  # name, command...
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$name -o trace -- "$@" > $OUT/prof_${TAG}_$name.log 2>&1 )
  find $OUT/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_${TAG}_$name.csv
  rm -rf $OUT/prof_${TAG}_$name
}
prof train python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $COMMON
prof train_serial python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $COMMON --wgrad-streams 0
prof infer_r18_b128 python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3
prof infer_hg1_b128 python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3 --net hourglass_1
prof train_hg1 python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $COMMON --net hourglass_1
AWR_WINOGRAD=3 prof train_hg1_winograd python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $COMMON --net hourglass_1
AWR_WINOGRAD=3 prof train_winograd python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $COMMON
AWR_WINOGRAD=1 prof infer_r18_b128_winograd python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3
AWR_WINOGRAD=1 prof infer_hg1_b128_winograd python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 128 --steps 10 --warmup 3 --net hourglass_1
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $COMMON --wgrad-streams 0"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $CMD > $OUT/pmc_sq_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $CMD > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $CMD > $OUT/pmc_write_$TAG.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $OUT $TAG > $OUT/pmc_summary_$TAG.json
rm -rf $OUT/pmc_sq_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
head -c 1200 $OUT/pmc_summary_$TAG.json
ls $OUT | grep $TAG
