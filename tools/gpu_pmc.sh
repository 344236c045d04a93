#!/bin/bash
# PMC passes over a short bench run (separate passes; --kernel-trace only, as gpurun requires).
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 --no-accurate-mode --no-data-path --no-winograd --wgrad-streams 0 $EXTRA"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $CMD > $OUT/pmc_sq_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $CMD > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $CMD > $OUT/pmc_write_$TAG.log 2>&1
ls -la $OUT/pmc_sq_$TAG $OUT/pmc_fetch_$TAG | head -20
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $OUT $TAG > $OUT/pmc_summary_$TAG.json
