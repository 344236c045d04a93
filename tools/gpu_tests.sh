#!/bin/bash
# GPU box: the -m gpu parity suite (+ optional extra pytest args), log + parity report into gpurun_out/.  Usage: tools/gpu_tests.sh <tag> [pytest args]
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --tb=short -x "$@" 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_$TAG.log
tail -25 $OUT/gpu_tests_$TAG.log
cp $OUT/parity_report.json $OUT/parity_report_$TAG.json 2>/dev/null
