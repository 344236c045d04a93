#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_r03f.log; tail -30 $OUT/gpu_tests_r03f.log
cp $OUT/parity_report.json $OUT/parity_report_r03f.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
