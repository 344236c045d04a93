#!/bin/bash
# Final round-3 session on ONE box: full -m gpu suite, smoke, the profile set (tools/gpu_profiles.sh), the sweep, the step timeline, the bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E        +" > $OUT/gpu_tests_r03final.log; tail -8 $OUT/gpu_tests_r03final.log
cp $OUT/parity_report.json $OUT/parity_report_r03final.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profiles.sh r03 > $OUT/profiles_r03.log 2>&1; tail -3 $OUT/profiles_r03.log
bash tools/gpu_trace.sh r03 > /dev/null 2>&1; head -3 $OUT/timeline_r03.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_r03final.json 2> $OUT/bench_r03final.err; cut -c1-200 $OUT/bench_r03final.json
bash tools/gpu_sweep.sh > $OUT/sweep_r03final.txt 2>&1; grep -v amdgpu.ids $OUT/sweep_r03final.txt
