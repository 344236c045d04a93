#!/bin/bash
# Round 4: plan autotuner that times a side-stream weight gradient BESIDE its layer's data gradient (pair makespan) vs every launch alone (AWR_TUNE_CORUN=0).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4r; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "golden or bitwise or side_streams or autotun" 2>&1 | grep -v "^E        +" | tail -4 | tee $OUT/ops.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['other_kernels']['conv_wgrad_kernel']['tflops'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2 3 4; do
  run "tune-alone" AWR_TUNE_CORUN=0
  run "tune-beside" AWR_X=0
done
