#!/bin/bash
# Round 4: kernel-row weight gradient -- fragment look-ahead variants (AWR_ROW_FENCE = 2 shipped / 4 / 8 pixel pairs), new split-K default, whole step with the autotuner choosing.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "wgrad" 2>&1 | tail -3 | tee $OUT/ops.log
for v in default fence4 fence8; do
  LIBV=""; [ $v != default ] && LIBV="AWR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libawr_$v.so"
  echo "== $v" | tee -a $OUT/wgradset_row.txt
  env $LIBV timeout 600 python tools/microbench_gemm.py wgradset 2>&1 | grep "3x3 \|hg 3x3" | grep -v s2 | sed 's/(1, 1)\/2048.*row\/512/ ... row\/512/' | tee -a $OUT/wgradset_row.txt
done
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels']['conv_wgrad_kernel'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2; do
  run "row-off" AWR_WGRAD_ROW=0
  run "default(tuner)" AWR_X=0
  run "row-everywhere" AWR_WGRAD_ROW=1
done
python bench.py $C --wgrad-streams 0 --per-layer $OUT/per_layer_f32.txt > /dev/null 2>&1
python bench.py $C --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_hg1.txt > /dev/null 2>&1
