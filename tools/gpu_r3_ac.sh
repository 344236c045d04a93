#!/bin/bash
# round 3, session ac: epilogue with 32-bit row offsets + buffer loads / stores (fewer VALU instructions): tests, microbench, same-box A/B vs the previous library
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
PREV=$GRAFT_REPO_ROOT/awr-adaptive-weighting-regression_amd/lib/libawr_prev.so
python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -2
python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short -x -k "golden" 2>&1 | grep -v "^E        +" | tail -2
echo "== new"; python tools/microbench_1x1_epilogue.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
echo "== prev"; AWR_LIB_PATH=$PREV python tools/microbench_1x1_epilogue.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
C="--no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity"
run() { python bench.py --steps 20 --warmup 5 $C "$@" 2>>$OUT/r3ac.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
echo -n "r18 train new   "; run
echo -n "r18 train prev  "; AWR_LIB_PATH=$PREV run
done
for i in 1 2; do
echo -n "hg1 train new   "; run --net hourglass_1 --steps 10
echo -n "hg1 train prev  "; AWR_LIB_PATH=$PREV run --net hourglass_1 --steps 10
echo -n "hg1 infer new   "; run --mode infer --net hourglass_1 --batch 128
echo -n "hg1 infer prev  "; AWR_LIB_PATH=$PREV run --mode infer --net hourglass_1 --batch 128
echo -n "r18 infer new   "; run --mode infer --batch 128
echo -n "r18 infer prev  "; AWR_LIB_PATH=$PREV run --mode infer --batch 128
done
grep -v "amdgpu.ids" $OUT/r3ac.err | tail -3
