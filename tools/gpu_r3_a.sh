#!/bin/bash
# Round-3 session A: new NHWC kernels + yardstick + bench plumbing tests, NHWC vs NCHW A/B on this box.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_head_gpu.py tests/test_abi.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $OUT/r3a_head.log; tail -5 $OUT/r3a_head.log
python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -s 2>&1 | grep -v "^E        +" | tail -60 > $OUT/r3a_nets.log; tail -30 $OUT/r3a_nets.log
python -m pytest tests/test_bench_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -30 > $OUT/r3a_bench.log; tail -12 $OUT/r3a_bench.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 2>>$OUT/r3a.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NHWC', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'], json.dumps(d['roofline_hbm']))"
AWR_NCHW_BOUNDARY=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 2>>$OUT/r3a.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NCHW', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'], json.dumps(d['roofline_hbm']))"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --coord-weight 1 2>>$OUT/r3a.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NHWC cw1', d['value'], d['ms_per_step'], json.dumps(d['roofline_hbm']))"
AWR_NCHW_BOUNDARY=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --coord-weight 1 2>>$OUT/r3a.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NCHW cw1', d['value'], d['ms_per_step'], json.dumps(d['roofline_hbm']))"
tail -5 $OUT/r3a.err
