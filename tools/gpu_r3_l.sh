#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3l.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default       ', d['value'], d['ms_per_step'])"
AWR_BUCKET_SCATTER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3l.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bucket scatter', d['value'], d['ms_per_step'])"
done
AWR_BUCKET_SCATTER=1 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "fused_train_step_golden or deterministic or full" 2>&1 | tail -4
tail -2 $OUT/r3l.err
