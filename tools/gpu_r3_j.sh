#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python - <<'P'
import torch, awr_amd, os
from awr_amd.trainer import TrainEngine
net = awr_amd.get_deconv_net(18, 14, 2).cuda()
eng = TrainEngine(net, 8, 128, 1.0, autotune=False)
ops = eng.plan.op_names("bwd")
print("bwd ops", len(ops), "bn_bwd_reduce", ops.count("awr_bn_bwd_reduce"))
P
python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" | tail -6
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3j.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bnr2   ', d['value'], d['ms_per_step'])"
AWR_NO_BNR2=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3j.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no bnr2', d['value'], d['ms_per_step'])"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity --deterministic 2>>$OUT/r3j.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('det', d['value'], d['ms_per_step'])"
tail -2 $OUT/r3j.err
