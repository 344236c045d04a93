#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python - <<'P'
import torch, awr_amd, os
from awr_amd.trainer import TrainEngine
for env in ("1", None):
    if env: os.environ["AWR_NO_DS_REORDER"] = env
P
for v in 0 1; do
if [ $v = 1 ]; then export AWR_NO_DS_REORDER=1; else unset AWR_NO_DS_REORDER; fi
python - <<'P'
import torch, awr_amd, os
from awr_amd.trainer import TrainEngine
net = awr_amd.get_deconv_net(18, 14, 2).cuda()
eng = TrainEngine(net, 8, 128, 1.0, autotune=False)
ops = eng.plan.op_names("bwd")
print("NO_DS_REORDER=%s" % os.environ.get("AWR_NO_DS_REORDER"), "bwd ops", len(ops), "bn_bwd_reduce", ops.count("awr_bn_bwd_reduce"))
P
done
unset AWR_NO_DS_REORDER
python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short -x -k "resnet" 2>&1 | grep -v "^E        +" | tail -6
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reorder   ', d['value'], d['ms_per_step'])"
AWR_NO_DS_REORDER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no reorder', d['value'], d['ms_per_step'])"
done
