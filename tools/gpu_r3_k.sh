#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nets_gpu.py tests/test_net_abi_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E        +" | tail -6
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap   ', d['value'], d['ms_per_step'])"
AWR_NO_PACK_OVERLAP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no overlap', d['value'], d['ms_per_step'])"
done
AWR_NO_PACK_OVERLAP=1 AWR_NO_BNR2=1 AWR_NO_DS_REORDER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('none of the three', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity --graph 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --net hourglass_1 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1', d['value'], d['ms_per_step'])"
AWR_NO_PACK_OVERLAP=1 python bench.py --steps 10 --warmup 3 --net hourglass_1 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 no overlap', d['value'], d['ms_per_step'])"
tail -2 $OUT/r3k.err
