#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_head_gpu.py -m gpu -q --tb=short 2>&1 | tail -6
python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -s -k "yardstick or trainer_end" 2>&1 | grep -v "^E        +" | tail -40 > $OUT/r3d_nets.log; grep -E "median error|passed|failed|Error" $OUT/r3d_nets.log | cut -c1-400
for depth in 1 2; do for wgs in 512 1024 2048 4096; do
AWR_NHWC_DEPTH=$depth AWR_NHWC_WGS=$wgs python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity 2>>$OUT/r3d.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('depth $depth wgs $wgs', d['ms_per_step'], h['head_loss_step_nhwc']['avg_us'], h['head_forward_nhwc']['avg_us'])"
done; done
AWR_NHWC_DEPTH=2 AWR_NHWC_WGS=1024 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-mode --no-extras --no-b256 --no-parity --coord-weight 1 2>>$OUT/r3d.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('cw1 depth 2 wgs 1024', d['ms_per_step'], h['head_loss_step_nhwc']['avg_us'], h['head_forward_nhwc']['avg_us'])"
tail -3 $OUT/r3d.err
