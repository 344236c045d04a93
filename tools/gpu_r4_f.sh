#!/bin/bash
# Round 4, sixth GPU pass: low-priority side streams, BatchNorm backward in the consumer for 1x1 convs only, accumulation-mode probe.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4f; mkdir -p $OUT
python tools/probes/accum_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/accum_probe.txt
timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "hourglass_1 or resnet_18" 2>&1 | grep -v "^E        +" | tail -6 | tee $OUT/nets.log
C="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256"
run() { lab=$1; shift
  env "$@" python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r18 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/bench_ab.txt
  env "$@" python bench.py $C --net hourglass_1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hg1 b64 $lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'])" | tee -a $OUT/bench_ab.txt
}
for i in 1 2; do
  run "lazy0" AWR_LAZY_BNB=0
  run "lazy1(1x1)" AWR_LAZY_BNB=1
  run "lazy0+side-low-prio" AWR_LAZY_BNB=0 AWR_SIDE_PRIORITY=low
  run "lazy1+side-low-prio" AWR_LAZY_BNB=1 AWR_SIDE_PRIORITY=low
done
