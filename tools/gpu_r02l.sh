#!/bin/bash
TAG=r02l
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export AWR_TUNE_CACHE=$OUT/tune_cache_$TAG.json
timeout 900 python -m pytest tests/test_net_abi_gpu.py tests/test_abi.py -q --tb=short -x 2>&1 | tail -5
B="python bench.py --no-split-mode --no-extras --no-cpu-baseline --no-parity"
show() { python -c "
import json,sys; d=json.loads(open('$1').read()); print('$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_mfma_frac'], d['roofline']['gemm_seconds_per_step'])"; }
$B --steps 30 --warmup 5 > $OUT/b_${TAG}_b64.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b64.json r18_b64
$B --steps 30 --warmup 5 --graph > $OUT/b_${TAG}_b64_graph.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b64_graph.json r18_b64_graph
$B --steps 30 --warmup 5 --wgrad-streams 0 > $OUT/b_${TAG}_b64_serial.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b64_serial.json r18_b64_serial
$B --steps 30 --warmup 5 --deterministic > $OUT/b_${TAG}_b64_det.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b64_det.json r18_b64_det
$B --steps 10 --warmup 3 --batch 256 > $OUT/b_${TAG}_b256.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b256.json r18_b256
$B --steps 10 --warmup 3 --net hourglass_1 > $OUT/b_${TAG}_hg1.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_hg1.json hg1_b64
$B --steps 10 --warmup 3 --batch 16 > $OUT/b_${TAG}_b16.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b16.json r18_b16
$B --steps 10 --warmup 3 --batch 4 > $OUT/b_${TAG}_b4.json 2>> $OUT/bench_$TAG.err; show $OUT/b_${TAG}_b4.json r18_b4
for net in resnet_18 hourglass_1; do for b in 4 64 128; do
python bench.py --mode infer --batch $b --steps 20 --warmup 3 --net $net 2>> $OUT/bench_$TAG.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('infer $net b$b', d['value'], d['ms_per_step'], d['mfma_frac'])"
done; done
python tools/check_hg2_256.py 2 128 2>&1 | tail -4
python tools/cpu_issue_time.py 2>&1 | tail -5
