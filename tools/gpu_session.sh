#!/bin/bash
# GPU-box sessions of one round, one function per study (replaces the per-session scripts of rounds 3-4; their outputs are in profiles/).
#   gpurun --timeout N -- 'bash tools/gpu_session.sh <study> [args]'
# Every study writes under gpurun_out/<study>/ ; summaries worth keeping are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
STUDY=${1:?study name}; shift
OUT=gpurun_out/$STUDY; mkdir -p $OUT
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-parity --no-split-mode --no-extras --no-b256 --no-hourglass-train --no-accurate-mode"

line() {   # line <label> <bench args...> : one bench run, one summary line (images/s, ms, roofline.frac, step_mfma_frac)
  lab=$1; shift
  python bench.py $QUIET "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('$lab', d['value'], d['ms_per_step'], r.get('frac'), r.get('step_mfma_frac'))"
}

case $STUDY in
epi)      # round 5: accumulator orientation / epilogue form of the LDS-DMA GEMM (AWR_EPI = 0 | 1 | 2)
  timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward" 2>&1 | tail -3 | tee $OUT/ops.log
  for e in 0 1 2; do AWR_EPI=$e timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | tee $OUT/fwdset_epi$e.txt; done
  for e in 0 2; do AWR_EPI=$e timeout 300 python tools/microbench_gemm.py ksweep 2>&1 | grep -v amdgpu.ids | tee $OUT/ksweep_epi$e.txt; done
  for i in 1 2 3; do for e in 0 1 2; do
    AWR_EPI=$e line "r18 b64 epi$e" | tee -a $OUT/bench_ab.txt
    AWR_EPI=$e line "hg1 b64 epi$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_EPI=$e line "hg1 infer b128 epi$e" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
sdma)     # round 5: split-operand mode on LDS-DMA with a pre-cut activation image -- parity, isolated launches
  timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -k "split" 2>&1 | tail -25 | tee $OUT/ops.log
  timeout 900 python tools/microbench_gemm.py splitset 2>&1 | grep -v amdgpu.ids | tee $OUT/splitset.txt
  for i in 1 2; do for w in 0 1; do
    AWR_WGRAD_SPLIT=$w line "r18 b64 split-mode wgrad_split=$w" --gemm-products 6 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
elem)     # HBM-bound BatchNorm kernels against a plain copy at the sizes the plans run them at
  for sz in "262144 128" "262144 256" "262144 64" "65536 128" "65536 256" "1048576 64"; do
    echo "== npix C = $sz"; python tools/microbench_elem.py $sz 2>&1 | grep -v amdgpu.ids
  done | tee $OUT/elem.txt
  ;;
queues)   # HIP hardware queues: the runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues
  for i in 1 2; do for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q line "r18 b64 queues=$q" | tee -a $OUT/bench_ab.txt
    GPU_MAX_HW_QUEUES=$q line "hg1 b64 queues=$q" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  GPU_MAX_HW_QUEUES=8 python -c "
import sys; sys.path.insert(0,'.')
import torch, awr_amd
from awr_amd import _lib as L
import ctypes as C
torch.zeros(1).cuda()
n,k=C.c_int(),C.c_int(); L.call('awr_stream_pool_info', C.byref(n), C.byref(k)); print('GPU_MAX_HW_QUEUES=8: pool', n.value, 'independent', k.value)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_ab.txt
  ;;
pair)     # round 5: the fused inference pair (conv2 -> bn3 -> ReLU -> conv3 + skip) with its first GEMM on LDS-DMA staging
  timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short -k "pair or fused_conv or config3 or config_3" 2>&1 | tail -6 | tee $OUT/tests.log
  for i in 1 2 3; do for v in 0 1; do
    AWR_FUSE2_DMA=$v line "hg1 infer b128 pair_dma=$v" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  AWR_FUSE2_DMA=1 python bench.py --mode infer --net hourglass_1 --batch 128 --steps 10 --warmup 3 --per-layer $OUT/per_layer_hg1_infer_b128.txt > /dev/null 2>&1
  head -12 $OUT/per_layer_hg1_infer_b128.txt
  ;;
afflds)   # round 5: the fused input affine as an in-LDS pass (AWR_AFF_LDS=1, default) against the fragment-side form (0)
  timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward or prologue or affine" 2>&1 | tail -3 | tee $OUT/ops.log
  for e in 0 1; do AWR_AFF_LDS=$e timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | tee $OUT/fwdset_afflds$e.txt; done
  for i in 1 2 3; do for e in 0 1; do
    AWR_AFF_LDS=$e line "r18 b64 afflds$e" | tee -a $OUT/bench_ab.txt
    AWR_AFF_LDS=$e line "hg1 b64 afflds$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_AFF_LDS=$e line "hg1 infer b128 afflds$e" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  for e in 0 1; do AWR_AFF_LDS=$e python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 AWR_AFF_LDS=$e |" | tee -a $OUT/bench_ab.txt; done
  ;;
halfbnb)  # round 5: half-batch BatchNorm-backward wavefront (AWR_HALF_BNB_MIN_ROWS: 0 = off, default 32768 rows per half)
  timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "half_batch or side_streams or deterministic or bucketed" 2>&1 | tail -4 | tee $OUT/tests.log
  for i in 1 2 3; do for v in 0 32768 8192; do
    AWR_HALF_BNB_MIN_ROWS=$v line "r18 b64 half_min=$v" | tee -a $OUT/bench_ab.txt
    AWR_HALF_BNB_MIN_ROWS=$v line "hg1 b64 half_min=$v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  for v in 0 32768; do AWR_HALF_BNB_MIN_ROWS=$v line "r18 b256 half_min=$v" --batch 256 | tee -a $OUT/bench_ab.txt; done
  for v in 0 32768; do AWR_HALF_BNB_MIN_ROWS=$v python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 half_min=$v |" | tee -a $OUT/bench_ab.txt; done
  ;;
traffic)  # does HBM / fabric traffic cost the GEMM family its clock?  16-float stages (shipped) vs 32-float stages (study build:
          # hipcc -DAWR_DMA_STUDY -> awr-..._amd/lib_study/libawr_hip.so, AWR_DMA=1): PMC FETCH / WRITE per launch, MFMA busy, serial step time
  export AWR_LIB_PATH=$GRAFT_REPO_ROOT/awr-adaptive-weighting-regression_amd/lib_study/libawr_hip.so
  for m in 2 1; do
    for i in 1 2; do AWR_DMA=$m line "r18 b64 serial AWR_DMA=$m" --wgrad-streams 0 | tee -a $OUT/bench_ab.txt; AWR_DMA=$m line "r18 b64 AWR_DMA=$m" | tee -a $OUT/bench_ab.txt; done
    AWR_DMA=$m bash tools/gpu_pmc.sh traffic_dma$m > /dev/null 2>&1
    cp gpurun_out/pmc_summary_traffic_dma$m.json $OUT/
    rm -rf gpurun_out/pmc_sq_traffic_dma$m gpurun_out/pmc_fetch_traffic_dma$m gpurun_out/pmc_write_traffic_dma$m
    python - <<PY | tee -a $OUT/bench_ab.txt
import json
d = json.load(open("$OUT/pmc_summary_traffic_dma$m.json"))
for fam, keys in (("fwd/dgrad", ("conv_gemm_dma_kernel", "conv_gemm_kernel")), ("wgrad", ("conv_wgrad",))):
    ent = [v for k, v in d.items() if any(x in k for x in keys)]
    n = sum(v["launches"] for v in ent)
    if n:
        f = sum(v["launches"] * v["fetch_MB_per_launch_x2"] for v in ent) / n
        w = sum(v["launches"] * v.get("write_MB_per_launch", 0.0) for v in ent) / n
        b = sum(v["launches"] * v.get("mfma_busy_frac", 0.0) for v in ent) / n
        print("AWR_DMA=$m %-10s launch-weighted: fetch x2 %.1f MB + write %.1f MB = %.1f MB per launch, MFMA busy %.3f (%d launches)" % (fam, f, w, f + w, b, n))
PY
  done
  ;;
deep)     # round 5: deep pipeline (four stage buffers) for launches of at most two workgroups per CU (AWR_DEEP = 0 | 1)
  timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward or prologue or dgrad_with" 2>&1 | tail -3 | tee $OUT/ops.log
  timeout 600 python tools/microbench_gemm.py smallset 2>&1 | grep -v amdgpu.ids | tee $OUT/smallset.txt
  for i in 1 2 3; do for e in 0 1; do
    AWR_DEEP=$e line "r18 b64 deep$e" | tee -a $OUT/bench_ab.txt
    AWR_DEEP=$e line "hg1 b64 deep$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_DEEP=$e line "hg1 infer b128 deep$e" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
    AWR_DEEP=$e line "r18 infer b4 deep$e" --mode infer --batch 4 --steps 200 --warmup 20 | tee -a $OUT/bench_ab.txt
    AWR_DEEP=$e line "r18 train b4 deep$e" --batch 4 --steps 50 | tee -a $OUT/bench_ab.txt
    AWR_DEEP=$e line "r18 train b16 deep$e" --batch 16 --steps 50 | tee -a $OUT/bench_ab.txt
  done; done
  for e in 0 1; do AWR_DEEP=$e python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 AWR_DEEP=$e |" | tee -a $OUT/bench_ab.txt; done
  ;;
deepwgs)  # deep pipeline: up to how many workgroups per launch (AWR_DEEP_MAX_WGS; 0 = off)
  for i in 1 2; do for w in 0 128 256 384 512; do
    AWR_DEEP_MAX_WGS=$w line "r18 b64 maxwgs$w" | tee -a $OUT/bench_ab.txt
    AWR_DEEP_MAX_WGS=$w line "hg1 b64 maxwgs$w" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_DEEP_MAX_WGS=$w line "r18 train b4 maxwgs$w" --batch 4 --steps 50 | tee -a $OUT/bench_ab.txt
    AWR_DEEP_MAX_WGS=$w line "r18 train b16 maxwgs$w" --batch 16 --steps 50 | tee -a $OUT/bench_ab.txt
    AWR_DEEP_MAX_WGS=$w line "r18 infer b4 maxwgs$w" --mode infer --batch 4 --steps 200 --warmup 20 | tee -a $OUT/bench_ab.txt
    AWR_DEEP_MAX_WGS=$w line "hg1 train b16 maxwgs$w" --net hourglass_1 --batch 16 --steps 30 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
lazybnb)  # un-materialised BatchNorm backward inside the consuming data gradient, 1x1 convolutions only (AWR_LAZY_BNB=1), re-measured on the round-5 binary
  for i in 1 2 3; do for v in 0 1; do
    AWR_LAZY_BNB=$v line "r18 b64 lazy_bnb=$v" | tee -a $OUT/bench_ab.txt
    AWR_LAZY_BNB=$v line "hg1 b64 lazy_bnb=$v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  for v in 0 1; do AWR_LAZY_BNB=$v python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 lazy_bnb=$v |" | tee -a $OUT/bench_ab.txt; done
  ;;
poolstats) # round 5: max-pool / up-sampling add that accumulate the next BatchNorm's statistics (AWR_FUSED_POOL_STATS = 0 | 1)
  timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -q --tb=short -x -k "fused_statistics or maxpool or upsample or hourglass or golden or deterministic or yardstick" 2>&1 | tail -4 | tee $OUT/tests.log
  for i in 1 2 3; do for v in 0 1; do
    AWR_FUSED_POOL_STATS=$v line "hg1 b64 pool_stats=$v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  for i in 1 2; do for v in 0 1; do AWR_FUSED_POOL_STATS=$v python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 pool_stats=$v |" | tee -a $OUT/bench_ab.txt; done; done
  AWR_FUSED_POOL_STATS=1 python bench.py --steps 5 --warmup 2 $QUIET --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_hg1.txt > /dev/null 2>&1; tail -13 $OUT/per_layer_hg1.txt
  ;;
pairpool) # round 5: the 2x2 max-pool of a fused pair's output written by the pair itself (AWR_PAIR_POOL = 0 | 1)
  timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_full_size_gpu.py tests/test_abi.py -m gpu -q --tb=short -k "pair or fused_conv or config3 or config_3 or abi" 2>&1 | tail -6 | tee $OUT/tests.log
  for i in 1 2 3; do for v in 0 1; do
    AWR_PAIR_POOL=$v line "hg1 infer b128 pair_pool=$v" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  AWR_PAIR_POOL=1 python bench.py --mode infer --net hourglass_1 --batch 128 --steps 10 --warmup 3 --per-layer $OUT/per_layer_hg1_infer_b128.txt > /dev/null 2>&1
  head -14 $OUT/per_layer_hg1_infer_b128.txt
  ;;
deep1x1)  # round 5, second session: the deep pipeline for single-tap (1x1) launches of ANY size (AWR_DEEP_1X1 = 0 | 1) -- bytes in flight per CU against HBM latency
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward" 2>&1 | tail -3 | tee $OUT/ops.log
  for e in 0 1; do AWR_DEEP_1X1=$e timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | grep "1x1\|batch" | tee $OUT/fwdset_deep1x1_$e.txt; done
  for i in 1 2 3; do for e in 0 1; do
    AWR_DEEP_1X1=$e line "hg1 b64 deep1x1=$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_DEEP_1X1=$e line "r18 b64 deep1x1=$e" | tee -a $OUT/bench_ab.txt
    AWR_DEEP_1X1=$e line "hg1 infer b128 deep1x1=$e" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  for e in 0 1; do AWR_DEEP_1X1=$e python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 deep1x1=$e |" | tee -a $OUT/bench_ab.txt; done
  for e in 0 1; do AWR_DEEP_1X1=$e python bench.py --steps 5 --warmup 2 $QUIET --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_hg1_deep1x1_$e.txt > /dev/null 2>&1; done
  ;;
probe1x1) # round 5, second session: memory-system probes of the LDS-DMA GEMM on the 1x1 shapes (study builds variants/libawr_probe<v>.so, -DAWR_DMA_PROBE=v:
          # bit 0 = A-operand requests beyond the first stage go nowhere, bit 1 = the same for the weights, bit 2 = the epilogue's stores are dropped; results WRONG, timing only)
  for v in 0 1 2 3 4 7; do
    echo "== AWR_DMA_PROBE=$v"
    AWR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libawr_probe$v.so timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | grep "1x1\|hg 3x3\|layer1"
  done | tee $OUT/probe1x1.txt
  ;;
ntstore)  # round 5, second session: cache policy of the GEMM epilogues' output stores / operand loads (study builds variants/libawr_<v>.so: st2 = stores nt,
          # st3 = stores sc0 nt, st2ld2 = stores nt + epilogue operand loads nt) against the shipped library
  for v in shipped st2 st3 st2ld2; do
    echo "== $v"
    L=""; [ $v != shipped ] && L=$GRAFT_REPO_ROOT/variants/libawr_$v.so
    AWR_LIB_PATH=$L timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | grep "1x1\|hg 3x3\|layer1\|layer3\|deconv 256->256 @32"
  done | tee $OUT/fwdset.txt
  for i in 1 2 3; do for v in shipped st2 st3 st2ld2; do
    L=""; [ $v != shipped ] && L=$GRAFT_REPO_ROOT/variants/libawr_$v.so
    AWR_LIB_PATH=$L line "r18 b64 $v" | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "hg1 b64 $v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "hg1 infer b128 $v" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
ntpolicy) # round 5, second session: streaming output stores by size / K extent (awr_conv_args.out_nt = 0 -> AWR_NT_MIN_MB, AWR_NT_MAX_K; 0 MB = never)
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward or prologue or dgrad_with" 2>&1 | tail -3 | tee $OUT/ops.log
  for i in 1 2; do for cfg in "0 0" "64 100000" "128 100000" "256 100000" "128 512" "256 512" "500 100000"; do
    set -- $cfg
    export AWR_NT_MIN_MB=$1 AWR_NT_MAX_K=$2
    line "r18 b64 nt_min_mb=$1 max_k=$2" | tee -a $OUT/bench_ab.txt
    line "hg1 b64 nt_min_mb=$1 max_k=$2" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    line "hg1 infer b128 nt_min_mb=$1 max_k=$2" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
    line "r18 b256 nt_min_mb=$1 max_k=$2" --batch 256 --steps 8 | tee -a $OUT/bench_ab.txt
  done; done
  for cfg in "0 0" "128 100000" "256 512" "500 100000" "0 0" "128 100000"; do
    set -- $cfg
    AWR_NT_MIN_MB=$1 AWR_NT_MAX_K=$2 python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 nt_min_mb=$1 max_k=$2 |" | tee -a $OUT/bench_ab.txt
  done
  ;;
ntfinal)  # streaming-store rule as shipped (default) against off (AWR_NT_MIN_MB=0): the large-tensor shapes, three interleaved repetitions
  timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "streaming or batch_parts or bit_identical" 2>&1 | tail -3 | tee $OUT/ops.log
  for i in 1 2 3; do for v in 0 256; do
    AWR_NT_MIN_MB=$v python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 nt_min_mb=$v |" | tee -a $OUT/bench_ab.txt
    AWR_NT_MIN_MB=$v line "r18 b256 nt_min_mb=$v" --batch 256 --steps 8 | tee -a $OUT/bench_ab.txt
    AWR_NT_MIN_MB=$v line "r18 b64 nt_min_mb=$v" | tee -a $OUT/bench_ab.txt
    AWR_NT_MIN_MB=$v line "hg1 b64 nt_min_mb=$v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
faststats) # round 5, second session: BatchNorm statistics taken from the accumulators (EM 5, AWR_FAST_STATS = 0 | 1)
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "statistics or bit_identical or every_tile or conv_forward or prologue or streaming" 2>&1 | tail -3 | tee $OUT/ops.log
  for e in 0 1; do AWR_FAST_STATS=$e timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | tee $OUT/fwdset_faststats$e.txt; done
  for i in 1 2 3; do for e in 0 1; do
    AWR_FAST_STATS=$e line "r18 b64 fast_stats=$e" | tee -a $OUT/bench_ab.txt
    AWR_FAST_STATS=$e line "hg1 b64 fast_stats=$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_FAST_STATS=$e line "r18 b256 fast_stats=$e" --batch 256 --steps 8 | tee -a $OUT/bench_ab.txt
  done; done
  for e in 0 1 0 1; do AWR_FAST_STATS=$e python tools/check_hg2_256.py 128 2>&1 | tail -1 | sed "s|^|config5 fast_stats=$e |" | tee -a $OUT/bench_ab.txt; done
  timeout 1200 python -m pytest tests/test_nets_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -3 | tee $OUT/nets.log
  ;;
faststats2) # EM 5 second form (sums of the STORED value): golden / full-size suites under both settings, low-batch steps
  for e in 1 0; do echo "== AWR_FAST_STATS=$e"; AWR_FAST_STATS=$e timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^E        +" | tail -8; done | tee $OUT/suites.log
  timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "statistics" 2>&1 | tail -2 | tee $OUT/ops.log
  for i in 1 2 3; do for e in 0 1; do
    AWR_FAST_STATS=$e line "r18 b4 fast_stats=$e" --batch 4 | tee -a $OUT/bench_ab.txt
    AWR_FAST_STATS=$e line "r18 b16 fast_stats=$e" --batch 16 | tee -a $OUT/bench_ab.txt
    AWR_FAST_STATS=$e line "hg1 b16 fast_stats=$e" --net hourglass_1 --batch 16 | tee -a $OUT/bench_ab.txt
    AWR_FAST_STATS=$e line "hg1 b64 fast_stats=$e" --net hourglass_1 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
faststats3) # EM 5 third form (four short fp32 chains per tile, fp64 from there): the parity suites as shipped + accuracy sweep
  timeout 1200 python -m pytest tests/test_nets_gpu.py tests/test_full_size_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^E        +" | tail -30 | tee $OUT/suites.log
  timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "statistics" 2>&1 | tail -2 | tee $OUT/ops.log
  python tools/probes/stats_paths.py 2>&1 | grep -v amdgpu.ids | grep "tile=(0, 0)" | tee $OUT/stats_paths.txt
  ;;
epirows)  # epilogue rows restructured (one store-policy branch per tile, a row's operands requested together): in-tree library against variants/libawr_old.so
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "bit_identical or every_tile or conv_forward or prologue or dgrad_with or streaming or statistics or batch_parts" 2>&1 | tail -3 | tee $OUT/ops.log
  for i in 1 2 3; do for v in old new; do
    L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/variants/libawr_old.so
    AWR_LIB_PATH=$L line "r18 b64 $v" | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "hg1 b64 $v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "hg1 infer b128 $v" --net hourglass_1 --mode infer --batch 128 | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "r18 b16 $v" --batch 16 | tee -a $OUT/bench_ab.txt
    AWR_LIB_PATH=$L line "r18 b256 $v" --batch 256 --steps 8 | tee -a $OUT/bench_ab.txt
  done; done
  for v in old new; do L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/variants/libawr_old.so
    AWR_LIB_PATH=$L python bench.py --steps 5 --warmup 2 $QUIET --wgrad-streams 0 --per-layer $OUT/per_layer_r18_$v.txt > /dev/null 2>&1
    AWR_LIB_PATH=$L python bench.py --steps 5 --warmup 2 $QUIET --wgrad-streams 0 --net hourglass_1 --per-layer $OUT/per_layer_hg1_$v.txt > /dev/null 2>&1
  done
  ;;
evidence) # final evidence of the session: timelines of the final binary, PMC traffic of the Hourglass-1 step with the streaming-store rule on / off
  bash tools/gpu_trace.sh r05 --no-hourglass-train --no-accurate-mode > /dev/null 2>&1
  bash tools/gpu_trace.sh r05_hg1 --no-hourglass-train --no-accurate-mode --net hourglass_1 > /dev/null 2>&1
  cp gpurun_out/timeline_r05.txt gpurun_out/timeline_r05_hg1.txt $OUT/
  rm -f gpurun_out/kernel_trace_r05.csv gpurun_out/kernel_trace_r05_hg1.csv
  for v in 0 256; do
    AWR_NT_MIN_MB=$v EXTRA="--net hourglass_1" bash tools/gpu_pmc.sh hg1_nt$v > /dev/null 2>&1
    cp gpurun_out/pmc_summary_hg1_nt$v.json $OUT/
    rm -rf gpurun_out/pmc_sq_hg1_nt$v gpurun_out/pmc_fetch_hg1_nt$v gpurun_out/pmc_write_hg1_nt$v
  done
  ls -la $OUT
  ;;
kernarg)  # where the runtime keeps kernel arguments (the 620-byte awr_conv_args every wave s_loads in its prologue): HIP_FORCE_DEV_KERNARG = unset | 0 | 1
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    echo "== HIP_FORCE_DEV_KERNARG=$v"
    timeout 600 python tools/microbench_gemm.py fwdset 2>&1 | grep -v amdgpu.ids | grep "hg 1x1\|hg 3x3\|layer1"
  done | tee $OUT/fwdset.txt
  for i in 1 2 3; do for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    line "r18 b64 kernarg=$v" | tee -a $OUT/bench_ab.txt
    line "hg1 b64 kernarg=$v" --net hourglass_1 | tee -a $OUT/bench_ab.txt
    line "r18 b4 kernarg=$v" --batch 4 | tee -a $OUT/bench_ab.txt
    line "r18 infer b4 kernarg=$v" --mode infer --batch 4 | tee -a $OUT/bench_ab.txt
  done; done
  ;;
tests)    # the whole GPU suite
  timeout 1700 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E        +" | tail -15 | tee $OUT/tests.log
  ;;
*) echo "unknown study $STUDY"; exit 2;;
esac
