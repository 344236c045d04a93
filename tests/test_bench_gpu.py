"""GPU: bench.py honours the driver's contract -- one JSON line with the required fields, also when launched
through torch.distributed.run (the N > 1 launch form) with the data-parallel code path forced on a 1-rank group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline"], cwd=REPO,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["algorithmic_gflop_per_image"] - 22.603) < 0.01                      # 22.60 GFLOP / image
    assert d["joint_err_mm_vs_oracle"]["mean"] < 1e-3                                 # north_star: 1e-3 mm mean
    # round 3: what ran, the HBM-bound kernels' roofline block, config 4's per-GPU shape beside the headline
    assert len(d["ranks_seen"]) == 1 and d["ranks_seen"][0]["rank"] == 0 and d["dist_backend"] is None
    for k, v in d["roofline_hbm"].items():
        assert v["bytes_algorithmic"] > 0 and v["avg_us"] > 0 and abs(v["frac_of_8TBps"] - v["gbps"] / 8000.0) < 2e-3, (k, v)
    assert "adam_step" in d["roofline_hbm"] and ("head_loss_step_nhwc" in d["roofline_hbm"] or "dense_loss" in d["roofline_hbm"])
    assert d["b256"]["n_gpus"] == 1 and abs(d["b256"]["value"] - 256 * 1e3 / d["b256"]["ms_per_step"]) < 1e-2 * d["b256"]["value"]
    assert all(v["gbps"] > 0 for v in d["b256"]["roofline_hbm"].values()) and len(d["b256"]["roofline_hbm"]) >= 2      # round 6: the head kernels where they have work
    # round 5: the parity mode (blocked accumulation) in the driver-run line -- the headline's step, the scoring pass, the joint error in that mode
    am = d["accurate_mode"]
    assert am["train"]["value"] > 0 and 0.5 < am["train"]["vs_headline"] < 1.2 and am["infer_b128"]["value"] > 0
    assert am["joint_err_mm_vs_oracle"]["mean"] < 1e-3
    # round 6: the opt-in Winograd forward mode with BOTH roofline fractions (batch 8 has few eligible launches: the record must still be there)
    wm = d["winograd_mode"]["train"]
    assert wm["value"] > 0 and 0.5 < wm["vs_headline"] < 1.5
    if wm.get("winograd_launches"):
        assert wm["mfma_flops_per_algorithmic_flop"] < 1.0 and wm["step_mfma_frac_executed"] < wm["step_mfma_frac"]


def test_bench_under_torchrun_with_forced_dp_path():
    env = dict(os.environ, AWR_FORCE_DP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline",
           "--no-extras", "--no-winograd", "--no-accurate-mode", "--no-split-mode"]      # (those sub-records are test_bench_single_process_line's)
    out = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp1" and d["value"] > 0
    # a 1-rank RCCL group runs the transport A/B for real: the library's own communicator exchanges the buckets, replicas (one) agree
    nr = d["native_rccl"]
    assert "error" not in nr and nr["value"] > 0 and nr["dp_selftest"]["replicas_bitwise_equal_after_steps"] is True, nr


@pytest.mark.timeout(1200)
def test_bench_two_ranks_exactly_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`: two real ranks (sharing GPU 0, gloo instead of
    RCCL -- the only difference to the 8-GPU run): one JSON line from rank 0, whole-job throughput, weak scaling."""
    env = dict(os.environ, AWR_DIST_BACKEND="gloo", AWR_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-b256"]
    out = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=1100, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = _last_json(out.stdout)
    _check_two_rank_line(d)
    assert d["config"]["launcher"] == "torch.distributed.run"


def _check_two_rank_line(d):
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "cpu_baseline" not in d and "split_mode" not in d            # rank-0-at-N=1 extras only
    # the line proves what ran: backend, one entry per rank (rank, device, pid), replicas bitwise equal after the steps, bucket timeline
    assert d["dist_backend"] == "gloo" and sorted(r["rank"] for r in d["ranks_seen"]) == [0, 1]
    assert len({r["pid"] for r in d["ranks_seen"]}) == 2
    st = d["dp_selftest"]
    assert st["replicas_bitwise_equal_after_steps"] is True and st["steps_checked"] == 3
    tl = st["bucket_timeline_rank0"]
    assert len(tl["buckets"]) >= 2 and all(b["end_ms"] >= b["start_ms"] for b in tl["buckets"]) and tl["backward_end_ms"] > 0


@pytest.mark.timeout(1200)
def test_bench_gpus_2_spawns_its_own_ranks():
    """a plain `python3 bench.py --gpus 2` (no torchrun) used to be a SystemExit.  It now starts its rank processes
    itself; here two ranks share GPU 0 over gloo (the 1-GPU box), everything else is the 8-GPU launch."""
    env = dict(os.environ, AWR_DIST_BACKEND="gloo", AWR_FORCE_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--dp-selftest"]
    out = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=1100, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = _last_json(out.stdout)
    _check_two_rank_line(d)
    assert d["config"]["launcher"].startswith("bench.py")
    # round 5, what the first 8-GPU run has to answer in one go: (i) config 4's per-GPU shape (256 / GPU) with the replicas checked after ITS
    # steps, (ii) the transport A/B (library-owned RCCL communicator; over gloo the record says why it was skipped), (iii) every rank pinned
    # to CPUs (its GPU's NUMA node, or an equal share)
    b = d["b256"]
    assert b["n_gpus"] == 2 and b["dp_selftest"]["replicas_bitwise_equal_after_steps"] is True
    assert "native_rccl" in d and ("skipped" in d["native_rccl"] or d["native_rccl"]["value"] > 0)
    for r in d["ranks_seen"]:
        assert "cpu_affinity" in r and (r["cpu_affinity"].get("n_cpus", 0) >= 1 or "error" in r["cpu_affinity"]), r
