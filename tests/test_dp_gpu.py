"""GPU: the complete data-parallel TrainEngine path with TWO real ranks (both on GPU 0, gloo process group over CUDA tensors):
what the 8-GPU RCCL run does, minus the transport."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, backend="gloo", native=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(str(tmp_path), "dp")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AWR_TEST_BACKEND=backend,
                   AWR_DETERMINISTIC="1", AWR_TEST_NATIVE="1" if native else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [torch.load("%s.rank%d" % (out, r)) for r in range(2)]


@pytest.mark.timeout(1200)
def test_two_rank_data_parallel_engine(tmp_path):
    """(a) Both ranks feed the SAME shard: the averaged gradient equals each rank's own, so two data-parallel steps must land on
    the parameters of a single-process run from rank 0's initial weights (the broadcast replaced rank 1's).  (b) Different shards
    per rank: losses differ, parameters do not (every rank applies the same averaged gradient), BN running statistics stay
    rank-local (what stock DDP does)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import awr_amd
    import awr_oracle as O
    from awr_amd.trainer import TrainEngine
    r0, r1 = _run(tmp_path)
    a0, a1 = r0["same"], r1["same"]
    assert torch.equal(a0["params"], a1["params"]) and torch.equal(a0["buffers"], a1["buffers"])
    awr_amd.set_deterministic(True)                  # the workers ran with AWR_DETERMINISTIC=1: the reference run must match them bitwise
    torch.manual_seed(1234)
    net = awr_amd.get_deconv_net(18, 14, 2).cuda()
    eng = TrainEngine(net, 2, 128, 1.0, coord_weight=1.0, lr=1e-3, use_graph=False, autotune=False)
    ref_losses = []
    for s in range(2):
        img, jt = O.synth_batch(2, 128, 14, seed=70 + s)
        ref_losses.append(float(eng.step(img.cuda(), jt.cuda())[0][2]))
    ref = net.flat_params()[:net.n_active].cpu()
    awr_amd.set_deterministic(False)
    # both ranks fed the same shard: (g + g) / 2 == g exactly, so two data-parallel steps ARE two single-process steps, bit for bit
    assert a0["losses"] == ref_losses
    assert torch.equal(a0["params"], ref)
    b0, b1 = r0["split"], r1["split"]
    assert b0["losses"] != b1["losses"]
    assert torch.equal(b0["params"], b1["params"])
    for mode in ("same", "split"):                    # bitwise-equal replicas after every single step
        for a, b in zip(r0[mode]["per_step"], r1[mode]["per_step"]):
            assert torch.equal(a, b), mode
    assert not torch.equal(b0["buffers"], b1["buffers"])
    # sharded Trainer.test (10 frames, batch 4: a ragged last batch, 3 batches over 2 ranks): both ranks report the same mpe, equal to
    # the single-process evaluation of the same (broadcast) weights
    from awr_amd.config import Config
    from awr_amd.trainer import SyntheticHands, Trainer
    assert r0["test_mpe"] == r1["test_mpe"] and torch.equal(r0["test_params"], r1["test_params"])

    class Cfg(Config):
        net, kernel_size, batch_size, num_workers, max_epoch, output_dir, load_model, exp_id, use_hipgraph, vis_freq = \
            "resnet_18", 1.0, 4, 0, 1, str(tmp_path), "", "single", False, 0
    tr = Trainer(Cfg(), None, SyntheticHands(10, seed=2))
    tr.net.flat_params()[:tr.net.n_active].copy_(r0["test_params"].cuda())
    tr.net.weights_changed()
    assert abs(tr.test(1) - r0["test_mpe"]) < 1e-4


@pytest.mark.timeout(1200)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the 8-GPU box lights this up)")
def test_two_rank_data_parallel_engine_over_rccl(tmp_path):
    """The same two-rank run over the real transport: backend "nccl" (= RCCL over xGMI), one GPU per rank.  Replicas must hold
    bitwise-equal parameters after EVERY step (same all-reduced gradient, same fused optimiser kernel), whatever their shards."""
    r0, r1 = _run(tmp_path, backend="nccl")
    for mode in ("same", "split"):
        assert len(r0[mode]["per_step"]) == len(r1[mode]["per_step"]) >= 1
        for a, b in zip(r0[mode]["per_step"], r1[mode]["per_step"]):
            assert torch.equal(a, b), mode
    assert torch.equal(r0["same"]["buffers"], r1["same"]["buffers"])
    assert r0["split"]["losses"] != r1["split"]["losses"] and not torch.equal(r0["split"]["buffers"], r1["split"]["buffers"])


@pytest.mark.timeout(1200)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one RCCL rank per device)")
def test_two_rank_data_parallel_engine_over_the_librarys_own_rccl_communicator(tmp_path):
    """The same run with the bucket exchange issued natively (awr_dp_*: librccl.so through dlopen, awr_plan_set_dp): replicas bitwise
    equal after every step, and bitwise equal to the torch.distributed transport (same buckets, same SUM, deterministic mode)."""
    r0, r1 = _run(tmp_path, backend="nccl", native=True)
    for mode in ("same", "split"):
        for a, b in zip(r0[mode]["per_step"], r1[mode]["per_step"]):
            assert torch.equal(a, b), mode
    t0, _ = _run(tmp_path, backend="nccl", native=False)
    assert torch.equal(r0["same"]["params"], t0["same"]["params"])
