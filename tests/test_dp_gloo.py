"""CPU, world_size 2, gloo: the data-parallel plumbing of TrainEngine (bucketed SUM all-reduce of the
flat gradient arena, 1/world folded into the optimiser, rank-0 broadcast of parameters/buffers),
plus the batched evaluator against its golden vector."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import awr_amd  # noqa: F401
    from awr_amd.trainer import GradSync
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sync = GradSync(n, torch.distributed.group.WORLD, n_buckets=4, min_bucket=1000)
        assert sync.world == world and abs(sync.grad_scale - 1.0 / world) < 1e-12
        # buckets tile [0, n) exactly, 16-byte aligned interior edges
        assert sync.buckets[0][0] == 0 and sync.buckets[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(sync.buckets, sync.buckets[1:]))
        assert all(lo % 4 == 0 for lo, _ in sync.buckets)
        g = torch.Generator().manual_seed(100 + rank)
        params = torch.randn(n, generator=g)
        bufs = torch.randn(37, generator=g)
        sync.broadcast(params, bufs)
        grads = torch.randn(n, generator=g)
        mine = grads.clone()
        sync.allreduce(grads)
        # epoch metric that drives the LR scheduler: rank-local error sums / frame counts -> the same global mean everywhere
        gm = sync.global_mean(10.0 * (rank + 1), 4 + rank)
        assert abs(gm - 30.0 / 9.0) < 1e-12, gm
        q.put((rank, params[:5].tolist(), bufs[:3].tolist(), mine.double().sum().item(), grads.double().sum().item(),
               (grads * sync.grad_scale)[:4].tolist()))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_two_ranks_gloo():
    world, n = 2, 10007
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, p0, b0, s0, t0, a0), (r1, p1, b1, s1, t1, a1) = res
    assert p0 == p1 and b0 == b1                       # rank 0's parameters / BN buffers everywhere
    assert abs(t0 - (s0 + s1)) < 1e-3 and abs(t0 - t1) < 1e-6      # SUM all-reduce, identical on both ranks
    assert a0 == a1                                    # averaged gradient == what the optimiser kernel consumes


def test_single_process_is_a_noop():
    import awr_amd  # noqa: F401
    from awr_amd.trainer import GradSync
    s = GradSync(5000)
    g = torch.arange(5000.0)
    s.allreduce(g)
    s.broadcast(g)
    assert s.global_mean(6.0, 4) == 1.5 and s.global_mean(0.0, 0) != s.global_mean(0.0, 0)      # NaN for an empty epoch
    assert s.world == 1 and s.grad_scale == 1.0 and torch.equal(g, torch.arange(5000.0))


def test_evaluator_matches_golden(golden_dir):
    import numpy as np
    import awr_amd  # noqa: F401
    from awr_amd.evaluator import EvalUtil
    g = np.load(os.path.join(golden_dir, "eval_feed.npz"))
    ev = EvalUtil(128, (588.03, 587.07, 320.0, 240.0), -1, 14)
    ev.feed_batch(g["jt_uvd"][:8], g["jt_xyz_gt"][:8], g["center"][:8], g["M"][:8], g["cube"][:8])
    for i in range(8, 16):
        ev.feed(g["jt_uvd"][i], g["jt_xyz_gt"][i], g["center"][i], g["M"][i], g["cube"][i])
    mpe, med, auc, pck, th = ev.get_measures()
    assert abs(mpe - float(g["mpe"])) < 1e-3 and abs(auc - float(g["auc"])) < 1e-6 and abs(med - float(g["med"])) < 1e-3
    np.testing.assert_allclose(pck, g["pck"], atol=1e-9)
    np.testing.assert_allclose(np.array(ev.jt_uvd_pred), g["uvd"], atol=2e-3)
