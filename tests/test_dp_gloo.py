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


def _loader_worker(rank, world, port, q):
    """one rank of a 2-rank job iterating the parameter-block dataset the way Trainer._loader does (DistributedSampler, same epoch seed)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import awr_amd  # noqa: F401
    from awr_amd import nyu_device as DV
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(4)
        n = 22
        centers = np.stack([rng.uniform(-100, 100, n), rng.uniform(-80, 80, n), rng.uniform(600, 900, n)], 1)
        labels = centers[:, None, :] + rng.uniform(-60, 60, (n, 14, 3))
        data = DV.DeviceNYU.from_arrays((n, 480, 640), labels, centers, "test", img_size=128)
        sampler = torch.utils.data.distributed.DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, drop_last=False)
        sampler.set_epoch(3)
        seen = []
        for blocks, jt_xyz, jt_uvd, center, M, cube in torch.utils.data.DataLoader(data, batch_size=4, sampler=sampler, num_workers=0):
            assert blocks.dtype == torch.uint8 and blocks.shape[1] == DV.BLOCK_BYTES
            for b in blocks:
                blk = awr_amd._lib.NyuSample.from_buffer_copy(bytes(b.numpy()))
                seen.append(int(blk.frame))
        # every rank steps the same number of batches (the gradient all-reduce of the step needs that)
        counts = [None] * world
        torch.distributed.all_gather_object(counts, len(seen))
        q.put((rank, seen, counts))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_device_loader_blocks_shard_over_two_ranks_gloo():
    """Data parallel + device data path: each rank draws its 1 / world shard of the epoch's permutation as parameter blocks (every rank holds the whole
    frame store, a block addresses its frame by index); together the ranks cover the dataset, with equal batch counts."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loader_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, c0), (r1, s1, c1) = res
    assert c0 == c1 == [11, 11]
    assert sorted(set(s0) | set(s1)) == list(range(22)) and not (set(s0) & set(s1))
