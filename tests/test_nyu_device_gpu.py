"""GPU: the device-side NYU data path (csrc/awr_nyu.hip through the C ABI) against the host loader awr_amd.nyu_data -- the numpy
restatement of dataloader/loader.py:19-179 / nyu_loader.py:38-90 that tests/test_nyu_data_cpu.py pins to the reference-generated
vectors.  Integer / fixed-point / IEEE arithmetic: every comparison is torch.equal (bit-exact), no tolerance anywhere."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import awr_amd  # noqa: F401
    from awr_amd import nyu_data, nyu_device
    return nyu_data, nyu_device


def _frames(n, rng, dtype=np.uint16):
    """synthetic 480 x 640 depth frames: far wall, a tilted hand-sized disc, sensor holes (zeros)"""
    yy, xx = np.mgrid[0:480, 0:640]
    out, centers = [], []
    for _ in range(n):
        c = np.array([rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(550, 1000)])
        u, v = 588.03 * c[0] / c[2] + 320.0, -587.07 * c[1] / c[2] + 240.0
        d = np.full((480, 640), 1500.0 + rng.uniform(-200, 200))
        hand = (xx - u) ** 2 + (yy - v) ** 2 < (70 * 750.0 / c[2]) ** 2
        d[hand] = c[2] + 0.3 * (xx[hand] - u) - 0.2 * (yy[hand] - v) + rng.uniform(-3, 3, int(hand.sum()))
        d[rng.rand(480, 640) < 0.05] = 0
        out.append(np.round(d) if dtype == np.uint16 else d)
        centers.append(c)
    return np.stack(out).astype(dtype), np.array(centers)


def _host_sample(ND, aug, frame_f32, c_uvd, cube, jt, op, trans, scale, rot):
    img, M = ND.crop(frame_f32.copy(), c_uvd, cube, np.array([128, 128]))
    out = aug.augment(img.copy(), jt.copy(), c_uvd.copy(), cube.copy(), M.copy(), op, trans, scale, rot)
    return img, out


def test_crop_matches_loader_crop_on_the_reference_fixture_inputs(mods, golden_dir):
    """loader_fns.npz's frame (float32, non-integer depths, 30 % holes) and its six centres: crop + its two reductions"""
    ND, DV = mods
    from awr_amd import _lib as L
    g = np.load(os.path.join(golden_dir, "loader_fns.npz"))
    rng = np.random.RandomState(int(g["seed"]))
    centers_xyz = np.stack([rng.uniform(-200, 200, 6), rng.uniform(-150, 150, 6), rng.uniform(500, 1100, 6)], 1)
    cube = np.array([300.0, 300.0, 300.0])
    depth = rng.uniform(400, 1300, (480, 640)).astype(np.float32)
    depth[rng.rand(480, 640) < 0.3] = 0
    store = DV.FrameStore(depth[None])
    blocks, want, Ms = [], [], []
    for c in centers_xyz:
        cuvd = ND.xyz2uvd(c, ND.PARAS, -1).astype(np.float64)
        b = L.NyuSample()
        Ms.append(DV.set_crop(b, 0, cuvd, cube, np.array([128, 128]), ND.PARAS, 480, 640))
        DV.set_normalize(b, c, cube)
        blocks.append(b)
        img, M = ND.crop(depth.copy(), cuvd, cube, np.array([128, 128]))
        assert np.array_equal(M, Ms[-1])
        want.append(img)
    crop, stats = DV.crop_batch(store, DV.blocks_to_tensor(blocks), 128)
    want = np.stack(want)
    assert torch.equal(crop.cpu(), torch.from_numpy(want))
    assert np.array_equal(stats[:, 0].cpu().numpy(), want.reshape(6, -1).max(1))
    assert np.array_equal(stats[:, 1].cpu().numpy(), np.array([w[w > 0].min() for w in want]))
    # test-time path (nyu_loader.py:59-60): crop + normalize in the one-launch form
    out = DV.Renderer(store, 128, 8)(DV.blocks_to_tensor(blocks))
    ref = np.stack([ND.normalize(w.max(), w.copy(), c, cube).astype(np.float32) for w, c in zip(want, centers_xyz)])
    assert torch.equal(out.cpu()[:, 0], torch.from_numpy(ref))


def test_window_partly_outside_the_frame_is_zero_padded(mods):
    ND, DV = mods
    from awr_amd import _lib as L
    rng = np.random.RandomState(11)
    depth = rng.uniform(500, 900, (480, 640)).astype(np.float32)
    store = DV.FrameStore(depth[None])
    cube = np.array([300.0, 300.0, 300.0])
    blocks, want = [], []
    for uvd in ([5.0, 8.0, 600.0], [636.0, 470.0, 700.0], [320.0, -20.0, 500.0], [700.0, 240.0, 450.0], [30.0, 460.0, 1200.0]):
        c = np.array(uvd)
        b = L.NyuSample()
        DV.set_crop(b, 0, c, cube, np.array([128, 128]), ND.PARAS, 480, 640)
        blocks.append(b)
        want.append(ND.crop(depth.copy(), c, cube, np.array([128, 128]))[0])
    crop, _ = DV.crop_batch(store, DV.blocks_to_tensor(blocks), 128)
    assert torch.equal(crop.cpu(), torch.from_numpy(np.stack(want)))
    with pytest.raises(ValueError):
        DV.set_crop(L.NyuSample(), 0, np.array([2000.0, 240.0, 600.0]), cube, np.array([128, 128]), ND.PARAS, 480, 640)


def test_resamplers_match_the_fixed_point_restatement(mods):
    """awr_nyu_warp vs nyu_data.warp_affine / warp_perspective: rotations, similarity + projective maps, borders, odd sizes"""
    ND, DV = mods
    rng = np.random.RandomState(7)
    for trial in range(40):
        sh, sw = int(rng.randint(8, 140)), int(rng.randint(8, 140))
        dh, dw = (sh, sw) if trial % 2 else (int(rng.randint(8, 140)), int(rng.randint(8, 140)))
        img = (rng.rand(sh, sw) * 400 + 300).astype(np.float32)
        img[rng.rand(sh, sw) < 0.2] = 0
        border = float(rng.choice([0.0, 2.5, 300.0]))
        M = ND.rotation_matrix_2d((sw / 2 + rng.uniform(-3, 3), sh / 2 + rng.uniform(-3, 3)), rng.uniform(-180, 180), rng.uniform(0.7, 1.4))
        M[:, 2] += rng.uniform(-10, 10, 2)
        src = torch.from_numpy(img).cuda()[None]
        got = DV.warp(src, np.concatenate([ND._invert_affine(M).ravel(), [0, 0, 1]])[None], DV.OP_AFFINE, (dw, dh), border)
        assert torch.equal(got.cpu()[0], torch.from_numpy(ND.warp_affine(img, M, (dw, dh), border))), ("affine", trial)
        Hm = np.vstack([M, [rng.uniform(-3e-4, 3e-4), rng.uniform(-3e-4, 3e-4), 1.0]])
        got = DV.warp(src, np.linalg.inv(Hm).ravel()[None], DV.OP_PERSPECTIVE, (dw, dh), border)
        assert torch.equal(got.cpu()[0], torch.from_numpy(ND.warp_perspective(img, Hm, (dw, dh), border))), ("perspective", trial)
    # a singular / far-out map: W == 0 rows and coordinates that clip at +-2^31 behave like the restatement
    img = (rng.rand(16, 16) * 100).astype(np.float32)
    iH = np.array([[1e9, 0, 0], [0, 1.0, 0], [0, 0.5, -2.0]])
    got = DV.warp(torch.from_numpy(img).cuda()[None], iH.ravel()[None], DV.OP_PERSPECTIVE, (16, 16), 7.0)
    assert torch.equal(got.cpu()[0], torch.from_numpy(ND.warp_perspective(img, np.linalg.inv(iH), (16, 16), 7.0)))


def test_normalize_matches_in_both_promotions(mods):
    ND, DV = mods
    from awr_amd import _lib as L
    rng = np.random.RandomState(5)
    img = rng.uniform(300, 1100, (4, 96, 96)).astype(np.float32)
    img[rng.rand(4, 96, 96) < 0.2] = 0
    img[:, 0, :5] = img.reshape(4, -1).max(1)[:, None]
    blocks, want = [], []
    centers = [np.array([10.0, 20.0, 700.0]), np.array([10.0, 20.0, 650.5], np.float32), np.array([0.0, 0.0, 801.3]), np.array([1.0, 2.0, 612.25], np.float32)]
    cubes = [np.array([300.0, 300.0, 300.0]), np.array([300.0, 300.0, 317.3]), np.array([250.0, 250.0, 250.0], np.float32), np.array([250.0, 250.0, 263.1], np.float32)]
    for i in range(4):
        b = L.NyuSample()
        DV.set_normalize(b, centers[i], cubes[i])
        blocks.append(b)
        want.append(np.asarray(ND.normalize(img[i].max(), img[i].copy(), centers[i], cubes[i])).astype(np.float32))
    assert [b.norm32 for b in blocks] == [0, 0, 0, 1]
    got = DV.normalize(torch.from_numpy(img).cuda(), torch.from_numpy(img.reshape(4, -1).max(1)).cuda(), DV.blocks_to_tensor(blocks))
    assert torch.equal(got.cpu(), torch.from_numpy(np.stack(want)))


def test_augmentation_replays_the_reference_fixture(mods, golden_dir):
    """The inputs of tests/golden/loader_aug.npz (the reference's own random_aug / augment run on them, test_nyu_data_cpu.py): labels,
    cube, centre and matrix from the device path's host half equal the golden values; the image equals the host restatement's."""
    ND, DV = mods
    from awr_amd import _lib as L
    g = np.load(os.path.join(golden_dir, "loader_aug.npz"))
    aug, paug = ND.Augmenter(ND.PARAS, -1), DV.ParamAugmenter(ND.PARAS, -1)
    rng = np.random.RandomState(int(g["seed"]))
    yy, xx = np.mgrid[0:480, 0:640]
    frames, blocks, want = [], [], []
    for i, d in enumerate(g["draws"]):
        op, trans, scale, rot = aug.random_aug(10, 0.1, 180)
        op2, trans2, scale2, rot2 = paug.random_aug(10, 0.1, 180)                               # the two streams stay aligned
        assert op2 == op and scale2 == scale and rot2 == rot and np.array_equal(trans, trans2)
        c_xyz = np.array([rng.uniform(-120, 120), rng.uniform(-90, 90), rng.uniform(600, 900)])
        c_uvd = ND.xyz2uvd(c_xyz, ND.PARAS, -1).astype(np.float64)
        depth = np.full((480, 640), 1400.0, np.float32)
        hand = (xx - c_uvd[0]) ** 2 + (yy - c_uvd[1]) ** 2 < (60 * 750.0 / c_xyz[2]) ** 2
        depth[hand] = (c_xyz[2] + 0.25 * (xx[hand] - c_uvd[0]) - 0.15 * (yy[hand] - c_uvd[1])).astype(np.float32)
        cube = np.array([300.0, 300.0, 300.0])
        jt = rng.uniform(-100, 100, (14, 3))
        _, out = _host_sample(ND, aug, depth, c_uvd, cube, jt, op, trans, scale, rot)
        b = L.NyuSample()
        M = DV.set_crop(b, i, c_uvd, cube, np.array([128, 128]), ND.PARAS, 480, 640)
        paug.begin(b)
        pout = paug.augment(DV._Deferred((128, 128)), jt.copy(), c_uvd.copy(), cube.copy(), M.copy(), op, trans, scale, rot)
        flat = np.concatenate([np.asarray(pout[k], np.float64).ravel() for k in (1, 2, 3, 4)])
        np.testing.assert_array_equal(flat, g["case%d" % i])
        frames.append(depth)
        blocks.append(b)
        want.append(np.asarray(out[0]).astype(np.float32))
    store = DV.FrameStore(np.stack(frames))
    got = DV.Renderer(store, 128, len(blocks))(DV.blocks_to_tensor(blocks))
    assert torch.equal(got.cpu()[:, 0], torch.from_numpy(np.stack(want)))
    assert {b.op for b in blocks} == {0, 1, 2}


def test_thousand_random_draws_are_bit_identical(mods):
    """>= 1 000 random (op, trans, scale, rot) draws over uint16 frames -- the production frame type -- including no-op draws, centres
    near the frame border and zero-depth centres (loader.py:113, :172: those skip the recrop); one-launch LDS form AND the two-kernel
    form (awr_nyu_crop + awr_nyu_augment) against Augmenter.augment."""
    ND, DV = mods
    from awr_amd import _lib as L
    rng = np.random.RandomState(2024)
    frames, centers = _frames(12, rng)
    store = DV.FrameStore(frames)
    aug, paug = ND.Augmenter(ND.PARAS, -1), DV.ParamAugmenter(ND.PARAS, -1)
    f32 = frames.astype(np.float32)
    N, blocks, want, labels = 1024, [], [], 0
    ops = {}
    for i in range(N):
        k = int(rng.randint(0, 12))
        c_uvd = ND.xyz2uvd(centers[k] + rng.uniform(-40, 40, 3), ND.PARAS, -1).astype(np.float64)
        cube = np.array([300.0, 300.0, 300.0]) * (5.0 / 6.0 if i % 7 == 0 else 1.0)
        jt = rng.uniform(-100, 100, (14, 3))
        op, trans, scale, rot = aug.random_aug(*((10, 0.1, 180) if i % 2 else (None, None, None)))
        if i % 97 == 0:
            trans = np.zeros(3)
        if i % 89 == 0:
            scale, rot = 1.0, 0.0
        img, M = ND.crop(f32[k].copy(), c_uvd, cube, np.array([128, 128]))
        c_aug = c_uvd.copy()
        if i % 53 == 0:
            c_aug[2] = 0.0                                  # a zero-depth centre handed to augment
        if c_aug[2] == 0.0 and op is None:
            op = "trans"
        if c_aug[2] == 0.0 and op == "rot":
            op = "scale"                                     # (uvd2xyz of a zero-depth centre is fine for trans / scale)
        out = aug.augment(img.copy(), jt.copy(), c_aug.copy(), cube.copy(), M.copy(), op, trans, scale, rot)
        b = L.NyuSample()
        M2 = DV.set_crop(b, k, c_uvd, cube, np.array([128, 128]), ND.PARAS, 480, 640)
        paug.begin(b)
        pout = paug.augment(DV._Deferred((128, 128)), jt.copy(), c_aug.copy(), cube.copy(), M2.copy(), op, trans, scale, rot)
        for a, p in zip(out[1:], pout[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(p)) and np.asarray(a).dtype == np.asarray(p).dtype
            labels += 1
        ops[(op, b.op)] = ops.get((op, b.op), 0) + 1
        blocks.append(b)
        want.append(np.asarray(out[0]).astype(np.float32))
    want = torch.from_numpy(np.stack(want))
    bt = DV.blocks_to_tensor(blocks)
    render = DV.Renderer(store, 128, 256)
    got = torch.cat([render(bt[i:i + 256]).cpu() for i in range(0, N, 256)])[:, 0]
    bad = (got != want).flatten(1).any(1).nonzero().flatten().tolist()
    assert not bad, ("one-launch form differs", bad[:8], [blocks[j].op for j in bad[:8]])
    render.check()
    crop, stats = DV.crop_batch(store, bt, 128)
    two, status = DV.augment_batch(crop, stats, bt)
    assert torch.equal(two.cpu()[:, 0], want) and int(status.sum()) == 0
    assert labels == 4 * N
    assert min(ops.get(k, 0) for k in (("trans", 1), ("scale", 1), ("rot", 2), (None, 0))) > 100, ops
    assert ops.get(("trans", 0), 0) + ops.get(("scale", 0), 0) >= 10, ops          # no-op draws and zero-depth centres took the skip branch


def test_large_crops_go_through_scratch(mods):
    """dsize = 256 (BASELINE config 5's input side): 256 KB per crop does not fit in LDS -> awr_nyu_batch composes the two kernels"""
    ND, DV = mods
    from awr_amd import _lib as L
    rng = np.random.RandomState(9)
    frames, centers = _frames(3, rng)
    store = DV.FrameStore(frames)
    aug, paug = ND.Augmenter(ND.PARAS, -1), DV.ParamAugmenter(ND.PARAS, -1)
    blocks, want = [], []
    for i, (op, trans, scale, rot) in enumerate([("trans", np.array([12.0, -8.0, 15.0]), 1.0, 0.0), ("rot", np.zeros(3), 1.0, 37.0),
                                                 ("scale", np.zeros(3), 1.08, 0.0)]):
        c_uvd = ND.xyz2uvd(centers[i], ND.PARAS, -1).astype(np.float64)
        cube = np.array([300.0, 300.0, 300.0])
        jt = rng.uniform(-100, 100, (21, 3))
        img, M = ND.crop(frames[i].astype(np.float32), c_uvd, cube, np.array([256, 256]))
        want.append(np.asarray(aug.augment(img.copy(), jt.copy(), c_uvd.copy(), cube.copy(), M.copy(), op, trans, scale, rot)[0]).astype(np.float32))
        b = L.NyuSample()
        M2 = DV.set_crop(b, i, c_uvd, cube, np.array([256, 256]), ND.PARAS, 480, 640)
        paug.begin(b)
        paug.augment(DV._Deferred((256, 256)), jt.copy(), c_uvd.copy(), cube.copy(), M2.copy(), op, trans, scale, rot)
        blocks.append(b)
    r = DV.Renderer(store, 256, 4)
    assert r._scratch is not None
    assert torch.equal(r(DV.blocks_to_tensor(blocks)).cpu()[:, 0], torch.from_numpy(np.stack(want)))
    from awr_amd._lib import AwrError
    with pytest.raises(AwrError, match="scratch"):
        r._scratch = None
        r(DV.blocks_to_tensor(blocks))


def test_recrop_of_an_empty_crop_is_reported(mods):
    """the reference raises (np.min of an empty selection, loader.py:116); the kernel flags the sample, Renderer.check raises"""
    ND, DV = mods
    from awr_amd import _lib as L
    store = DV.FrameStore(np.zeros((1, 480, 640), np.uint16))
    paug = DV.ParamAugmenter(ND.PARAS, -1)
    c = np.array([320.0, 240.0, 700.0])
    cube = np.array([300.0, 300.0, 300.0])
    b = L.NyuSample()
    M = DV.set_crop(b, 0, c, cube, np.array([128, 128]), ND.PARAS, 480, 640)
    paug.begin(b)
    paug.augment(DV._Deferred((128, 128)), np.zeros((14, 3)), c, cube, M, "trans", np.array([5.0, 5.0, 5.0]), 1.0, 0.0)
    r = DV.Renderer(store, 128, 2)
    r(DV.blocks_to_tensor([b]))
    with pytest.raises(ValueError, match="empty crop"):
        r.check(1)


def test_device_dataset_equals_host_dataset(mods, tmp_path):
    """DeviceNYU + FrameStore(build_frame_cache) + Renderer == nyu_data.NYU, sample for sample, in both phases (PNG directory in the
    NYU layout; train phase runs the RandomState(23455) stream of loader.py:11), through a DataLoader's collate."""
    ND, DV = mods
    from test_nyu_data_cpu import _write_fake_nyu
    rng = np.random.RandomState(5)
    root = str(tmp_path)
    _write_fake_nyu(root, 10, rng)
    for phase in ("test", "train"):
        if phase == "train":
            os.rename(os.path.join(root, "test"), os.path.join(root, "train"))
            os.rename(os.path.join(root, "center_test_refined.txt"), os.path.join(root, "center_train_refined.txt"))
        kw = dict(img_size=128, aug_para=[10, 0.1, 180]) if phase == "train" else dict(img_size=128)
        host, dev = ND.NYU(root, phase, **kw), DV.DeviceNYU(root, phase, **kw)
        cache = DV.build_frame_cache(root, phase)
        assert np.load(cache, mmap_mode="r").shape == (10, 480, 640)
        store = DV.FrameStore(cache)
        render = DV.Renderer(store, 128, 16)
        loader = torch.utils.data.DataLoader(dev, batch_size=4, shuffle=False, num_workers=0)
        k = 0
        for blocks, jt_xyz, jt_uvd, center, M, cube in loader:
            img = render(blocks).cpu()
            for j in range(blocks.shape[0]):
                h = host[k]
                assert torch.equal(img[j], h[0]), (phase, k)
                for a, b in zip((jt_xyz[j], jt_uvd[j], center[j], M[j], cube[j]), h[1:]):
                    assert torch.equal(a, b), (phase, k)
                k += 1
        assert k == 10
        assert DV.build_frame_cache(root, phase) == cache          # kept, not rebuilt


def test_trainer_on_the_device_loader_trains_the_same_network(mods, tmp_path):
    """config.device_loader through awr_amd.trainer.Trainer (train.py:27-227 on the engines): PNG directory in the NYU layout -> uint16 frame cache ->
    frames in HBM -> parameter-block datasets -> one render launch per batch.  In deterministic mode the run is bitwise reproducible, and because
    the rendered images are bit-identical to the host loader's the trained parameters and the test error are IDENTICAL to a host-loader run
    (same shuffle, same augmentation stream), with and without DataLoader worker processes."""
    import awr_amd
    from awr_amd.config import Config
    from awr_amd.trainer import Trainer
    from test_nyu_data_cpu import _write_fake_nyu
    ND, DV = mods
    rng = np.random.RandomState(21)
    root = os.path.join(str(tmp_path), "data", "nyu")
    os.makedirs(root)
    _write_fake_nyu(root, 12, rng)
    os.rename(os.path.join(root, "test"), os.path.join(root, "train"))
    os.rename(os.path.join(root, "center_test_refined.txt"), os.path.join(root, "center_train_refined.txt"))
    _write_fake_nyu(root, 6, rng)

    def run(device_loader, workers, tag):
        class Cfg(Config):
            net = "resnet_18"
            kernel_size = 1.0
            batch_size = 4
            num_workers = workers
            max_epoch = 1
            print_freq = 2
            vis_freq = 0
            data_dir = os.path.join(str(tmp_path), "data")
            output_dir = os.path.join(str(tmp_path), "out")
            load_model = ""
            exp_id = tag
            use_hipgraph = False
        Cfg.device_loader = device_loader
        torch.manual_seed(0)
        tr = Trainer(Cfg())
        assert isinstance(tr.trainData, DV.DeviceNYU) == device_loader and bool(tr._render) == device_loader
        torch.manual_seed(1)
        tr.train()
        return tr.net.flat_params().clone(), tr.test(-1)

    awr_amd.set_deterministic(True)
    try:
        p_host, mpe_host = run(False, 0, "host")
        p_dev, mpe_dev = run(True, 0, "dev")
    finally:
        awr_amd.set_deterministic(False)
    assert os.path.exists(os.path.join(root, "train_frames_u16.npy")) and os.path.exists(os.path.join(root, "test_frames_u16.npy"))
    assert torch.equal(p_host, p_dev) and mpe_host == mpe_dev
    # with DataLoader worker processes (they replay the random stream per worker, like the reference: a different run) -- in a process of its own: forking
    # workers out of a pytest process that has been through the whole suite takes 40 s per DataLoader (measured: 130 of the suite's 600 s), out of a fresh one
    # a fraction of a second
    import subprocess
    import sys
    code = ("import os, sys, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "import awr_amd\n"
            "from awr_amd.config import Config\n"
            "from awr_amd.trainer import Trainer\n"
            "class Cfg(Config):\n"
            "    net = 'resnet_18'; kernel_size = 1.0; batch_size = 4; num_workers = 2; max_epoch = 1; print_freq = 2; vis_freq = 0\n"
            "    data_dir = %r; output_dir = %r; load_model = ''; exp_id = 'dev2'; use_hipgraph = False; device_loader = True\n"
            "awr_amd.set_deterministic(True)\n"
            "torch.manual_seed(0)\n"
            "tr = Trainer(Cfg())\n"
            "assert tr._render\n"
            "tr.train()\n"
            "mpe = tr.test(-1)\n"
            "assert torch.isfinite(tr.net.flat_params()).all() and np.isfinite(mpe)\n"
            "print('WORKERS_OK', mpe)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(str(tmp_path), "data"), os.path.join(str(tmp_path), "out"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WORKERS_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
