"""CPU: the HOST half of the device data path (awr_amd.nyu_device; no kernel runs here): the parameter-block builder executes the same
augmentation control flow and label arithmetic as the host loader (nyu_data.Augmenter, pinned to the reference by tests/golden/loader_aug.npz),
the awr_nyu_sample struct matches include/awr_hip.h, and the parameter-block dataset yields the host dataset's labels."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import awr_amd  # noqa: F401
    from awr_amd import _lib, nyu_data, nyu_device
    return _lib, nyu_data, nyu_device


def test_sample_struct_matches_the_header(mods):
    L, ND, DV = mods
    text = open(os.path.join(REPO, "include", "awr_hip.h")).read()
    body = text[text.index("typedef struct awr_nyu_sample {"):text.index("} awr_nyu_sample;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, rest = decl.rsplit(" ", 1) if "," not in decl else decl.split(" ", 1)
        for n in rest.split(","):
            names.append((ctype.strip(), n.strip()))
    want = []
    for ctype, n in names:
        m = re.match(r"(\w+)\[(\d+)\]", n)
        want.append((m.group(1), ctype, int(m.group(2))) if m else (n, ctype, 1))
    got = [(f[0], f[1]) for f in L.NyuSample._fields_]
    assert [w[0] for w in want] == [g[0] for g in got]
    size = {"int64_t": 8, "int32_t": 4, "double": 8}
    for (name, ctype, count), (gname, gtype) in zip(want, got):
        assert C.sizeof(gtype) == size[ctype] * count, name
    assert C.sizeof(L.NyuSample) == DV.BLOCK_BYTES == 200


def test_parameter_blocks_follow_the_host_augmentation(mods):
    """500 draws: chosen op, labels, cube, centre, matrix identical to the host loader's; block contents consistent with the op"""
    L, ND, DV = mods
    aug, paug = ND.Augmenter(ND.PARAS, -1), DV.ParamAugmenter(ND.PARAS, -1)
    rng = np.random.RandomState(3)
    depth = rng.uniform(500, 900, (480, 640)).astype(np.float32)
    seen = set()
    for i in range(500):
        c = np.array([rng.uniform(-120, 120), rng.uniform(-90, 90), rng.uniform(550, 950)])
        cu = ND.xyz2uvd(c, ND.PARAS, -1).astype(np.float64)
        cube = np.array([300.0, 300.0, 300.0]) * (5.0 / 6.0 if i % 5 == 0 else 1.0)
        jt = rng.uniform(-100, 100, (14, 3))
        op, trans, scale, rot = aug.random_aug(10, 0.1, 180)
        assert paug.random_aug(10, 0.1, 180)[0] == op
        img, M = ND.crop(depth.copy(), cu, cube, np.array([128, 128]))
        out = aug.augment(img.copy(), jt.copy(), cu.copy(), cube.copy(), M.copy(), op, trans, scale, rot)
        b = L.NyuSample()
        M2 = DV.set_crop(b, 7, cu, cube, np.array([128, 128]), ND.PARAS, 480, 640)
        assert np.array_equal(M, M2) and b.frame == 7 and b.rw <= 128 and b.rh <= 128 and b.ox >= 0 and b.oy >= 0
        paug.begin(b)
        pout = paug.augment(DV._Deferred((128, 128)), jt.copy(), cu.copy(), cube.copy(), M2.copy(), op, trans, scale, rot)
        for a, p in zip(out[1:], pout[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(p)) and np.asarray(a).dtype == np.asarray(p).dtype
        assert b.op == {"trans": DV.OP_PERSPECTIVE, "scale": DV.OP_PERSPECTIVE, "rot": DV.OP_AFFINE, None: DV.OP_NONE}[op]
        assert b.half > 0 and b.far > b.lo and abs((b.far - b.lo) - 2 * b.half) < 1e-9
        if b.op == DV.OP_PERSPECTIVE:
            assert abs(b.m[8]) > 0 and b.zend2 > b.zstart2
        seen.add(b.op)
    assert seen == {0, 1, 2}
    with pytest.raises(ValueError):
        DV.set_crop(L.NyuSample(), 0, np.array([-900.0, 240.0, 600.0]), np.array([300.0, 300.0, 300.0]), np.array([128, 128]), ND.PARAS, 480, 640)


def test_parameter_block_dataset_yields_the_host_labels(mods):
    L, ND, DV = mods
    rng = np.random.RandomState(9)
    n, nf = 40, 4
    frames = rng.randint(500, 900, (nf, 480, 640)).astype(np.uint16)
    centers = np.stack([rng.uniform(-100, 100, n), rng.uniform(-80, 80, n), rng.uniform(600, 900, n)], 1)
    labels = centers[:, None, :] + rng.uniform(-60, 60, (n, 14, 3))
    kw = dict(frame_of=np.arange(n) % nf, img_size=128, aug_para=[10, 0.1, 180])
    for phase in ("train", "test"):
        host = ND.NYU.from_arrays(frames, labels, centers, phase, **kw)
        dev = DV.DeviceNYU.from_arrays(frames.shape, labels, centers, phase, **kw)
        assert len(host) == len(dev) == n
        for i in range(n):
            h, d = host[i], dev[i]
            assert d[0].dtype == torch.uint8 and d[0].numel() == DV.BLOCK_BYTES
            blk = L.NyuSample.from_buffer_copy(bytes(d[0].numpy()))
            assert blk.frame == i % nf
            for a, b in zip(h[1:], d[1:]):
                assert torch.equal(a, b)
    # blocks survive a DataLoader's collate, worker processes included
    ld = torch.utils.data.DataLoader(DV.DeviceNYU.from_arrays(frames.shape, labels, centers, "test", **kw), batch_size=8, num_workers=2)
    first = next(iter(ld))
    assert first[0].shape == (8, DV.BLOCK_BYTES) and first[2].shape == (8, 14, 3)


def test_frame_store_and_renderer_need_a_gpu(mods):
    L, ND, DV = mods
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.AwrError, match="needs a GPU"):
        DV.FrameStore(np.zeros((1, 480, 640), np.uint16))
