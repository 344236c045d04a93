"""CPU: the cv2-free NYU test-time pipeline.  Helpers against the reference-generated vectors
(tests/golden/loader_fns.npz); the dataset object end to end on a synthetic NYU-layout directory."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ND():
    import awr_amd  # noqa: F401
    from awr_amd import nyu_data
    return nyu_data


def test_helpers_match_reference_vectors(ND, golden_dir):
    g = np.load(os.path.join(golden_dir, "loader_fns.npz"))
    rng = np.random.RandomState(int(g["seed"]))
    centers_xyz = np.stack([rng.uniform(-200, 200, 6), rng.uniform(-150, 150, 6), rng.uniform(500, 1100, 6)], 1)
    assert np.allclose(centers_xyz, g["centers_xyz"])
    cube = np.array([300.0, 300.0, 300.0])
    depth = rng.uniform(400, 1300, (480, 640)).astype(np.float32)
    depth[rng.rand(480, 640) < 0.3] = 0
    for i, c in enumerate(centers_xyz):
        cuvd = ND.xyz2uvd(c, ND.PARAS, -1).astype(np.float64)
        np.testing.assert_allclose(cuvd, g["center_uvd"][i], rtol=0, atol=1e-4)
        b = ND.center2bounds(cuvd, cube)
        assert list(b[:4]) == [int(v) for v in g["bounds"][i][:4]]
        cr = ND.bounds2crop(depth.copy(), *b)
        assert list(cr.shape) == list(g["crop_shape"][i])
        np.testing.assert_array_equal(ND.center2transmat(cuvd, cube, np.array([128, 128])), g["M"][i])
        small = cr[:96, :96].astype(np.float32).copy()
        np.testing.assert_array_equal(ND.normalize(small.max(), small.copy(), c, cube), g["norm"][i])
        rng.uniform(100, 500, (14, 3))          # keep the stream aligned with the generator


def test_resize_nearest_follows_opencv_index_rule(ND):
    img = np.arange(7 * 5, dtype=np.float32).reshape(7, 5)
    out = ND.resize_nearest(img, (10, 14))                      # (w, h)
    assert out.shape == (14, 10)
    assert out[0, 0] == img[0, 0] and out[13, 9] == img[6, 4] and out[2, 3] == img[1, 1]     # floor(dst * src/dst)
    assert np.array_equal(ND.resize_nearest(img, (5, 7)), img)


def _write_fake_nyu(root, n, rng):
    from PIL import Image
    import scipy.io as sio
    os.makedirs(os.path.join(root, "test"))
    centers, xyz_all = [], np.zeros((1, n, 36, 3))
    for i in range(n):
        c = np.array([rng.uniform(-80, 80), rng.uniform(-60, 60), rng.uniform(650, 850)])
        depth = np.full((480, 640), 1500.0)
        uvd = np.array([588.03 * c[0] / c[2] + 320.0, -587.07 * c[1] / c[2] + 240.0])      # xyz2uvd with flip = -1
        yy, xx = np.mgrid[0:480, 0:640]
        hand = (xx - uvd[0]) ** 2 + (yy - uvd[1]) ** 2 < 55 ** 2
        depth[hand] = c[2] + 0.2 * (xx[hand] - uvd[0])
        d = depth.astype(np.int64)
        rgb = np.stack([np.zeros_like(d), d // 256, d % 256], -1).astype(np.uint8)           # depth = G*256 + B
        Image.fromarray(rgb).save(os.path.join(root, "test", "depth_1_%07d.png" % (i + 1)))
        centers.append(c)
        xyz_all[0, i] = c + rng.uniform(-60, 60, (36, 3))
    sio.savemat(os.path.join(root, "test", "joint_data.mat"), {"joint_xyz": xyz_all, "joint_uvd": np.zeros((1, n, 36, 3))})
    np.savetxt(os.path.join(root, "center_test_refined.txt"), np.array(centers))
    return np.array(centers), xyz_all[0]


def test_dataset_end_to_end_on_synthetic_directory(ND, tmp_path):
    rng = np.random.RandomState(3)
    centers, xyz = _write_fake_nyu(str(tmp_path), 4, rng)
    data = ND.NYU(str(tmp_path), "test", img_size=128)
    assert len(data) == 4
    img, jt_xyz, jt_uvd, center_xyz, M, cube = data[1]
    assert img.shape == (1, 128, 128) and img.dtype == torch.float32 and jt_xyz.shape == (14, 3) and M.shape == (3, 3)
    assert float(img.max()) == 1.0 and float(img.min()) >= -1.0                     # background exactly at the far plane
    fg = (img < 0.99).float().mean()
    assert 0.05 < float(fg) < 0.9                                                    # the hand disk is inside the crop
    np.testing.assert_allclose(center_xyz.numpy(), centers[1], rtol=1e-6)
    sel = xyz[1][ND.JOINT][ND.EVAL]
    np.testing.assert_allclose(jt_xyz.numpy(), (sel - centers[1]) / 150.0, atol=1e-5)
    # labels are consistent with the evaluator's inverse chain (eval_tool.py:38-46): uvd -> xyz recovers the ground truth
    from awr_amd.evaluator import EvalUtil
    ev = EvalUtil(128, ND.PARAS, -1, 14)
    ev.feed(jt_uvd.numpy(), jt_xyz.numpy(), center_xyz.numpy(), M.numpy(), cube.numpy())
    assert ev.get_measures()[0] < 0.05                                               # mm


def test_warps_follow_opencv_sampling_rule(ND):
    """cv2.warpAffine / warpPerspective (INTER_LINEAR, BORDER_CONSTANT) restated: identity, whole-pixel shifts, half-pixel
    interpolation, border value, 1/32-pixel coordinate quantisation."""
    rng = np.random.RandomState(0)
    img = (rng.rand(32, 40) * 100 + 500).astype(np.float32)
    eye2 = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(ND.warp_affine(img, eye2, (40, 32)), img)
    assert np.array_equal(ND.warp_perspective(img, np.eye(3), (40, 32)), img)
    sh = ND.warp_affine(img, np.array([[1.0, 0, 3], [0, 1.0, -2]]), (40, 32), border=7.0)        # dst(x,y) = src(x-3, y+2)
    assert np.array_equal(sh[:30, 3:], img[2:, :37]) and np.all(sh[:, :3] == 7.0) and np.all(sh[30:, :] == 7.0)
    half = ND.warp_affine(img, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), (40, 32))                 # dst(x) = src(x - 0.5)
    np.testing.assert_allclose(half[:, 1:], 0.5 * (img[:, :-1] + img[:, 1:]), rtol=1e-6)
    np.testing.assert_allclose(half[:, 0], 0.5 * img[:, 0], rtol=1e-6)                             # the other tap is the border (0)
    # coordinates are quantised to 1/32 pixel: a 1/100-pixel shift samples at round(0.01 * 32) / 32 = 0
    tiny = ND.warp_perspective(img, np.array([[1.0, 0, 0.01], [0, 1.0, 0], [0, 0, 1.0]]), (40, 32))
    assert np.array_equal(tiny, img)
    # rotating twice by 180 degrees about the pixel centre (w//2, h//2) returns the interior
    R = ND.rotation_matrix_2d((20, 16), 180, 1)
    back = ND.warp_affine(ND.warp_affine(img, R, (40, 32)), R, (40, 32))
    np.testing.assert_allclose(back[1:, 1:], img[1:, 1:], rtol=1e-6)
    np.testing.assert_allclose(R, [[-1, 0, 40], [0, -1, 32]], atol=1e-12)


def test_augmentation_matches_reference_vectors(ND, golden_dir):
    """tests/golden/loader_aug.npz: the reference's Loader.random_aug / augment (dataloader/loader.py:53-179) run on the same
    inputs with cv2's three resamplers delegated to the numpy restatements: random stream, chosen op, joints, cube, centre and
    crop matrix agree exactly."""
    g = np.load(os.path.join(golden_dir, "loader_aug.npz"))
    aug = ND.Augmenter(ND.PARAS, -1)
    rng = np.random.RandomState(int(g["seed"]))
    yy, xx = np.mgrid[0:480, 0:640]
    ops_seen = set()
    for i, d in enumerate(g["draws"]):
        op, trans, scale, rot = aug.random_aug(10, 0.1, 180)
        assert ND.Augmenter.OPS.index(op) == int(d[0])
        np.testing.assert_array_equal(np.concatenate([trans, [scale, rot]]), d[1:])
        ops_seen.add(op)
        c_xyz = np.array([rng.uniform(-120, 120), rng.uniform(-90, 90), rng.uniform(600, 900)])
        c_uvd = ND.xyz2uvd(c_xyz, ND.PARAS, -1).astype(np.float64)
        depth = np.full((480, 640), 1400.0, np.float32)
        hand = (xx - c_uvd[0]) ** 2 + (yy - c_uvd[1]) ** 2 < (60 * 750.0 / c_xyz[2]) ** 2
        depth[hand] = (c_xyz[2] + 0.25 * (xx[hand] - c_uvd[0]) - 0.15 * (yy[hand] - c_uvd[1])).astype(np.float32)
        cube = np.array([300.0, 300.0, 300.0])
        jt = rng.uniform(-100, 100, (14, 3))
        img, M = ND.crop(depth.copy(), c_uvd, cube, np.array([128, 128]))
        out = aug.augment(img.copy(), jt.copy(), c_uvd.copy(), cube.copy(), M.copy(), op, trans, scale, rot)
        flat = np.concatenate([np.asarray(out[k], np.float64).ravel() for k in (1, 2, 3, 4)])
        np.testing.assert_array_equal(flat, g["case%d" % i])
        assert abs(float(np.asarray(out[0], np.float64).sum()) - float(g["imgsum%d" % i])) <= 1e-6 * abs(float(g["imgsum%d" % i]))
        assert out[0].min() >= -1.0 and out[0].max() <= 1.0
    assert ops_seen == {"trans", "scale", "rot", None}


def test_train_phase_yields_augmented_consistent_samples(ND, tmp_path):
    """phase='train': one of translate / scale / rotate / nothing per sample; whatever was drawn, image and labels stay
    consistent -- the joints (normalised uvd through the crop matrix) still sit on the hand pixels of the augmented crop."""
    rng = np.random.RandomState(5)
    root = str(tmp_path)
    centers, xyz = _write_fake_nyu(root, 6, rng)
    os.rename(os.path.join(root, "test"), os.path.join(root, "train"))
    os.rename(os.path.join(root, "center_test_refined.txt"), os.path.join(root, "center_train_refined.txt"))
    data = ND.NYU(root, "train", img_size=128, aug_para=[10, 0.1, 180])
    plain = ND.NYU(root, "train", val=True, img_size=128)
    changed = 0
    for i in range(6):
        img, jt_xyz, jt_uvd, center_xyz, M, cube = data[i]
        img0 = plain[i][0]
        assert img.shape == (1, 128, 128) and torch.isfinite(img).all() and torch.isfinite(jt_uvd).all()
        assert float(img.max()) <= 1.0 and float(img.min()) >= -1.0
        changed += int(not torch.equal(img, img0))
        # the synthetic hand is a disc of radius 55 px around the centre in the original image: the crop centre (label 0,0
        # after normalisation = the hand centre) must be foreground, the far corner background
        assert float(img[0, 64, 64]) < 0.99 and float(img[0, 2, 2]) > 0.999
    assert changed >= 3            # RandomState(23455): trans, scale, scale, None, scale, trans


def test_resamplers_reproduce_analytic_known_answers(ND):
    """The three cv2 resamplers are UNPINNED (no cv2 here: tests/golden/pin_report.json "unpinned").  What CAN be pinned without
    OpenCV: transformations whose exact result any correct nearest / bilinear implementation must produce bit for bit."""
    rng = np.random.RandomState(3)
    sq = (rng.rand(24, 24) * 300 + 400).astype(np.float32)
    img = (rng.rand(32, 40) * 300 + 400).astype(np.float32)
    # quarter turns about the centre of the pixel grid ((n-1)/2, (n-1)/2): a pure permutation of pixels
    c = ((24 - 1) / 2.0, (24 - 1) / 2.0)
    for k in (1, 2, 3):
        out = ND.warp_affine(sq, ND.rotation_matrix_2d(c, 90 * k, 1.0), (24, 24))
        assert np.array_equal(out, np.rot90(sq, k)), k                 # cv2: positive angle = counter-clockwise, like np.rot90
    # half turn of a rectangle about its grid centre: both axes flipped
    out = ND.warp_affine(img, ND.rotation_matrix_2d(((40 - 1) / 2.0, (32 - 1) / 2.0), 180, 1.0), (40, 32))
    assert np.array_equal(out, img[::-1, ::-1])
    # whole-pixel shift through the perspective path, with a border value
    sh = ND.warp_perspective(img, np.array([[1.0, 0, -4], [0, 1.0, 5], [0, 0, 1.0]]), (40, 32), border=2.5)   # dst(x,y) = src(x+4, y-5)
    assert np.array_equal(sh[5:, :36], img[:27, 4:]) and np.all(sh[:5] == 2.5) and np.all(sh[:, 36:] == 2.5)
    # exact x2 magnification (dst(x) = src(x/2)): even samples are the source pixels, odd samples the mean of two neighbours
    up = ND.warp_affine(img, np.array([[2.0, 0, 0], [0, 2.0, 0]]), (80, 64))
    assert np.array_equal(up[::2, ::2], img)
    assert np.array_equal(up[::2, 1:-1:2], (img[:, :-1] * np.float32(0.5) + img[:, 1:] * np.float32(0.5)).astype(np.float32))
    assert np.array_equal(up[::2, -1], (img[:, -1] * np.float32(0.5)).astype(np.float32))       # right neighbour = border (0)
    # exact x1/2 minification samples the even source pixels
    dn = ND.warp_affine(img, np.array([[0.5, 0, 0], [0, 0.5, 0]]), (20, 16))
    assert np.array_equal(dn, img[::2, ::2])
    # INTER_NEAREST: integer zoom factors are pure index arithmetic
    assert np.array_equal(ND.resize_nearest(img, (20, 16)), img[::2, ::2])
    assert np.array_equal(ND.resize_nearest(img, (80, 64)), np.repeat(np.repeat(img, 2, 0), 2, 1))
    assert np.array_equal(ND.resize_nearest(img, (120, 96)), np.repeat(np.repeat(img, 3, 0), 3, 1))


def test_resize_nearest_follows_opencv_division_order(ND):
    """OpenCV's resizeNN computes fx = dst / src, ifx = 1. / fx, index = min(cvFloor(i * ifx), src - 1) in doubles (imgproc/resize.cpp).
    A literal scalar restatement of that loop is the independent check here; the one-division shortcut
    floor(i * (src / dst)) is NOT equivalent (a different pixel for ~5 % of size pairs) and the exact-rational rule (i * src) // dst
    is not either -- both are counted so a later 'simplification' shows up."""
    import math
    rng = np.random.RandomState(3)
    naive = exact = total = 0
    for sw, w in [(int(a), int(b)) for a, b in zip(rng.randint(1, 700, 600), rng.randint(1, 257, 600))] + [(6, 34), (241, 128), (173, 128), (320, 128)]:
        row = np.arange(sw, dtype=np.float32)[None, :]
        got = ND.resize_nearest(row, (w, 1))[0].astype(np.int64)
        fx = w / float(sw)
        ifx = 1.0 / fx
        lit = np.array([min(int(math.floor(i * ifx)), sw - 1) for i in range(w)])
        assert np.array_equal(got, lit), (sw, w)
        total += 1
        naive += int((np.minimum(np.floor(np.arange(w) * (sw / float(w))).astype(np.int64), sw - 1) != lit).any())
        exact += int((np.minimum((np.arange(w) * sw) // w, sw - 1) != lit).any())
    assert naive > 0 and exact > 0, "the shortcuts became equivalent?"
    # rows and columns are decoded independently
    img = rng.rand(37, 53).astype(np.float32)
    out = ND.resize_nearest(img, (128, 90))
    ys = [min(int(math.floor(i * (1.0 / (90 / 37.0)))), 36) for i in range(90)]
    xs = [min(int(math.floor(i * (1.0 / (128 / 53.0)))), 52) for i in range(128)]
    assert np.array_equal(out, img[np.array(ys)][:, np.array(xs)])


def _lipschitz(img, border):
    """largest jump between neighbouring samples of the image extended by the constant border: the bilinear interpolant moves by at
    most this much per pixel of coordinate error along an axis"""
    p = np.pad(img.astype(np.float64), 1, constant_values=border)
    return max(np.abs(np.diff(p, axis=0)).max(), np.abs(np.diff(p, axis=1)).max())


def test_bilinear_warps_cross_checked_against_scipy(ND):
    """warp_affine / warp_perspective (the restatements of cv2.warpAffine / warpPerspective the augmentation uses,
    dataloader/loader.py:53-179) against an INDEPENDENT bilinear sampler -- scipy.ndimage.map_coordinates(order=1,
    mode='grid-constant') at the exact inverse-mapped coordinates.  OpenCV quantises source coordinates to 1/32 pixel (INTER_BITS = 5)
    with round-to-nearest, i.e. up to 1/64 pixel of coordinate error per axis (plus 2^-10-pixel rounding of the affine increments), so
    the two may differ by at most L * (1/64 + 1/64 + slack) where L bounds the jump between neighbouring samples.  That turns 'compared with
    itself' into 'cross-checked within the fixed-point bound'; OpenCV's own rounding stays unverified (no cv2 here: pin_report.json)."""
    from scipy import ndimage
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float64)
    worst = 0.0
    for trial in range(24):
        img = (300.0 + 40.0 * np.sin(xx / (5.0 + trial)) * np.cos(yy / (7.0 + trial % 5)) + 10.0 * rng.rand(128, 128)).astype(np.float32)
        border = float(rng.choice([0.0, 300.0]))
        L = _lipschitz(img, border)
        ang, sc = rng.uniform(-180, 180), rng.uniform(0.8, 1.25)
        M = ND.rotation_matrix_2d((64.0 + rng.uniform(-3, 3), 64.0 + rng.uniform(-3, 3)), ang, sc)
        M[:, 2] += rng.uniform(-12, 12, 2)
        got = ND.warp_affine(img, M, (128, 128), border)
        iM = np.linalg.inv(np.vstack([M, [0, 0, 1]]))
        sx, sy = iM[0, 0] * xx + iM[0, 1] * yy + iM[0, 2], iM[1, 0] * xx + iM[1, 1] * yy + iM[1, 2]
        ref = ndimage.map_coordinates(img.astype(np.float64), [sy, sx], order=1, mode="grid-constant", cval=border)
        err = np.abs(got - ref).max()
        assert err <= L * (2.0 / 64 + 2.0 / 1024) + 1e-3, ("affine", trial, err, L)
        worst = max(worst, err / L)
        # homography: the same similarity plus a small projective part (what a scale / rotation augmentation composes to is affine;
        # warpPerspective is called with such matrices in homogeneous form, loader.py:127-151)
        Hm = np.vstack([M, [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1.0]])
        got = ND.warp_perspective(img, Hm, (128, 128), border)
        iH = np.linalg.inv(Hm)
        wq = iH[2, 0] * xx + iH[2, 1] * yy + iH[2, 2]
        sx, sy = (iH[0, 0] * xx + iH[0, 1] * yy + iH[0, 2]) / wq, (iH[1, 0] * xx + iH[1, 1] * yy + iH[1, 2]) / wq
        ref = ndimage.map_coordinates(img.astype(np.float64), [sy, sx], order=1, mode="grid-constant", cval=border)
        err = np.abs(got - ref).max()
        assert err <= L * (2.0 / 64) + 1e-3, ("perspective", trial, err, L)
        worst = max(worst, err / L)
    assert worst > 1e-4      # the quantisation is really there (a float sampler would agree to rounding)
