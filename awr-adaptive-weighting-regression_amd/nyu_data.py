"""NYU hand-pose dataset without cv2: the test-time path of the reference's data pipeline
(dataloader/nyu_loader.py:14-90, dataloader/loader.py:19-51, :88-101, :181-260) restated with numpy + PIL.

    data = NYU(root, 'test', img_size=128, cube=[300, 300, 300])
    img, jt_xyz, jt_uvd, center_xyz, M, cube = data[i]        # the 6-tuple of nyu_loader.py:66

What is provided: PNG depth decode (depth = G*256 + B, computed in float -- the reference's uint8 arithmetic at
nyu_loader.py:73 overflows on numpy >= 2), cube crop around the refined hand centre, nearest-neighbour resize with
cv2.INTER_NEAREST index semantics, depth normalisation to [-1,1], label transforms, the per-frame test cube rule
(frames >= 2440 use 5/6 of the cube, nyu_loader.py:31-32), and the training-time augmentation (one of translate /
scale / rotate / nothing per sample, loader.py:53-179) with the reference's random stream (RandomState(23455),
loader.py:11) and label arithmetic.  The numpy-only helpers, the random stream and the augmentation's label / matrix /
cube outputs are pinned against the reference by tools/gen_golden.py (tests/golden/loader_fns.npz, loader_aug.npz).
The three image resamplers -- cv2.resize(INTER_NEAREST), cv2.warpAffine and cv2.warpPerspective with INTER_LINEAR /
BORDER_CONSTANT -- are restated from OpenCV's published fixed-point scheme (coordinates in 1/32 pixel, 10-bit affine
deltas) and are NOT pinned: cv2 cannot be installed here.
"""
import os
from glob import glob

import numpy as np
import torch

from .evaluator import uvd2xyz, xyz2uvd

JOINT = np.array([0, 1, 3, 5, 6, 7, 9, 11, 12, 13, 15, 17, 18, 19, 21, 23, 24, 25, 27, 28, 32, 30, 31])   # nyu_loader.py:9
EVAL = np.array([0, 2, 4, 6, 8, 10, 12, 14, 16, 17, 18, 21, 22, 20])                                         # nyu_loader.py:11
PARAS = (588.03, 587.07, 320.0, 240.0)                                                                       # nyu_loader.py:23


def read_depth_png(path):
    """NYU synthetic-style PNG: depth in mm = G*256 + B (cv2.imread is BGR: channels 1 and 0 there)."""
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    return rgb[:, :, 1] * 256.0 + rgb[:, :, 2]


def center2bounds(center, csize, paras=PARAS):
    """loader.py:181-188 (int() truncation included)."""
    center, csize, f = np.asarray(center, np.float64), np.asarray(csize, np.float64), np.asarray(paras[:2], np.float64)
    ustart, vstart = center[:2] - (csize[:2] / 2.0) / center[2] * f + 0.5
    uend, vend = center[:2] + (csize[:2] / 2.0) / center[2] * f + 0.5
    return int(ustart), int(uend), int(vstart), int(vend), center[2] - csize[2] / 2.0, center[2] + csize[2] / 2.0


def bounds2crop(img, ustart, uend, vstart, vend, zstart, zend, thresh_z=True, bg=0):
    """loader.py:190-208: crop with zero padding outside the image, clamp depths to the cube."""
    h, w = img.shape[:2]
    bbox = [max(vstart, 0), min(vend, h), max(ustart, 0), min(uend, w)]
    out = img[bbox[0]:bbox[1], bbox[2]:bbox[3]]
    out = np.pad(out, ((abs(vstart) - bbox[0], abs(vend) - bbox[1]), (abs(ustart) - bbox[2], abs(uend) - bbox[3])), mode="constant",
                 constant_values=bg)
    if thresh_z:
        out = out.copy()
        m1 = np.logical_and(out < zstart, out != 0)
        m2 = np.logical_and(out > zend, out != 0)
        out[m1] = zstart
        out[m2] = 0
    return out


def resize_nearest(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST), OpenCV's resizeNN (imgproc/resize.cpp): with fx = dst / src as a
    double and ifx = 1. / fx, source index = min(floor(dst_index * ifx), src - 1).  The order of the two divisions matters: for 4.7 % of
    the (src, dst) size pairs up to 700 x 256 the one-division form floor(i * (src / dst)) picks a different source pixel somewhere
    (tests/test_nyu_data_cpu.py::test_resize_nearest_follows_opencv_division_order)."""
    w, h = size
    sh, sw = img.shape[:2]
    ify, ifx = 1.0 / (h / float(sh)), 1.0 / (w / float(sw))
    ys = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    return img[ys][:, xs]


def center2transmat(center, csize, dsize, paras=PARAS):
    """loader.py:211-240: original-image pixels -> crop pixels (scale + translate)."""
    ustart, uend, vstart, vend, _, _ = center2bounds(center, csize, paras)
    t1 = np.eye(3)
    t1[0][2], t1[1][2] = -ustart, -vstart
    w, h = uend - ustart, vend - vstart
    scale = min(dsize[0] / w, dsize[1] / h)
    size = (int(w * scale), int(h * scale))
    sc = scale * np.eye(3)
    sc[2][2] = 1
    t2 = np.eye(3)
    t2[0][2] = int(np.floor(dsize[0] / 2.0 - size[0] / 2.0))
    t2[1][2] = int(np.floor(dsize[1] / 2.0 - size[1] / 2.0))
    return np.dot(t2, np.dot(sc, t1)).astype(np.float32)


def crop_geometry(center, csize, dsize, paras=PARAS):
    """The geometry of Loader.crop (loader.py:31-47): the window in the frame, its extent after the nearest-neighbour resize and where
    that lands in the dsize crop.  Shared by the host crop below and by the device path's parameter blocks (awr_amd.nyu_device)."""
    dsize = np.asarray(dsize)
    bounds = center2bounds(center, csize, paras)
    w, h = bounds[1] - bounds[0], bounds[3] - bounds[2]
    scale = min(dsize[0] / w, dsize[1] / h)
    size = (int(w * scale), int(h * scale))
    us, vs = (dsize - np.asarray(size)) / 2.0
    return bounds, size, (us, vs)


def crop(img, center, csize, dsize, paras=PARAS):
    """loader.py:19-51."""
    dsize = np.asarray(dsize)
    (ustart, uend, vstart, vend, zstart, zend), size, (us, vs) = crop_geometry(center, csize, dsize, paras)
    cropped = bounds2crop(img, ustart, uend, vstart, vend, zstart, zend)
    cropped = resize_nearest(cropped, size)
    res = np.zeros(dsize, dtype=np.float32)
    res[int(vs):int(vs + size[1]), int(us):int(us + size[0])] = cropped
    return res, center2transmat(center, csize, dsize, paras)


def normalize(depth_max, img, center, cube):
    """loader.py:88-101: background / invalid -> far plane, clip to the cube, scale to [-1,1]."""
    img = img.copy()
    far = center[2] + cube[2] / 2.0
    img[img == depth_max] = far
    img[img == 0] = far
    img = np.clip(img, center[2] - cube[2] / 2.0, far)
    img -= center[2]
    img /= cube[2] / 2.0
    return img


def transform_jt_uvd(jt_uvd, M):
    """loader.py:254-260."""
    pts = np.hstack([jt_uvd[:, :2], np.ones((jt_uvd.shape[0], 1))])
    pts = np.dot(M, pts.T).T
    pts[:, :2] /= pts[:, 2:]
    return np.hstack([pts[:, :2], jt_uvd[:, 2:]]).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# bilinear warps with OpenCV's sampling rule (imgwarp.cpp): source coordinates are quantised to 1/32 pixel
# (INTER_BITS = 5), the four taps are weighted with (1 - fx/32, fx/32) x (1 - fy/32, fy/32), taps outside the image
# read the constant border value.
# ---------------------------------------------------------------------------------------------------------------
_INTER_BITS, _AB_BITS = 5, 10
_TAB = 1 << _INTER_BITS


def _bilinear_q5(img, X, Y, border):
    """img sampled at fixed-point coordinates X, Y (int64 arrays, 1/32 pixel)."""
    h, w = img.shape
    sx, sy = X >> _INTER_BITS, Y >> _INTER_BITS
    fx = (X & (_TAB - 1)).astype(np.float32) / _TAB
    fy = (Y & (_TAB - 1)).astype(np.float32) / _TAB

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(border)).astype(np.float32)
    top = tap(sy, sx) * (1 - fx) + tap(sy, sx + 1) * fx
    bot = tap(sy + 1, sx) * (1 - fx) + tap(sy + 1, sx + 1) * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


def _invert_affine(M):
    """cv2.warpAffine without WARP_INVERSE_MAP maps dst -> src through the inverse of the 2x3 matrix."""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    return np.array([[A11, A12, -A11 * M[0, 2] - A12 * M[1, 2]], [A21, A22, -A21 * M[0, 2] - A22 * M[1, 2]]])


def warp_affine(img, M, dsize, border=0.0):
    """cv2.warpAffine(img, M, (w, h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=border)."""
    w, h = int(dsize[0]), int(dsize[1])
    iM = _invert_affine(M)
    scale = float(1 << _AB_BITS)
    rd = (1 << _AB_BITS) // _TAB // 2
    xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    adelta = np.rint(iM[0, 0] * xs * scale).astype(np.int64)
    bdelta = np.rint(iM[1, 0] * xs * scale).astype(np.int64)
    X0 = np.rint((iM[0, 1] * ys + iM[0, 2]) * scale).astype(np.int64) + rd
    Y0 = np.rint((iM[1, 1] * ys + iM[1, 2]) * scale).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (_AB_BITS - _INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (_AB_BITS - _INTER_BITS)
    return _bilinear_q5(np.asarray(img, np.float32), X, Y, border)


def warp_perspective(img, H, dsize, border=0.0):
    """cv2.warpPerspective(img, H, (w, h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=border)."""
    w, h = int(dsize[0]), int(dsize[1])
    iH = np.linalg.inv(np.asarray(H, np.float64))
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    W = iH[2, 0] * xs + iH[2, 1] * ys + iH[2, 2]
    W = np.where(W != 0, _TAB / np.where(W != 0, W, 1.0), 0.0)
    fX = np.clip((iH[0, 0] * xs + iH[0, 1] * ys + iH[0, 2]) * W, -2.0 ** 31, 2.0 ** 31 - 1)
    fY = np.clip((iH[1, 0] * xs + iH[1, 1] * ys + iH[1, 2]) * W, -2.0 ** 31, 2.0 ** 31 - 1)
    return _bilinear_q5(np.asarray(img, np.float32), np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64), border)


def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D: positive angles rotate counter-clockwise in image coordinates (origin top-left)."""
    a = np.deg2rad(angle_deg)
    alpha, beta = scale * np.cos(a), scale * np.sin(a)
    return np.array([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]], [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]])


def rotate_pts(pt, center, angle_deg):
    """loader.py:242-252."""
    a = angle_deg * np.pi / 180.0
    out = pt.copy()
    dx, dy = pt[:, 0] - center[0], pt[:, 1] - center[1]
    out[:, 0] = dx * np.cos(a) - dy * np.sin(a)        # stored in pt's dtype (float32 from xyz2uvd) before the centre is
    out[:, 1] = dx * np.sin(a) + dy * np.cos(a)        # added back: two roundings, like the reference
    out[:, :2] += center[:2]
    return out.astype(np.float32)


class Augmenter:
    """The reference's per-sample augmentation (loader.py:53-179): one of translate (3D shift of the crop centre), scale
    (cube size) or rotate (in-plane, about the crop centre), or nothing, drawn from ONE RandomState(23455) per dataset
    object (loader.py:11 -- DataLoader workers therefore replay identical streams, as in the reference)."""

    OPS = ("trans", "scale", "rot", None)

    def __init__(self, paras=PARAS, flip=-1, seed=23455):
        self.paras, self.flip = paras, flip
        self.seed = np.random.RandomState(seed)

    def random_aug(self, sigma_trans=None, sigma_scale=None, sigma_rot=None):
        """loader.py:53-73 (same draws in the same order)."""
        sigma_trans = 35.0 if sigma_trans is None else sigma_trans
        sigma_scale = 0.05 if sigma_scale is None else sigma_scale
        sigma_rot = 180.0 if sigma_rot is None else sigma_rot
        op = self.OPS[self.seed.randint(0, len(self.OPS))]
        trans = self.seed.randn(3) * sigma_trans
        scale = abs(1.0 + self.seed.randn() * sigma_scale)
        rot = self.seed.uniform(-sigma_rot, sigma_rot)
        return op, trans, scale, rot

    def _xyz(self, uvd):
        return uvd2xyz(uvd, self.paras, self.flip)

    def _uvd(self, xyz):
        return xyz2uvd(xyz, self.paras, self.flip)

    # The four places where augmentation touches PIXELS are methods, so that awr_amd.nyu_device can run the SAME control flow and label
    # arithmetic while recording what the device kernels need instead of resampling on the host.
    def depth_max(self, img):
        return img.max()                                   # loader.py:76

    def fringe_floor(self, img):
        return np.min(img[img > 0]) - 1                    # loader.py:116 / :175: nv_val

    def warp_affine(self, img, M, dsize, border):
        return warp_affine(img, M, dsize, border=border)

    def normalize(self, depth_max, img, center, cube):
        return normalize(depth_max, img, center, cube)

    def recrop(self, img, center, cube, M, M_inv, dsize, thresh_z=True, bg=0.0, nv_val=0.0):
        """loader.py:123-137: re-sample the crop for a new centre / cube, clean the interpolation fringe, clamp to the cube."""
        img = warp_perspective(img, np.dot(M, M_inv), dsize, border=float(bg))
        img[img < nv_val] = bg
        if thresh_z:
            _, _, _, _, zstart, zend = center2bounds(center, cube, self.paras)
            near = np.logical_and(img < zstart, img != 0)
            far = np.logical_and(img > zend, img != 0)
            img[near] = zstart
            img[far] = 0.0
        return img.astype(np.float32)

    def translate(self, img, jt_xyz, center, cube, M, trans, pad_value=0):
        """loader.py:103-121."""
        if np.allclose(trans, 0.0):
            return img, jt_xyz, center, M
        new_center = self._uvd(self._xyz(center) + trans)
        if not np.allclose(center[2], 0.0) or np.allclose(new_center[2], 0.0):
            new_M = center2transmat(new_center, cube, np.array(img.shape), self.paras)
            img = self.recrop(img, new_center, cube, new_M, np.linalg.inv(M), img.shape, thresh_z=True, bg=pad_value,
                              nv_val=self.fringe_floor(img))
        else:
            new_M = M
        jt_xyz = jt_xyz + self._xyz(center) - self._xyz(new_center)
        return img, jt_xyz, new_center, new_M

    def rotate(self, img, jt_xyz, center, rot, pad_value=0):
        """loader.py:140-160."""
        if np.allclose(rot, 0.0):
            return img, jt_xyz
        rot = np.mod(rot, 360)
        rotM = rotation_matrix_2d((img.shape[1] // 2, img.shape[0] // 2), -rot, 1)
        img = self.warp_affine(img, rotM, (img.shape[1], img.shape[0]), pad_value)
        center_xyz = self._xyz(center)
        jt_uvd = rotate_pts(self._uvd(jt_xyz + center_xyz), center, rot)
        return img, self._xyz(jt_uvd) - center_xyz

    def scale(self, img, center, cube, M, scale, pad_value=0):
        """loader.py:163-179."""
        if np.allclose(scale, 1.0):
            return img, cube, M
        new_cube = cube * scale
        if not np.allclose(center[2], 0.0):
            new_M = center2transmat(center, new_cube, np.array(img.shape), self.paras)
            img = self.recrop(img, center, new_cube, new_M, np.linalg.inv(M), img.shape, bg=pad_value, nv_val=self.fringe_floor(img))
        else:
            new_M = M
        return img, new_cube, new_M

    def augment(self, img, jt_xyz, center, cube, M, op, trans, scale, rot):
        """loader.py:75-86."""
        depth_max = self.depth_max(img)
        if op == "trans":
            img, jt_xyz, center, M = self.translate(img, jt_xyz, center, cube, M, trans)
        elif op == "rot":
            img, jt_xyz = self.rotate(img, jt_xyz, center, rot)
        elif op == "scale":
            img, cube, M = self.scale(img, center, cube, M, scale)
        return self.normalize(depth_max, img, center, cube), jt_xyz, cube, center, M


class NYU(torch.utils.data.Dataset):
    def __init__(self, root, phase, val=False, img_size=128, aug_para=None, cube=(300, 300, 300), jt_num=14):
        import scipy.io as sio
        self._common(root, phase, val, img_size, aug_para, cube, jt_num)
        data_path = "{}/{}".format(root, phase)
        files = sorted(glob(data_path + "/depth_1*.png"))
        labels = sio.loadmat("{}/joint_data.mat".format(data_path))
        self.labels_xyz = labels["joint_xyz"][0][:, JOINT, :][:, EVAL, :]
        centers = np.loadtxt("{}/center_{}_refined.txt".format(root, phase)).reshape(-1, 3)
        n = min(len(files), len(self.labels_xyz), len(centers))
        self.files, self.centers, self.frames = files[:n], centers[:n], None
        self._cubes(n)
        print("loading dataset, containing %d images." % n)

    def _common(self, root, phase, val, img_size, aug_para, cube, jt_num):
        assert phase in ("train", "test")
        self.name, self.root, self.phase, self.val = "nyu", root, phase, val
        self.paras, self.flip = PARAS, -1
        self.cube = np.asarray(cube, dtype=np.float64)
        self.dsize = np.asarray([img_size, img_size])
        self.img_size, self.jt_num, self.aug_para = img_size, jt_num, aug_para
        self.aug = Augmenter(self.paras, self.flip)

    def _cubes(self, n):
        self.test_cube = np.ones([max(n, 8252), 3]) * self.cube
        self.test_cube[2440:, :] = self.test_cube[2440:, :] * 5.0 / 6.0           # nyu_loader.py:31-32

    @classmethod
    def from_arrays(cls, frames, labels_xyz, centers, phase, frame_of=None, val=False, img_size=128, aug_para=None, cube=(300, 300, 300), jt_num=14):
        """The same dataset over decoded frames held in memory ((n_frames, 480, 640) depth in mm, e.g. the uint16 cache of
        awr_amd.nyu_device.build_frame_cache) instead of a PNG directory: labels_xyz (n, J, 3) camera-space joints, centers (n, 3) refined
        hand centres, frame_of[i] = the frame of sample i (default i)."""
        self = cls.__new__(cls)
        self._common(None, phase, val, img_size, aug_para, cube, jt_num)
        self.frames, self.labels_xyz, self.centers = frames, np.asarray(labels_xyz), np.asarray(centers, np.float64).reshape(-1, 3)
        n = len(self.centers)
        self.files = [None] * n
        self.frame_of = np.arange(n) if frame_of is None else np.asarray(frame_of)
        self._cubes(n)
        return self

    def read_frame(self, index):
        if self.frames is None:
            return read_depth_png(self.files[index])
        return np.asarray(self.frames[self.frame_of[index]], dtype=np.float32)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        img = self.read_frame(index)
        jt_xyz = self.labels_xyz[index].astype(np.float64).copy()
        cube = self.test_cube[index] if self.phase == "test" else self.cube
        center_xyz = self.centers[index].astype(np.float64).copy()
        center_uvd = xyz2uvd(center_xyz, self.paras, self.flip).astype(np.float64)
        jt_xyz -= center_xyz
        img, M = crop(img, center_uvd, cube, self.dsize, self.paras)
        if self.phase == "train" and not self.val:                               # nyu_loader.py:55-60
            op, trans, scale, rot = self.aug.random_aug(*(self.aug_para or (None, None, None)))
            img, jt_xyz, cube, center_uvd, M = self.aug.augment(img, jt_xyz, center_uvd, cube, M, op, trans, scale, rot)
            center_xyz = uvd2xyz(center_uvd, self.paras, self.flip)
        else:
            img = normalize(img.max(), img, center_xyz, cube)
        jt_uvd = transform_jt_uvd(xyz2uvd(jt_xyz + center_xyz, self.paras, self.flip), M)
        jt_uvd[:, :2] = jt_uvd[:, :2] / (self.img_size / 2.0) - 1
        jt_uvd[:, 2] = (jt_uvd[:, 2] - center_xyz[2]) / (cube[2] / 2.0)
        jt_xyz = jt_xyz / (cube / 2.0)
        return (torch.from_numpy(img[np.newaxis, :].astype(np.float32)), torch.from_numpy(jt_xyz.astype(np.float32)),
                torch.from_numpy(jt_uvd.astype(np.float32)), torch.from_numpy(center_xyz.astype(np.float32)),
                torch.from_numpy(M.astype(np.float32)), torch.from_numpy(np.asarray(cube, np.float32)))
