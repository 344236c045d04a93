"""NYU hand-pose dataset without cv2: the test-time path of the reference's data pipeline
(dataloader/nyu_loader.py:14-90, dataloader/loader.py:19-51, :88-101, :181-260) restated with numpy + PIL.

    data = NYU(root, 'test', img_size=128, cube=[300, 300, 300])
    img, jt_xyz, jt_uvd, center_xyz, M, cube = data[i]        # the 6-tuple of nyu_loader.py:66

What is provided: PNG depth decode (depth = G*256 + B, computed in float -- the reference's uint8 arithmetic at
nyu_loader.py:73 overflows on numpy >= 2), cube crop around the refined hand centre, nearest-neighbour resize with
cv2.INTER_NEAREST index semantics, depth normalisation to [-1,1], label transforms, the per-frame test cube rule
(frames >= 2440 use 5/6 of the cube, nyu_loader.py:31-32).  What is NOT provided: the training-time augmentation
(random translate / scale / rotate through cv2.warpPerspective / warpAffine, loader.py:53-179): `phase='train'`
yields un-augmented crops.  The numpy-only helpers are pinned against the reference by tools/gen_golden.py
(tests/golden/loader_fns.npz); the resize follows OpenCV's documented index rule and is not pinned (cv2 is not
installable here).
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
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): src index = min(floor(dst * src/dst_size), src-1)."""
    w, h = size
    sh, sw = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h) * (sh / float(h))).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w) * (sw / float(w))).astype(np.int64), sw - 1)
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


def crop(img, center, csize, dsize, paras=PARAS):
    """loader.py:19-51."""
    dsize = np.asarray(dsize)
    ustart, uend, vstart, vend, zstart, zend = center2bounds(center, csize, paras)
    cropped = bounds2crop(img, ustart, uend, vstart, vend, zstart, zend)
    w, h = uend - ustart, vend - vstart
    scale = min(dsize[0] / w, dsize[1] / h)
    size = (int(w * scale), int(h * scale))
    cropped = resize_nearest(cropped, size)
    res = np.zeros(dsize, dtype=np.float32)
    us, vs = (dsize - np.asarray(size)) / 2.0
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


class NYU(torch.utils.data.Dataset):
    def __init__(self, root, phase, val=False, img_size=128, aug_para=None, cube=(300, 300, 300), jt_num=14):
        assert phase in ("train", "test")
        import scipy.io as sio
        self.name, self.root, self.phase, self.val = "nyu", root, phase, val
        self.paras, self.flip = PARAS, -1
        self.cube = np.asarray(cube, dtype=np.float64)
        self.dsize = np.asarray([img_size, img_size])
        self.img_size, self.jt_num, self.aug_para = img_size, jt_num, aug_para
        data_path = "{}/{}".format(root, phase)
        files = sorted(glob(data_path + "/depth_1*.png"))
        labels = sio.loadmat("{}/joint_data.mat".format(data_path))
        self.labels_xyz = labels["joint_xyz"][0][:, JOINT, :][:, EVAL, :]
        centers = np.loadtxt("{}/center_{}_refined.txt".format(root, phase)).reshape(-1, 3)
        n = min(len(files), len(self.labels_xyz), len(centers))
        self.files, self.centers = files[:n], centers[:n]
        self.test_cube = np.ones([max(n, 8252), 3]) * self.cube
        self.test_cube[2440:, :] = self.test_cube[2440:, :] * 5.0 / 6.0           # nyu_loader.py:31-32
        print("loading dataset, containing %d images." % n)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        img = read_depth_png(self.files[index])
        jt_xyz = self.labels_xyz[index].astype(np.float64).copy()
        cube = self.test_cube[index] if self.phase == "test" else self.cube
        center_xyz = self.centers[index].astype(np.float64).copy()
        center_uvd = xyz2uvd(center_xyz, self.paras, self.flip).astype(np.float64)
        jt_xyz -= center_xyz
        img, M = crop(img, center_uvd, cube, self.dsize, self.paras)
        img = normalize(img.max(), img, center_xyz, cube)
        jt_uvd = transform_jt_uvd(xyz2uvd(jt_xyz + center_xyz, self.paras, self.flip), M)
        jt_uvd[:, :2] = jt_uvd[:, :2] / (self.img_size / 2.0) - 1
        jt_uvd[:, 2] = (jt_uvd[:, 2] - center_xyz[2]) / (cube[2] / 2.0)
        jt_xyz = jt_xyz / (cube / 2.0)
        return (torch.from_numpy(img[np.newaxis, :].astype(np.float32)), torch.from_numpy(jt_xyz.astype(np.float32)),
                torch.from_numpy(jt_uvd.astype(np.float32)), torch.from_numpy(center_xyz.astype(np.float32)),
                torch.from_numpy(M.astype(np.float32)), torch.from_numpy(np.asarray(cube, np.float32)))
