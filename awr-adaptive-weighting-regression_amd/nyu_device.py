"""NYU data path with the image work on the GPU (SURVEY.md 8f-2 "candidates for a GPU kernel later"; csrc/awr_nyu.hip).

The reference's loader (dataloader/nyu_loader.py:38-90, dataloader/loader.py:19-179) spends its time per sample in a PNG decode, a
window gather + nearest resize, one bilinear warp and a normalisation -- ~1.5-1.9 ms of one host core per augmented sample, i.e. eight
DataLoader workers feed about ONE MI355X.  Here

  * the decoded depth frames live in HBM for the whole run (`FrameStore`: uint16 millimetres, 72 757 x 480 x 640 x 2 B = 44.7 GB of the
    288 GB; built once from the PNGs into a uint16 memmap by `build_frame_cache`, so the PNG decode leaves the epoch loop);
  * the host keeps what is tiny and inherently sequential, exactly as awr_amd.nyu_data does it (the SAME methods run: `ParamAugmenter`
    only replaces the four pixel-touching hooks of `Augmenter`): the RandomState(23455) stream, the 3x3 matrices numpy computes in
    float32 / float64, the label arithmetic.  Per sample it emits a 200-byte `awr_nyu_sample` block instead of a 64 KB image;
  * `DeviceNYU` is a map-style dataset yielding (block, jt_xyz, jt_uvd, center_xyz, M, cube): DataLoader workers collate blocks, so
    the reference's per-worker random-stream behaviour (loader.py:11) is preserved for any num_workers;
  * `render(blocks)` turns a batch of blocks into the (B, 1, S, S) float32 batch with ONE kernel launch (crop staged in LDS).

Results are bit-identical (torch.equal) to nyu_data.NYU.__getitem__ -- tests/test_nyu_device_gpu.py.  There is no host fallback inside
this module: without libawr_hip.so / a GPU it raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from . import nyu_data as ND
from .evaluator import uvd2xyz, xyz2uvd

OP_NONE, OP_PERSPECTIVE, OP_AFFINE = 0, 1, 2
BLOCK_BYTES = C.sizeof(L.NyuSample)


class _Deferred:
    """Stands in for the crop while the host runs Augmenter's control flow: only its shape is ever looked at."""

    def __init__(self, dsize):
        self.shape = (int(dsize[1]), int(dsize[0]))


class ParamAugmenter(ND.Augmenter):
    """nyu_data.Augmenter with the pixel work deferred: translate / rotate / scale / augment run unchanged (same draws, same label and
    matrix arithmetic); the four pixel hooks record into `self.block` what csrc/awr_nyu.hip needs."""

    def begin(self, block):
        self.block = block

    def depth_max(self, img):
        return None                                           # the kernel reduces the crop it has just built

    def fringe_floor(self, img):
        return None                                           # ... and its smallest positive value

    def recrop(self, img, center, cube, M, M_inv, dsize, thresh_z=True, bg=0.0, nv_val=0.0):
        b = self.block
        if b.op != OP_NONE:
            raise L.AwrError("one resampling per sample (loader.py:75-86)")
        if float(bg) != 0.0 or not thresh_z:
            raise L.AwrError("the device recrop implements the reference's call: border 0, cube clamp on")
        H = np.dot(M, M_inv)                                                  # loader.py:127, float32 like the reference
        iH = np.linalg.inv(np.asarray(H, np.float64))                          # nyu_data.warp_perspective
        b.op = OP_PERSPECTIVE
        b.m[:] = [float(v) for v in iH.ravel()]
        _, _, _, _, zstart, zend = ND.center2bounds(center, cube, self.paras)
        b.zstart2, b.zend2 = float(zstart), float(zend)
        return img

    def warp_affine(self, img, M, dsize, border):
        b = self.block
        if b.op != OP_NONE:
            raise L.AwrError("one resampling per sample (loader.py:75-86)")
        if float(border) != 0.0:
            raise L.AwrError("the device rotation implements the reference's call: border 0")
        iM = ND._invert_affine(M)
        b.op = OP_AFFINE
        b.m[:] = [float(v) for v in iM.ravel()] + [0.0, 0.0, 1.0]
        return img

    def normalize(self, depth_max, img, center, cube):
        set_normalize(self.block, center, cube)
        return img


def set_normalize(block, center, cube):
    """Loader.normalize's scalars (loader.py:88-101) in numpy's own promotion: float64 unless centre AND cube are float32."""
    half = cube[2] / 2.0
    far = center[2] + half
    lo = center[2] - half
    block.norm32 = int(np.asarray(far).dtype == np.float32 and np.asarray(half).dtype == np.float32 and np.asarray(center[2]).dtype == np.float32)
    block.lo, block.far, block.center_z, block.half = float(lo), float(far), float(center[2]), float(half)


def set_crop(block, frame, center, csize, dsize, paras, fh, fw):
    """Loader.crop's geometry (loader.py:19-51) -> block; returns the crop matrix M (center2transmat)."""
    (ustart, uend, vstart, vend, zstart, zend), size, (us, vs) = ND.crop_geometry(center, csize, dsize, paras)
    cw, ch = uend - ustart, vend - vstart
    if not (cw > 0 and ch > 0 and uend > 0 and vend > 0 and ustart < fw and vstart < fh):
        raise ValueError("crop window [%d, %d) x [%d, %d) does not meet the %d x %d frame" % (ustart, uend, vstart, vend, fw, fh))
    if size[0] <= 0 or size[1] <= 0:
        raise ValueError("crop window of %d x %d pixels resizes to nothing" % (cw, ch))
    block.frame = int(frame)
    block.ustart, block.vstart, block.cw, block.ch = int(ustart), int(vstart), int(cw), int(ch)
    block.rw, block.rh = int(size[0]), int(size[1])
    block.ox, block.oy = int(us), int(vs)
    block.ifx, block.ify = 1.0 / (size[0] / float(cw)), 1.0 / (size[1] / float(ch))        # nyu_data.resize_nearest
    block.zstart, block.zend = float(zstart), float(zend)
    block.op, block.norm32 = OP_NONE, 0
    return ND.center2transmat(center, csize, dsize, paras)


def blocks_to_tensor(blocks):
    """list of L.NyuSample -> (n, BLOCK_BYTES) uint8 CPU tensor"""
    arr = (L.NyuSample * len(blocks))(*blocks)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).view(len(blocks), BLOCK_BYTES)


class FrameStore:
    """The dataset's decoded depth frames, resident in HBM: (n, fh, fw) uint16 (millimetres) or float32."""

    def __init__(self, frames, device="cuda", chunk=256):
        if isinstance(frames, str):
            frames = np.load(frames, mmap_mode="r")
        if frames.ndim != 3:
            raise ValueError("frames must be (n, h, w)")
        if frames.dtype == np.uint16:
            self.ftype, tdt = 0, torch.uint16
        elif frames.dtype == np.float32:
            self.ftype, tdt = 1, torch.float32
        else:
            raise ValueError("frame store holds uint16 or float32 frames, not %s" % frames.dtype)
        if not torch.cuda.is_available():
            raise L.AwrError("the device data path needs a GPU (the host path is awr_amd.nyu_data.NYU)")
        self.n, self.fh, self.fw = (int(v) for v in frames.shape)
        self.data = torch.empty((self.n, self.fh, self.fw), dtype=tdt, device=device)
        for lo in range(0, self.n, chunk):                    # chunked: the memmap is never materialised on the host at once
            hi = min(self.n, lo + chunk)
            self.data[lo:hi].copy_(torch.from_numpy(np.array(frames[lo:hi])), non_blocking=False)      # (np.array: a writable copy of the memmap chunk)

    @property
    def nbytes(self):
        return self.data.numel() * self.data.element_size()


def build_frame_cache(root, phase, out=None):
    """Decode every depth_1*.png of <root>/<phase> once (nyu_loader.py:71-74: depth = G*256 + B) into a uint16 .npy the FrameStore
    memory-maps.  Returns the path; an existing cache with the right frame count is kept."""
    from glob import glob
    files = sorted(glob(os.path.join(root, phase, "depth_1*.png")))
    out = out or os.path.join(root, "%s_frames_u16.npy" % phase)
    if os.path.exists(out):
        have = np.load(out, mmap_mode="r")
        if have.shape[0] == len(files) and have.dtype == np.uint16:
            return out
    if not files:
        raise FileNotFoundError("no depth_1*.png under %s" % os.path.join(root, phase))
    first = ND.read_depth_png(files[0])
    mm = np.lib.format.open_memmap(out, mode="w+", dtype=np.uint16, shape=(len(files),) + first.shape)
    for i, f in enumerate(files):
        d = ND.read_depth_png(f)
        mm[i] = d.astype(np.uint16)                           # G*256 + B <= 65535: exact
    mm.flush()
    del mm
    return out


class DeviceNYU(ND.NYU):
    """nyu_data.NYU with `img` replaced by the sample's parameter block: data[i] = (block uint8[BLOCK_BYTES], jt_xyz, jt_uvd,
    center_xyz, M, cube).  Labels, matrices and the random stream are those of NYU.__getitem__ (nyu_loader.py:38-64)."""

    def __init__(self, root, phase, frame_shape=(480, 640), **kw):
        super().__init__(root, phase, **kw)
        self.fh, self.fw = frame_shape
        self.frame_of = np.arange(len(self.files))

    def _common(self, *a):
        super()._common(*a)
        self.aug = ParamAugmenter(self.paras, self.flip)
        self.fh, self.fw = 480, 640

    @classmethod
    def from_arrays(cls, frames, *a, **kw):
        """labels / centres as nyu_data.NYU.from_arrays; `frames` only lends its shape (or pass the (n, fh, fw) shape tuple): the pixels
        are the FrameStore's business."""
        shape = tuple(frames) if isinstance(frames, (tuple, list)) else tuple(frames.shape)
        self = super().from_arrays(None, *a, **kw)
        self.fh, self.fw = int(shape[-2]), int(shape[-1])
        return self

    def __getitem__(self, index):
        blk = L.NyuSample()
        jt_xyz = self.labels_xyz[index].astype(np.float64).copy()
        cube = self.test_cube[index] if self.phase == "test" else self.cube
        center_xyz = self.centers[index].astype(np.float64).copy()
        center_uvd = xyz2uvd(center_xyz, self.paras, self.flip).astype(np.float64)
        jt_xyz -= center_xyz
        M = set_crop(blk, self.frame_of[index], center_uvd, cube, self.dsize, self.paras, self.fh, self.fw)
        if self.phase == "train" and not self.val:
            op, trans, scale, rot = self.aug.random_aug(*(self.aug_para or (None, None, None)))
            self.aug.begin(blk)
            _, jt_xyz, cube, center_uvd, M = self.aug.augment(_Deferred(self.dsize), jt_xyz, center_uvd, cube, M, op, trans, scale, rot)
            center_xyz = uvd2xyz(center_uvd, self.paras, self.flip)
        else:
            set_normalize(blk, center_xyz, cube)
        jt_uvd = ND.transform_jt_uvd(xyz2uvd(jt_xyz + center_xyz, self.paras, self.flip), M)
        jt_uvd[:, :2] = jt_uvd[:, :2] / (self.img_size / 2.0) - 1
        jt_uvd[:, 2] = (jt_uvd[:, 2] - center_xyz[2]) / (cube[2] / 2.0)
        jt_xyz = jt_xyz / (cube / 2.0)
        return (torch.frombuffer(bytearray(bytes(blk)), dtype=torch.uint8), torch.from_numpy(jt_xyz.astype(np.float32)),
                torch.from_numpy(jt_uvd.astype(np.float32)), torch.from_numpy(center_xyz.astype(np.float32)),
                torch.from_numpy(M.astype(np.float32)), torch.from_numpy(np.asarray(cube, np.float32)))


class Renderer:
    """blocks (B, BLOCK_BYTES) uint8 -> (B, 1, S, S) float32 on the device: one awr_nyu_batch launch on the current stream."""

    def __init__(self, store, img_size=128, max_batch=256):
        self.store, self.S = store, int(img_size)
        dev = store.data.device
        self._blocks = torch.empty((max_batch, BLOCK_BYTES), dtype=torch.uint8, device=dev)
        self.status = torch.zeros(max_batch, dtype=torch.int32, device=dev)
        n = self.S * self.S
        lds_fits = (n + 32) * 4 <= 160 * 1024
        self._scratch = None if lds_fits else torch.empty(max_batch * (n + 2), dtype=torch.float32, device=dev)

    def __call__(self, blocks, out=None):
        B = int(blocks.shape[0])
        if B > self._blocks.shape[0]:
            raise L.AwrError("batch of %d blocks exceeds the renderer's max_batch %d" % (B, self._blocks.shape[0]))
        if blocks.dtype != torch.uint8 or blocks.shape[1] != BLOCK_BYTES:
            raise L.AwrError("blocks must be (B, %d) uint8" % BLOCK_BYTES)
        dev_blocks = self._blocks[:B]
        dev_blocks.copy_(blocks, non_blocking=True)
        if out is None:
            out = torch.empty((B, 1, self.S, self.S), dtype=torch.float32, device=dev_blocks.device)
        st = self.store
        L.call("awr_nyu_batch", st.data.data_ptr(), st.ftype, st.fh, st.fw, dev_blocks.data_ptr(), B, self.S, L.ptr(out),
               self.status.data_ptr(), L.ptr(self._scratch), L.stream())
        return out

    def check(self, B=None):
        """Synchronising: raise where the reference would have (a recrop of a crop without a single positive pixel, loader.py:116)."""
        bad = torch.nonzero(self.status[:B] if B else self.status).flatten().tolist()
        if bad:
            raise ValueError("augmentation recrop of an empty crop (no positive depth) in batch positions %s" % bad)


# ---- operator-level entry points (tests, tools) ---------------------------------------------------------------------------------
def crop_batch(store, blocks, img_size):
    """awr_nyu_crop: -> (crop (B, S, S), stats (B, 2) = {max, smallest positive})"""
    B = int(blocks.shape[0])
    dev = store.data.device
    blk = blocks.to(dev)
    crop = torch.empty((B, img_size, img_size), dtype=torch.float32, device=dev)
    stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
    L.call("awr_nyu_crop", store.data.data_ptr(), store.ftype, store.fh, store.fw, blk.data_ptr(), B, img_size, L.ptr(crop), L.ptr(stats),
           L.stream())
    return crop, stats


def augment_batch(crop, stats, blocks):
    """awr_nyu_augment: materialised crops -> (B, 1, S, S) normalised images"""
    B, S = int(crop.shape[0]), int(crop.shape[-1])
    blk = blocks.to(crop.device)
    out = torch.empty((B, 1, S, S), dtype=torch.float32, device=crop.device)
    status = torch.zeros(B, dtype=torch.int32, device=crop.device)
    L.call("awr_nyu_augment", L.ptr(crop), L.ptr(stats), blk.data_ptr(), B, S, L.ptr(out), status.data_ptr(), L.stream())
    return out, status


def warp(src, m, op, dsize, border=0.0):
    """awr_nyu_warp: src (B, h, w) float32 on the device, m (B, 9) float64 destination->source maps; dsize = (w, h) like cv2"""
    B, sh, sw = (int(v) for v in src.shape)
    dw, dh = int(dsize[0]), int(dsize[1])
    m = torch.as_tensor(np.ascontiguousarray(m, dtype=np.float64)).to(src.device)
    dst = torch.empty((B, dh, dw), dtype=torch.float32, device=src.device)
    L.call("awr_nyu_warp", L.ptr(src), sh, sw, m.data_ptr(), int(op), float(border), B, dh, dw, L.ptr(dst), L.stream())
    return dst


def normalize(img, depth_max, blocks):
    """awr_nyu_normalize: img (B, ...) float32, depth_max (B,) float32, blocks carry lo / far / centre / half"""
    B = int(img.shape[0])
    n = img[0].numel()
    blk = blocks.to(img.device)
    out = torch.empty_like(img)
    L.call("awr_nyu_normalize", L.ptr(img), L.ptr(depth_max), blk.data_ptr(), B, n, L.ptr(out), L.stream())
    return out
