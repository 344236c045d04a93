"""Batched evaluator: the headline "mean 3D joint error (mm)" of the reference (util/eval_tool.py:20-122,
util/util.py:13-20) without the per-sample Python loop / per-sample np.linalg.inv / per-sample device sync
of train.py:141-148.  Same class name and methods as the reference's EvalUtil, plus `feed_batch`."""
import numpy as np


def uvd2xyz(pts, paras, flip=1):
    """Pinhole back-projection (util/util.py:13-20); paras = (fx, fy, u0, v0).  Like the reference, the arithmetic runs in
    float64 against the intrinsics and is stored in the dtype of `pts` (float32 from the evaluator, float64 from the loader)
    before the final float32 cast."""
    p = np.array(pts).reshape(-1, 3).copy()
    par = np.asarray(paras, np.float64)
    p[:, :2] = (p[:, :2] - par[2:]) * p[:, 2:] / par[:2]
    p[:, 1] *= flip
    return p.reshape(np.shape(pts)).astype(np.float32)


def xyz2uvd(pts, paras, flip=1):
    """util/util.py:3-10 (same dtype behaviour as uvd2xyz)."""
    p = np.array(pts).reshape(-1, 3).copy()
    par = np.asarray(paras, np.float64)
    p[:, 1] *= flip
    p[:, :2] = p[:, :2] * par[:2] / p[:, 2:] + par[2:]
    return p.reshape(np.shape(pts)).astype(np.float32)


class EvalUtil:
    def __init__(self, img_size, paras, flip, num_kp):
        self.img_size, self.paras, self.flip, self.num_kp = img_size, paras, flip, num_kp
        self.jt_uvd_pred = []          # original-image uvd per frame: what test.py:105-108 writes to results/*.txt
        self._err = []                 # (n, J) blocks of Euclidean errors in mm

    def feed_batch(self, jt_uvd_pred, jt_xyz_gt, center_xyz, M, cube):
        """All arguments batched on axis 0 (numpy or CPU tensors): eval_tool.py:20-46 for B frames at once."""
        jt = np.array(jt_uvd_pred, dtype=np.float32).copy()
        gt = np.asarray(jt_xyz_gt, np.float32)
        c = np.asarray(center_xyz, np.float32)
        Mi = np.linalg.inv(np.asarray(M, np.float32))                      # batched (B,3,3)
        cube = np.asarray(cube, np.float32)
        jt[:, :, :2] = (jt[:, :, :2] + 1) * self.img_size / 2.0            # :38
        jt[:, :, 2] = jt[:, :, 2] * cube[:, None, 2] / 2.0 + c[:, None, 2]  # :39
        hom = np.concatenate([jt[:, :, :2], np.ones(jt.shape[:2] + (1,), np.float64)], -1)
        jt[:, :, :2] = np.einsum("bij,bkj->bki", Mi.astype(np.float64), hom)[:, :, :2]   # :40-41
        self.jt_uvd_pred.extend(list(jt))
        xyz = uvd2xyz(jt, self.paras, self.flip)
        gt_mm = gt * (cube[:, None, :] / 2.0) + c[:, None, :]              # :46
        self._err.append(np.sqrt(np.sum(np.square(gt_mm - xyz), axis=2)))

    def feed(self, jt_uvd_pred, jt_xyz_gt, center_xyz, M, cube, jt_vis=0, skip_check=False):
        self.feed_batch(np.asarray(jt_uvd_pred)[None], np.asarray(jt_xyz_gt)[None], np.asarray(center_xyz)[None],
                        np.asarray(M)[None], np.asarray(cube)[None])

    def get_measures(self):
        """-> (mean error, median error, AUC, PCK curve, thresholds); eval_tool.py:80-122."""
        e = np.concatenate(self._err, 0).astype(np.float64)
        th = np.linspace(0, 50, 100)
        trapz = getattr(np, "trapezoid", None) or np.trapz
        norm = trapz(np.ones_like(th), th)
        mean = np.mean(e.mean(0))
        med = np.mean(np.median(e, 0))
        pck = (e[None, :, :] <= th[:, None, None]).mean(1).T            # (J, 100)
        auc = np.mean([trapz(pck[j], th) / norm for j in range(e.shape[1])])
        return mean, med, auc, pck.mean(0), th

    def plot_pck(self, path, pck_curve_all, thresholds):
        """eval_tool.py:124-135 (PIL instead of matplotlib)."""
        from .vis_tool import plot_pck
        plot_pck(path, pck_curve_all, thresholds)
