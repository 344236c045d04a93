"""Qualitative outputs of the reference's test loop without cv2 / matplotlib: the skeleton overlay of
util/vis_tool.py:17-60 (`VisualUtil.plot`, called every `vis_freq` test batches, train.py:203-213) and the PCK curve of
util/eval_tool.py:124-135 (`plot_pck`, train.py:216), both drawn with PIL and written as PNG."""
import numpy as np

# finger -> (joints drawn as dots, bones as (from, to)); the last entry of every NYU chain is the palm joint 13
_SKELETON = {
    "nyu": (
        ((0, 1), ((0, 1), (1, 13))),
        ((2, 3), ((2, 3), (3, 13))),
        ((4, 5), ((4, 5), (5, 13))),
        ((6, 7), ((6, 7), (7, 13))),
        ((8, 9, 10, 11, 12, 13), ((8, 9), (9, 10), (10, 13), (11, 13), (12, 13))),
    ),
    # 21-joint layouts (hands17 / msra style): wrist 0, then one chain per finger
    "hands": tuple(((f + 1, 6 + 3 * f, 7 + 3 * f, 8 + 3 * f) + ((0,) if f == 4 else ()),
                    ((0, f + 1), (f + 1, 6 + 3 * f), (6 + 3 * f, 7 + 3 * f), (7 + 3 * f, 8 + 3 * f))) for f in range(5)),
}
# thumb..pinky shades, RGB: prediction in red, ground truth in blue (vis_tool.py:10-15 lists the same shades as BGR)
_PRED = ((102, 0, 0), (179, 0, 0), (255, 0, 0), (255, 77, 77), (255, 153, 153))
_GT = ((0, 0, 102), (0, 0, 179), (0, 0, 255), (77, 77, 255), (153, 153, 255))


class VisualUtil:
    def __init__(self, dataset):
        self.dataset = dataset
        key = "nyu" if dataset == "nyu" else "hands" if ("hands" in dataset or dataset == "msra") else None
        if key is None:
            raise ValueError("no skeleton layout for dataset %r" % dataset)
        self.fingers = _SKELETON[key]

    def render(self, img, jt_uvd_pred, jt_uvd_gt=None):
        """img: normalised depth crop in [-1, 1] (any leading singleton dims); joints in crop pixels.  Returns a PIL image."""
        from PIL import Image, ImageDraw
        g = np.clip((np.asarray(img, np.float32).squeeze() + 1.0) * 100.0, 0, 255).astype(np.uint8)      # vis_tool.py:20
        canvas = Image.fromarray(np.repeat(g[:, :, None], 3, axis=2), "RGB")
        draw = ImageDraw.Draw(canvas)
        for joints, colors in ((jt_uvd_pred, _PRED), (jt_uvd_gt, _GT)):
            if joints is None:
                continue
            uv = np.asarray(joints, np.float32).reshape(-1, 3)[:, :2].astype(np.int64)                  # int() truncation, vis_tool.py:35
            for (dots, bones), color in zip(self.fingers, colors):
                for j in dots:
                    draw.ellipse([uv[j][0] - 2, uv[j][1] - 2, uv[j][0] + 2, uv[j][1] + 2], fill=color)
                for s, e in bones:
                    draw.line([tuple(uv[s]), tuple(uv[e])], fill=color, width=1)
        return canvas

    def plot(self, img, path, jt_uvd_pred, jt_uvd_gt=None):
        self.render(img, jt_uvd_pred, jt_uvd_gt if isinstance(jt_uvd_gt, np.ndarray) else None).save(path)


def plot_pck(path, pck_curve, thresholds, size=(640, 480)):
    """Percentage of correct keypoints against the error threshold in mm (eval_tool.py:124-135), as a plain line chart."""
    from PIL import Image, ImageDraw
    W, H = size
    left, right, top, bottom = 60, 20, 20, 45
    canvas = Image.new("RGB", (W, H), (255, 255, 255))
    d = ImageDraw.Draw(canvas)
    x0, x1, y0, y1 = left, W - right, H - bottom, top
    t = np.asarray(thresholds, np.float64)
    p = np.clip(np.asarray(pck_curve, np.float64) * 100.0, 0.0, 100.0)
    tmax = float(t.max()) if t.size and t.max() > 0 else 1.0
    sx = lambda v: x0 + (x1 - x0) * float(v) / tmax            # noqa: E731
    sy = lambda v: y0 + (y1 - y0) * float(v) / 100.0           # noqa: E731
    for k in range(0, 101, 20):                                # horizontal grid + y labels
        d.line([(x0, sy(k)), (x1, sy(k))], fill=(210, 210, 210))
        d.text((x0 - 30, sy(k) - 6), "%3d" % k, fill=(0, 0, 0))
    for k in range(6):                                         # vertical grid + x labels
        v = tmax * k / 5.0
        d.line([(sx(v), y0), (sx(v), y1)], fill=(210, 210, 210))
        d.text((sx(v) - 8, y0 + 6), "%g" % round(v, 1), fill=(0, 0, 0))
    d.rectangle([x0, y1, x1, y0], outline=(0, 0, 0))
    pts = [(sx(a), sy(b)) for a, b in zip(t, p)]
    if len(pts) > 1:
        d.line(pts, fill=(31, 119, 180), width=2)
    for x, y in pts[:: max(1, len(pts) // 25)]:
        d.ellipse([x - 2, y - 2, x + 2, y + 2], fill=(31, 119, 180))
    d.text(((x0 + x1) // 2 - 50, H - 18), "threshold in mm", fill=(0, 0, 0))
    d.text((4, 4), "% of correct keypoints", fill=(0, 0, 0))
    canvas.save(path)
