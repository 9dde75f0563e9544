"""ORACLE -- test infrastructure only.  ``cv2.resize(img, (w, h))`` with the default INTER_LINEAR on float32 images,
restated independently of the product (plain NumPy loops over output rows / columns) from OpenCV's published
algorithm (imgproc/resize.cpp: resizeGeneric_ + HResizeLinear / VResizeLinear):

    scale = src / dst (double);  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
    s < 0 -> s = 0, f = 0;   s >= src - 1 -> s = src - 1, f = 0
    horizontal:  row[d] = src[s] * (1 - f) + src[s + 1] * f        (float32)
    vertical:    out    = row0  * (1 - g) + row1       * g         (float32)

This is what /root/reference/datasets/general_eval.py:107 calls.  cv2 is absent from the image, so the restatement is
anchored on the published algorithm and hand-computed vectors (tests/test_eval_io.py), not on cv2 output: that single
comparison is "parity unpinned"."""
import numpy as np


def _axis(n_out, n_in):
    idx, frac = np.empty(n_out, np.int64), np.empty(n_out, np.float32)
    scale = n_in / n_out
    for d in range(n_out):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= n_in - 1:
            s, f = n_in - 1, np.float32(0)
        idx[d], frac[d] = s, f
    return idx, frac


def resize_linear(img, new_h, new_w):
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    xi, xf = _axis(new_w, w)
    yi, yf = _axis(new_h, h)
    rows = np.empty((h, new_w) + img.shape[2:], np.float32)
    for d in range(new_w):
        s, f = xi[d], xf[d]
        rows[:, d] = img[:, s] * (np.float32(1) - f) + img[:, min(s + 1, w - 1)] * f
    out = np.empty((new_h, new_w) + img.shape[2:], np.float32)
    for d in range(new_h):
        s, g = yi[d], yf[d]
        out[d] = rows[s] * (np.float32(1) - g) + rows[min(s + 1, h - 1)] * g
    return out
