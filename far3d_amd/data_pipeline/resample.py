"""Filter tables of Pillow's 8-bit separable resampling (libImaging/Resample.c, Pillow 9.4 as pinned by the reference's
py38.yaml:183; the algorithm is unchanged in the Pillow 12 of this image, which pins the tests).  `Image.resize(size)` with the
default filter (BICUBIC for RGB images) is what AV2ResizeCropFlipRotImageV2._img_transform calls
(ref datasets/pipelines/custom_pipeline.py:281).  Host code: O(out_size * ksize) doubles, uploaded once per camera."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2      # Resample.c


def _bicubic(x):   # a = -0.5
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def _bilinear(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def pil_resample_coeffs(in_size, out_size, filter="bicubic"):
    """precompute_coeffs + normalize_coeffs_8bpc for the full-image box.  Returns (bounds (out,2) int32 [first sample, count],
    coeffs (out, ksize) int32 22-bit fixed point, ksize)."""
    fn, fsupport = FILTERS[filter]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        w = fn((x + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:            # same left-to-right double accumulation as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int32)
    return bounds, fixed, ksize
