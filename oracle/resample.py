"""TEST INFRASTRUCTURE (see oracle/__init__.py): numpy statement of Pillow's two-pass 8-bit resize, i.e. of what the HIP kernels
far3d_image_resample_h / _v compute from the tables of far3d_amd.data_pipeline.resample.pil_resample_coeffs.  Follows
libImaging/Resample.c (ImagingResampleHorizontal_8bpc / Vertical_8bpc) of the Pillow pinned by the reference (py38.yaml:183), which
`AV2ResizeCropFlipRotImageV2._img_transform` reaches through `Image.resize` (ref datasets/pipelines/custom_pipeline.py:281).
Pinned against Pillow itself in tests/test_data_contract_cpu.py."""
import numpy as np

from far3d_amd.data_pipeline.resample import PRECISION_BITS, pil_resample_coeffs


def resample_u8_reference(img, out_w, out_h, filter="bicubic"):
    """Horizontal pass to an 8-bit intermediate, then vertical.  img (H,W,3) uint8 -> (out_h,out_w,3) uint8."""
    H, W, _ = img.shape
    bh, kh, _ = pil_resample_coeffs(W, out_w, filter)
    bv, kv, _ = pil_resample_coeffs(H, out_h, filter)
    tmp = np.zeros((H, out_w, 3), dtype=np.uint8)
    src = img.astype(np.int64)
    for x in range(out_w):
        x0, n = bh[x]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * kh[x, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((out_h, out_w, 3), dtype=np.uint8)
    t64 = tmp.astype(np.int64)
    for y in range(out_h):
        y0, n = bv[y]
        acc = (1 << (PRECISION_BITS - 1)) + (t64[y0:y0 + n] * kv[y, :n].astype(np.int64)[:, None, None]).sum(0)
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out
