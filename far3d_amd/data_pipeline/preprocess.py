"""Device image pre-processing with the reference's pipeline semantics.

Mirrors, in this order, the `test_pipeline` steps between image loading and the format bundle
(ref projects/configs/far3d.py:190-193):
  AV2ResizeCropFlipRotImageV2   ref datasets/pipelines/custom_pipeline.py:48-338 (sampling :313-326, portrait camera :328-338,
                                pixel transform + `ida_mat` :277-311, intrinsics / lidar2img update :162,173-174)
  NormalizeMultiviewImage       ref datasets/pipelines/transform_3d.py:74-101 (mmcv.imnormalize: float32 (x - mean) * (1/std))
  AV2PadMultiViewImage          ref datasets/pipelines/custom_pipeline.py:341-378 ('same2max', pad value 0)
and the HWC -> CHW transpose of PETRFormatBundle3D (ref datasets/pipelines/formating.py:51-55).

The pixels never visit the host after the upload of the raw uint8 images: per camera two HIP launches (horizontal and vertical
pass of Pillow's 8-bit resampling restricted to the crop window; the second one also flips, normalises, pads and transposes)
write straight into the (N,3,H,W) tensor the engine consumes.  Random draws come from `rng` in the reference's order, so seeding
numpy like the reference does reproduces its augmentation parameters.  No CPU pixel path exists here (the reference's own
PIL path is the checker in tests/)."""
import ctypes

import numpy as np
import torch

from .. import lib as _lib
from .resample import pil_resample_coeffs

DEFAULT_AUG = dict(resize_lim=(0.47, 0.55), final_dim=(640, 960), bot_pct_lim=(0.0, 0.0), rot_lim=(0.0, 0.0), rand_flip=False)
DEFAULT_NORM = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)   # ref far3d.py:13-14


def sample_augmentation(H, W, conf, rng=np.random):
    """One draw of the image augmentation (ref custom_pipeline.py:313-326).  A seeded run must consume the generator exactly like
    the reference, so the ORDER of the draws is part of the contract: scale, bottom-crop fraction, horizontal offset, the flip
    coin (tossed only when flipping is enabled), rotation angle.  Returns (scale, (scaled_w, scaled_h), crop box, flip, angle)."""
    out_h, out_w = conf["final_dim"]
    scale = rng.uniform(*conf["resize_lim"])
    scaled_w, scaled_h = int(W * scale), int(H * scale)
    kept_rows = (1 - rng.uniform(*conf["bot_pct_lim"])) * scaled_h
    top = int(kept_rows) - out_h
    left = int(rng.uniform(0, max(0, scaled_w - out_w)))
    flip = bool(conf["rand_flip"] and rng.choice([0, 1]))
    angle = rng.uniform(*conf["rot_lim"])
    return scale, (scaled_w, scaled_h), (left, top, left + out_w, top + out_h), flip, angle


def sample_augmentation_portrait(H, W):
    """A portrait camera (H > W; AV2's ring_front_center) is enlarged until a landscape window of the transposed size fits and
    that window is cut from the centre (ref custom_pipeline.py:328-338).  Deterministic: no draws."""
    out_h, out_w = W, H
    scale = np.round((H + 50) / W, 2)
    scaled_w, scaled_h = int(W * scale), int(H * scale)
    left, top = int((scaled_w - out_w) / 2), int((scaled_h - out_h) / 2)
    return scale, (scaled_w, scaled_h), (left, top, left + out_w, top + out_h)


def ida_matrix(resize, crop, flip=False, rotate=0.0):
    """3x3 homography of the image augmentation, pixel (x, y, 1) of the source -> pixel of the network input
    (ref custom_pipeline.py:294-311 builds it step by step).  In closed form the chain is
        x' = R (F (s x - c) + f) + (b - R b)
    with s the scale, c the crop's top-left corner, F = diag(-1, 1) and f = (crop width, 0) when flipped (else F = I, f = 0),
    R = [[cos, sin], [-sin, cos]] of the rotation angle (degrees) about the crop centre b.  Evaluated in float64 and rounded once
    to float32.  With rotate == 0 (the only value the reference pipeline accepts, custom_pipeline.py:69 -- flipped or not) every
    entry is an integer or the float32 scale, so the result EQUALS the reference's float32 chain; with a rotation the reference
    rounds cos / sin and every intermediate product to float32 and the two differ by float32 rounding (<= 4e-6 relative to the
    largest entry; tests/test_host_cpu.py holds both statements)."""
    left, top, right, bottom = (float(v) for v in crop)
    F = np.diag([-1.0, 1.0]) if flip else np.eye(2)
    f = np.array([right - left, 0.0]) if flip else np.zeros(2)
    th = float(rotate) / 180.0 * np.pi
    R = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
    b = np.array([right - left, bottom - top]) / 2.0
    M = np.eye(3)
    M[:2, :2] = float(np.float32(resize)) * (R @ F)          # the reference scales a float32 identity
    M[:2, 2] = R @ (F @ -np.array([left, top]) + f) + (b - R @ b)
    return torch.from_numpy(M.astype(np.float32))


class ImagePreprocessor:
    """results = pre(results): `results['img']` a list of N raw camera images -- uint8 (H,W,3) numpy arrays / tensors (host or
    device), or the float32 arrays AV2LoadMultiViewImageFromFiles(to_float32=True) produces (they hold integers; the reference
    itself goes through np.uint8, custom_pipeline.py:73,103) -- plus `intrinsics` / `extrinsics` lists of 4x4 arrays.
    Afterwards `results['img']` is ONE device tensor (N,3,padH,padW), and intrinsics, cam2img, lidar2img, ida_mat, img_shape,
    pad_shape are updated exactly like the three reference transforms do."""

    def __init__(self, data_aug_conf=None, img_norm_cfg=None, device="cuda:0", out_dtype=torch.float32, rng=np.random):
        self.conf = dict(DEFAULT_AUG if data_aug_conf is None else data_aug_conf)
        norm = dict(DEFAULT_NORM if img_norm_cfg is None else img_norm_cfg)
        assert tuple(self.conf["rot_lim"]) == (0.0, 0.0), "Rotation is not currently supported"      # ref custom_pipeline.py:69
        self.mean = np.array(norm["mean"], dtype=np.float32)
        # mmcv.imnormalize: stdinv = 1 / np.float64(std); cv2 applies the scalar in the image's float32
        self.stdinv = (1.0 / np.float64(np.array(norm["std"], dtype=np.float32))).astype(np.float32)
        self.to_rgb = bool(norm.get("to_rgb", False))
        self.dev = torch.device(device)
        self.out_dtype = out_dtype
        self.rng = rng
        self._tables = {}

    # ---------------------------------------------------------------------------------------------- device pieces
    def _table(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            if len(self._tables) > 64:
                self._tables.clear()
            b, k, ks = pil_resample_coeffs(in_size, out_size, "bicubic")
            self._tables[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(k).to(self.dev), ks, b)
        return self._tables[key]

    def _resize_crop(self, src, resize_dims, crop, flip, out=None):
        """Image.resize(resize_dims) -> crop(crop) [-> FLIP_LEFT_RIGHT] of a device uint8 (H,W,3) image.  out=None: returns the
        uint8 (h,w,3) result; else out = (canvas (3,padH,padW) view, padH, padW) gets the normalised planar image."""
        lib = _lib.require_device()
        H, W = int(src.shape[0]), int(src.shape[1])
        newW, newH = resize_dims
        x0, y0, x1, y1 = crop
        if not (0 <= x0 < x1 <= newW and 0 <= y0 < y1 <= newH):
            raise ValueError("crop %s outside the resized image %s (PIL would pad with black; the reference config never does)" % (crop, resize_dims))
        bh, kh, ksh, _ = self._table(W, newW)
        bv, kv, ksv, bv_host = self._table(H, newH)
        row0 = int(bv_host[y0:y1, 0].min())
        row1 = int((bv_host[y0:y1, 0] + bv_host[y0:y1, 1]).max())
        outw, outh = x1 - x0, y1 - y0
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        tmp = torch.empty((row1 - row0, outw, 3), dtype=torch.uint8, device=self.dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.far3d_image_resample_h(P(src), src.stride(0), H, W, P(tmp), P(bh), P(kh), ksh, row0, row1 - row0, x0, outw, st),
                   "far3d_image_resample_h")
        if out is None:
            dst = torch.empty((outh, outw, 3), dtype=torch.uint8, device=self.dev)
            _lib.check(lib.far3d_image_resample_v(P(tmp), row0, row1 - row0, outw, P(bv), P(kv), ksv, y0, outh, 1 if flip else 0, 0, P(dst), 0,
                                                  0, 0, None, None, 0, st), "far3d_image_resample_v")
            return dst
        canvas, padH, padW = out
        mean = self.mean.ctypes.data_as(ctypes.c_void_p)
        stdinv = self.stdinv.ctypes.data_as(ctypes.c_void_p)
        _lib.check(lib.far3d_image_resample_v(P(tmp), row0, row1 - row0, outw, P(bv), P(kv), ksv, y0, outh, 1 if flip else 0, 1, P(canvas),
                                              1 if canvas.dtype == torch.bfloat16 else 0, padH, padW, mean, stdinv, 1 if self.to_rgb else 0, st),
                   "far3d_image_resample_v")
        return None

    def _upload(self, img):
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(np.uint8(img)))      # ref: Image.fromarray(np.uint8(img))
        if img.dtype != torch.uint8:
            img = img.to(torch.uint8)
        img = img.to(self.dev, non_blocking=True)
        if img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3:
            img = img.contiguous()
        return img

    # ---------------------------------------------------------------------------------------------- the transform
    def plan(self, shapes):
        """Host part: per camera (portrait pre-stage | None, resize_dims, crop, flip, ida_mat), drawing from `rng` in the
        reference's order (ref custom_pipeline.py:70-129).  shapes: [(H, W)] of the raw images."""
        plans = []
        for H, W in shapes:
            pre, ida_f = None, None
            if H > W:           # portrait camera: ref custom_pipeline.py:71-92
                r_f, dims_f, crop_f = sample_augmentation_portrait(H, W)
                pre = (dims_f, crop_f)
                ida_f = ida_matrix(r_f, crop_f)
                H, W = crop_f[3] - crop_f[1], crop_f[2] - crop_f[0]
            resize, resize_dims, crop, flip, rotate = sample_augmentation(H, W, self.conf, self.rng)
            ida = ida_matrix(resize, crop, flip, rotate)
            if ida_f is not None:
                ida = ida @ ida_f
            plans.append((pre, resize_dims, crop, flip, ida))
        return plans

    @staticmethod
    def update_calibration(results, plans):
        """ref custom_pipeline.py:162,173-179: fold ida_mat into the intrinsics, rebuild cam2img / lidar2img."""
        N = len(plans)
        ida_mats = []
        for i, p in enumerate(plans):
            results["intrinsics"][i][:3, :3] = np.asarray(p[4]) @ results["intrinsics"][i][:3, :3]
            ida_mats.append(p[4].numpy().copy())
        results["cam2img"] = results["intrinsics"]
        results["lidar2img"] = [results["intrinsics"][i] @ results["extrinsics"][i] for i in range(N)]
        results["ida_mat"] = ida_mats
        return results

    def __call__(self, results):
        imgs = results["img"]
        N = len(imgs)
        plans = self.plan([(int(im.shape[0]), int(im.shape[1])) for im in imgs])
        shapes = [(p[2][3] - p[2][1], p[2][2] - p[2][0]) for p in plans]
        # 'same2max' (ref custom_pipeline.py:360): max() over the shape TUPLES (lexicographic), exactly like the reference;
        # mmcv.impad then requires every image to fit, which holds because all crops share final_dim
        padH, padW = max(shapes)
        assert all(s[0] <= padH and s[1] <= padW for s in shapes)
        out = torch.empty((N, 3, padH, padW), dtype=self.out_dtype, device=self.dev)
        for i, (pre, resize_dims, crop, flip, ida) in enumerate(plans):
            src = self._upload(imgs[i])
            if pre is not None:
                src = self._resize_crop(src, pre[0], pre[1], False)
            self._resize_crop(src, resize_dims, crop, flip, out=(out[i], padH, padW))
        self.update_calibration(results, plans)
        results["img"] = out
        results["ori_shape"] = [(s[0], s[1], 3) for s in shapes]
        results["img_shape"] = [(padH, padW, 3)] * N
        results["pad_shape"] = [(padH, padW, 3)] * N
        results["img_norm_cfg"] = dict(mean=self.mean, std=1.0 / self.stdinv, to_rgb=self.to_rgb)
        results["pad_fixed_size"], results["pad_size_divisor"] = "same2max", None
        return results
