"""GPU: device image pre-processing (C ABI far3d_image_resample_h / _v behind far3d_amd.data_pipeline.ImagePreprocessor) against
(a) the fixture the reference's own pipeline classes produced (tests/golden/far3d_data_contract.npz: resize / crop / flip /
portrait camera / normalise / pad) and (b) Pillow itself at the AV2 sensor size, 2048x1550 -> 640x960.  Integer resampling is
bit-exact, and (pixel - mean) * (1/std) is the same two float32 operations as mmcv.imnormalize: results must be EQUAL."""
import os

import numpy as np
import pytest
import torch

from far3d_amd import data_pipeline as dp
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z = np.load(os.path.join(ROOT, "tests", "golden", "far3d_data_contract.npz"))


@pytest.mark.parametrize("case", [0, 1])
def test_pipeline_matches_reference_fixture(hip_lib, case):
    p = "pre%d_" % case
    conf = dict(dp.preprocess.DEFAULT_AUG, final_dim=(64, 96), rand_flip=bool(Z[p + "flip"]))
    pre = dp.ImagePreprocessor(conf, device=DEV, rng=np.random.RandomState(int(Z[p + "seed"])))
    res = dict(img=[Z[p + "in%d" % k].astype(np.float32) for k in range(3)],        # what AV2LoadMultiViewImageFromFiles(to_float32) hands over
               intrinsics=[m.copy() for m in Z[p + "intr_in"]], extrinsics=[m.copy() for m in Z[p + "extr"]])
    res = pre(res)
    want = torch.from_numpy(Z[p + "img"]).permute(0, 3, 1, 2)
    got = res["img"].cpu()
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.equal(got, want), "max abs diff %.3e" % (got - want).abs().max().item()
    assert np.allclose(np.stack(res["lidar2img"]), Z[p + "lidar2img"], rtol=1e-12, atol=1e-12)
    assert [tuple(s) for s in res["pad_shape"]] == [tuple(s) for s in Z[p + "pad_shape"]]


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_sensor_size_matches_pillow(hip_lib, out_dtype):
    """7 cameras at the AV2 ring-camera size (one portrait), resize_lim (0.47, 0.55), final 640x960."""
    from PIL import Image
    rs = np.random.RandomState(2)
    shapes = [(1550, 2048)] * 6 + [(2048, 1550)]
    imgs = []
    for k, (h, w) in enumerate(shapes):
        yy, xx = np.mgrid[0:h, 0:w]
        base = 120 + 80 * np.sin(xx / 37.0 + k) * np.cos(yy / 23.0)
        imgs.append(np.clip(base[..., None] + rs.randint(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8))
    pre = dp.ImagePreprocessor(device=DEV, out_dtype=out_dtype, rng=np.random.RandomState(5))
    plans = dp.ImagePreprocessor(device=DEV, rng=np.random.RandomState(5)).plan(shapes)
    res = pre(dict(img=[torch.from_numpy(im).to(DEV) for im in imgs],            # raw uint8 images already on the device
                   intrinsics=[np.eye(4) for _ in shapes], extrinsics=[np.eye(4) for _ in shapes]))
    got = res["img"]
    assert tuple(got.shape) == (7, 3, 640, 960) and got.dtype == out_dtype
    mean = np.array([103.530, 116.280, 123.675], dtype=np.float32)
    stdinv = (1.0 / np.float64(np.array([57.375, 57.120, 58.395], dtype=np.float32))).astype(np.float32)
    for i, (pl, im) in enumerate(zip(plans, imgs)):
        pil = Image.fromarray(im)
        if pl[0] is not None:
            pil = pil.resize(pl[0][0]).crop(pl[0][1])
        pil = pil.resize(pl[1]).crop(pl[2])
        want = (np.array(pil).astype(np.float32) - mean) * stdinv
        want = torch.from_numpy(want).permute(2, 0, 1)
        if out_dtype == torch.bfloat16:
            want = want.to(torch.bfloat16)
        assert torch.equal(got[i].cpu(), want), "camera %d: max abs diff %.3e" % (i, (got[i].cpu().float() - want.float()).abs().max().item())
