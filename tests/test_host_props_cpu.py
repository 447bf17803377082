"""CPU: property tests (hypothesis) of the host-side index / table / metric arithmetic around the hot path."""
import math

import numpy as np
import pandas as pd
from hypothesis import given, settings, strategies as st

from far3d_amd import dist as fdist
from far3d_amd.data_pipeline import av2_metric as M
from far3d_amd.data_pipeline import resample, streaming


@settings(max_examples=60, deadline=None)
@given(n=st.integers(1, 400), world=st.integers(1, 9))
def test_contiguous_shards_cover_every_index_in_order(n, world):
    shards = [streaming.contiguous_shard(n, world, r) for r in range(world)]
    assert len({len(s) for s in shards}) == 1                       # equal length (padded by wrap-around)
    flat = [i for s in shards for i in s]
    assert flat[:n] == list(range(n)) and set(flat) == set(range(n))    # order kept: a scene's frames stay on one rank
    assert all(flat[n + j] == j % n for j in range(len(flat) - n))


@settings(max_examples=60, deadline=None)
@given(n=st.integers(0, 300), k=st.integers(1, 9))
def test_interval_test_order_is_a_permutation_by_residue(n, k):
    o = streaming.interval_test_order(n, k)
    assert sorted(o) == list(range(n))
    assert [i % k for i in o] == sorted(i % k for i in range(n))      # offset 0 first, then offset 1, ...


@settings(max_examples=40, deadline=None)
@given(ids=st.lists(st.integers(0, 5), min_size=1, max_size=60), split=st.integers(1, 4))
def test_sequence_group_flags_are_monotone_runs(ids, split):
    f = streaming.sequence_group_flags(ids, split)
    assert len(f) == len(ids) and f[0] == 0 and (np.diff(f) >= 0).all() and (np.diff(f) <= 1).all()
    base = streaming.sequence_group_flags(ids, 1)
    assert (np.diff(base) != 0).tolist() == [a != b for a, b in zip(ids[:-1], ids[1:])]     # a new group exactly where the scene changes
    for g in np.unique(f):                                          # a split group never straddles two scenes
        assert len({base[i] for i in np.nonzero(f == g)[0]}) == 1


@settings(max_examples=40, deadline=None)
@given(n_in=st.integers(2, 300), n_out=st.integers(1, 200), filt=st.sampled_from(["bicubic", "bilinear"]))
def test_resample_tables_are_normalised_and_inside_the_image(n_in, n_out, filt):
    bounds, coeffs, ksize = resample.pil_resample_coeffs(n_in, n_out, filt)
    assert bounds.shape == (n_out, 2) and coeffs.shape == (n_out, ksize)
    assert (bounds[:, 0] >= 0).all() and (bounds[:, 1] >= 1).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all() and (bounds[:, 1] <= ksize).all()
    one = 1 << resample.PRECISION_BITS
    sums = coeffs.sum(axis=1)
    assert (np.abs(sums - one) <= ksize).all()                      # each row is a partition of unity up to per-tap rounding
    for i in range(n_out):
        assert (coeffs[i, bounds[i, 1]:] == 0).all()


@settings(max_examples=30, deadline=None)
@given(world=st.integers(1, 8), ncam=st.integers(1, 9))
def test_camera_shards_partition_the_cameras(world, ncam):
    per, owned = fdist.camera_shards(ncam, world)
    assert per == math.ceil(ncam / world) and len(owned) == world and all(len(o) == per for o in owned)
    flat = [c for o in owned for c in o]
    assert [c for c in flat if c >= 0] == list(range(ncam))         # contiguous blocks in rank order, every camera exactly once
    assert all(c == -1 for c in flat[ncam:])                        # padding slots only at the tail (7 cameras -> 8 slots on 8 GPUs)


def _det_tables(rng, n, m):
    ctr = rng.uniform(-100, 100, size=(m, 3)) * np.array([1, 1, 0.02])
    dims = rng.uniform(0.5, 5, size=(m, 3))
    yaw = rng.uniform(-3, 3, size=m)
    q = lambda a: np.stack([np.cos(a / 2), 0 * a, 0 * a, np.sin(a / 2)], -1)
    cols = list(M.ORDERED_CUBOID_COL_NAMES)
    gts = pd.DataFrame(np.concatenate([ctr, dims, q(yaw)], 1), columns=cols)
    gts["num_interior_pts"] = rng.integers(0, 5, size=m).astype(float)
    k = rng.integers(0, m, size=n)
    dts = pd.DataFrame(np.concatenate([ctr[k] + rng.normal(0, 1.5, (n, 3)), dims[k] * rng.uniform(0.8, 1.2, (n, 3)), q(yaw[k] + rng.normal(0, 0.4, n))], 1), columns=cols)
    dts["score"] = rng.permutation(n) / (n + 1.0)
    for df in (dts, gts):
        df["log_id"], df["timestamp_ns"], df["category"] = "l", 7, "BUS"
    return dts, gts


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n=st.integers(1, 40), m=st.integers(1, 12))
def test_metric_is_bounded_and_independent_of_the_row_order(seed, n, m):
    rng = np.random.default_rng(seed)
    dts, gts = _det_tables(rng, n, m)
    cfg = M.DetectionCfg(categories=("BUS",))
    _, _, a, ra = M.evaluate(dts, gts, cfg)
    _, _, b, rb = M.evaluate(dts.sample(frac=1.0, random_state=seed % 97).reset_index(drop=True), gts.sample(frac=1.0, random_state=3).reset_index(drop=True), cfg)
    assert np.allclose(a.to_numpy(dtype=float), b.to_numpy(dtype=float)) and np.allclose(ra.to_numpy(dtype=float), rb.to_numpy(dtype=float))
    row = a.loc["BUS"]
    assert 0 <= row["AP"] <= 1 and 0 <= row["RECALL"] <= 1 and 0 <= row["CDS"] <= row["AP"] + 1e-9
    assert 0 <= row["ATE"] <= cfg.tp_threshold_m and 0 <= row["ASE"] <= 1 and 0 <= row["AOE"] <= round(math.pi, 3)     # tables are rounded to 3 decimals
    assert (M.wrap_angles(rng.uniform(-50, 50, size=20)) < math.pi + 1e-12).all()


@settings(max_examples=60, deadline=None)
@given(st.floats(0.3, 0.8), st.integers(0, 300), st.integers(0, 200), st.booleans(), st.floats(-25.0, 25.0))
def test_ida_matrix_is_scale_crop_flip_rotation_about_the_crop_centre(scale, left, top, flip, angle):
    """The augmentation homography in closed form (data_pipeline/preprocess.py) against its geometric meaning: a source pixel is
    scaled, moved into the crop window, mirrored inside the window when flipped, then rotated about the window's centre -- so the
    centre of the window is a fixed point of the rotation and a source pixel lands where the steps, applied one by one, put it."""
    from far3d_amd.data_pipeline.preprocess import ida_matrix
    w, h = 96, 64
    crop = (left, top, left + w, top + h)
    Mx = ida_matrix(scale, crop, flip, angle).double().numpy()
    s = float(np.float32(scale))
    th = math.radians(angle)
    for x, y in ((0.0, 0.0), (123.0, 45.5), (700.0, 333.0)):
        px, py = s * x - left, s * y - top                 # scale, then crop
        if flip:
            px = w - px                                    # mirror inside the window
        cx, cy = w / 2.0, h / 2.0
        dx, dy = px - cx, py - cy                          # rotate about the window centre
        want = (cx + math.cos(th) * dx + math.sin(th) * dy, cy - math.sin(th) * dx + math.cos(th) * dy)
        got = Mx @ np.array([x, y, 1.0])
        assert abs(got[0] - want[0]) < 1e-3 and abs(got[1] - want[1]) < 1e-3 and got[2] == 1.0


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 9000), st.integers(1, 8))
def test_query_shard_rows_partition_the_queries(A, world):
    """Query-sharded decoder: rank r owns rows [r * per, min((r + 1) * per, A)) -- disjoint, in order, covering every query once; the
    gathered (world * per, E) buffer therefore holds the A rows in order followed only by padding."""
    class _QS(fdist.QueryShard):
        def __init__(self, rank, world):
            self.rank, self.world = rank, world
    per = _QS(0, world).rows_per_rank(A)
    assert per % 4 == 0 and per * world >= A and (per - 4) * world < A + 4 * world
    seen = []
    for r in range(world):
        a0, a1 = min(r * per, A), min((r + 1) * per, A)
        assert 0 <= a1 - a0 <= per
        seen += list(range(a0, a1))
    assert seen == list(range(A))


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 7), st.integers(1, 8), st.integers(1, 600), st.integers(1, 4096))
def test_capacity_block_rows_bound_the_exchange(num_cams, world, cam_cap, capacity):
    """Camera-sharded fixed-capacity mode: a rank's record block has min(capacity, per * cam_cap) rows -- never more than the
    global capacity (what the head can hold) nor more than its cameras can select."""
    per, shards = fdist.camera_shards(num_cams, world)
    block = min(capacity, per * cam_cap)
    assert 1 <= block <= capacity and block <= per * cam_cap
    # every camera-major prefix of valid rows across the blocks fits the compaction's destination or raises the overflow flag
    total_possible = sum(min(block, sum(1 for c in s if c >= 0) * cam_cap) for s in shards)
    assert total_possible <= world * block
