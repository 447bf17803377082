"""GPU: BASELINE configs[4] -- independent scene streams, one per process (the reference's own multi-GPU scheme: one replica per
rank fed by a contiguous shard of the streaming order, ref tools/test.py:229-234, datasets/samplers/distributed_sampler.py:41-45;
per-stream memory reset at a scene change, detectors/far3d.py:252-257).  On the one-GPU test box the replicas are two processes
sharing cuda:0: their frames interleave on the device, one stream changes scene in the middle, and each stream must produce,
bit for bit, what it produces when it runs alone (no state is shared between replicas; no collective on the data path).
Also: `bench.py --gpus 2 --mode replicas` as the driver would run it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(mode):
    from far3d_amd import engine, weights
    z = np.load(os.path.join(GOLD, "far3d_small_seq.npz"))
    rc = json.loads(bytes(z["recipe"]).decode())
    spec = weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"])
    sd = weights.init_state_dict(spec, seed=rc["weight_seed"])
    cfg = engine.default_cfg(backbone=rc["backbone"], num_cams=rc["num_cams"], num_query=rc["num_query"], num_propagated=rc["num_propagated"],
                             memory_len=rc["memory_len"], topk_proposals=rc["topk_proposals"], proposal_topk=6)
    eng = engine.Far3DEngine(sd, cfg, device="cuda:0", precision="fp32")
    eng.use_graph = mode != "eager"
    eng.pipeline = mode == "pipeline"
    return eng, rc


def _stream_frames(rc, stream):
    """Stream 0: the golden sequence (scene change at frame 2) + steady frames; stream 1: another scene (other images), no change."""
    from far3d_amd import synth
    frames = []
    for fi in (0, 1, 2, 3, 3, 3):
        if stream == 0:
            frames.append(synth.recipe_frame(rc, fi))
        else:
            rc1 = dict(rc, data_seed=rc["data_seed"] + 7, scene_change_at=None)
            data, metas = synth.recipe_frame(rc1, fi)
            metas[0]["scene_token"] = "other-scene"
            frames.append((data, metas))
    return frames


def _run_stream(stream, mode, barrier=None):
    eng, rc = _engine(mode)
    outs = []
    for data, metas in _stream_frames(rc, stream):
        if barrier is not None:
            barrier.wait(timeout=300)          # both replicas issue frame k together: their kernels interleave on the device
        o = eng.forward_frame(data, metas)
        eng.wait_outputs()
        outs.append((o["all_cls_scores"].cpu().numpy().copy(), o["all_bbox_preds"].cpu().numpy().copy(),
                     {k: v.cpu().numpy().copy() for k, v in eng.mem.items()}))
    return outs


def _worker(stream, mode, barrier, q):
    q.put((stream, _run_stream(stream, mode, barrier)))


@pytest.mark.parametrize("mode", ["pipeline"])      # graphs + frame pipeline; the scene starts inside it run eagerly
def test_two_replica_streams_on_one_gpu_equal_their_solo_runs(hip_lib, mode):
    import torch.multiprocessing as mp
    solo = {s: _run_stream(s, mode) for s in (0, 1)}
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q, barrier = ctx.Queue(), ctx.Barrier(2)
    procs = [ctx.Process(target=_worker, args=(s, mode, barrier, q)) for s in (0, 1)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=600) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    for s in (0, 1):
        assert len(res[s]) == len(solo[s])
        for fi, (a, b) in enumerate(zip(res[s], solo[s])):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "stream %d frame %d differs from its solo run" % (s, fi)
            for k in a[2]:
                assert np.array_equal(a[2][k], b[2][k]), "stream %d frame %d: memory '%s'" % (s, fi, k)
    # the two streams are different computations (so equality above is not vacuous), and stream 0 did reset at its scene change
    assert not np.array_equal(res[0][1][0], res[1][1][0])


def test_bench_replicas_mode_two_ranks(hip_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "replicas", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-in-tolerance", "--allow-shared-gpu"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["scaling"] == "weak"
    assert "independent scene streams" in line["config"]["parallelism"]
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] * 1e-3 - 2.0) < 1e-6      # two samples per step: one per replica
