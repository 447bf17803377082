#!/usr/bin/env python
"""Headline benchmark: samples/sec (one sample = one 7-camera frame, end to end) at VoV-99, 640x960 (BASELINE.json).

  python bench.py --gpus 1 --steps 100 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full frame through the HIP engine: 7 x (3,640,960) synthetic images resident in HBM -> VoV-99 -> FPN ->
2D head/depth -> 644 adaptive + 644 learned + 256 propagated queries (A = 1544) -> 6-layer decoder (self-attention over
2312 keys, fused perspective-aware aggregation, FFN) -> heads -> streaming-memory update -> top-300 box decode on device.
N > 1 (default --mode sharded): the cameras of the SAME sample are sharded across ranks (strong scaling), one RCCL all-gather
of the value maps; --mode replicas: one independent scene stream per GPU (BASELINE configs[4]; weak scaling, no collective).
Prints ONE JSON line on rank 0.

Headline (`value`, `dtype`, `ms_per_step`): the engine that MEETS the north-star logit tolerance (bf16x3: every
conv product as three bf16 MFMAs on pair-stored activations, exact-fp32 decoder; `parity.meets_tolerance` true) under the REFERENCE's
timing protocol (tools/analysis_tools/benchmark.py:84-111): K frames, a device sync (and a barrier for N > 1) before
and after EVERY frame, between the barrier + sync that bracket the timed region -- `value` = K / that time.  Side blocks of the same
run: `protocol.pipelined` (the same engine, K frames issued back to back with the frame pipeline on: throughput of a stream),
`protocol.sync_per_frame_groups` (the frame's cameras as two groups on parallel streams), and `fast_mode` (the bf16 engine -- BASELINE
configs[1]'s dtype, 40x outside the tolerance -- under both protocols; `--no-fast-mode` skips it).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# The frame pipeline keeps 3 camera streams + a head stream busy and this process builds two engines one after the other: with HIP's
# default of 4 hardware queues the second engine's streams end up sharing queues (in_tolerance 79.5 instead of 85.9 samples/s,
# profiles/r4/hw_queues_ab.txt).  Must be set before the HIP runtime initialises; far3d_amd.lib sets the same default on import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK = {"bf16": 2.5e15, "fp32": 157.3e12,     # dense FLOP/s (same guide)
             "bf16x3": 2.5e15 / 3}                 # split mode: three bf16 MFMAs per useful product
BACKBONE_FLOP_PER_CAM = 2 * 201.87e9   # VoV-99 @ 640x960
AGG_KERNEL = "aggregate_v8_kernel"
K_PROP = 92                # adaptive-query count fixed at 644 = 7 x 92 (static shapes)


def agg_algorithmic_bytes(N, S, C, A, P, G, L, ev):
    """value maps once + unreplicated sampling points + attention weights + camera-summed output."""
    return N * S * C * ev + A * P * N * 2 * 4 + N * A * G * L * P * 4 + A * C * 4


def oracle_frames(nframes):
    """The oracle (CPU port of the reference path, proven equal to the reference's own files by tools/gen_golden.py) on this
    box's host cores: frames 0.. of the benchmark sequence, ALL 7 cameras, each timed (frame 0 is the bounded CPU-baseline sample;
    frame 1 has the streaming memory in use).  Returns (cpu_baseline block, per-frame oracle outputs)."""
    from far3d_amd import synth, weights
    from oracle import far3d_oracle
    ncpu = os.cpu_count() or 1
    # torch's CPU convs oversubscribe badly on many-core hosts: pick the fastest thread count on a small probe
    probe_x, probe_w = torch.randn(1, 128, 80, 120), torch.randn(128, 128, 3, 3)
    best = (float("inf"), 1)
    for th in sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu}):
        torch.set_num_threads(th)
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        best = min(best, (time.perf_counter() - t0, th))
    cores = best[1]
    torch.set_num_threads(cores)
    sd = weights.init_state_dict(weights.detector_spec("V-99-eSE"), seed=0)
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=K_PROP))
    times, outs = [], []
    with torch.no_grad():
        for fi in range(nframes):
            data, metas = synth.make_frame(7, (640, 960), seed=0, frame_index=fi, ego_motion=True)
            t0 = time.perf_counter()
            o = orc.simple_test(data, metas)
            times.append(time.perf_counter() - t0)
            sel = [(int(n), int(i)) for n, i, _ in o["roi"]["valid_indices"].nonzero().numpy()]
            nq = 644
            outs.append(dict(logits=o["all_cls_scores"], sel=sel, ref=o["reference_points"].reshape(-1, 3)[nq:nq + len(sel)].clone(),
                             score2d=o["roi"]["bbox2d_scores"].reshape(-1)[:len(sel)].clone()))
    baseline = dict(value=1.0 / times[0], unit="samples/s", cores=cores, kind="port",
                    sample="1 frame of the benchmark workload, all 7 cameras, whole path (VoV-99+FPN+2D head+FarHead+decoder+decode) "
                           "timed once: %.2f s (second, streaming frame: %.2f s); torch %s fp32, %d threads (fastest of a probe; host has "
                           "%d logical CPUs)" % (times[0], times[1] if len(times) > 1 else float("nan"), torch.__version__, cores, ncpu))
    return baseline, outs


def parity_block(precision, eng_frames, orc_frames):
    """The engine's first frames (frame 0 eager, frame 1 = the captured hipGraph's first replay) against the oracle's logits."""
    par = []
    for fi, (got, o) in enumerate(zip(eng_frames, orc_frames)):
        # adaptive queries are comparable row by row only where both picked the same 2D peak: map them through (camera, cell)
        want_sel = o["sel"]
        pos = {k: j for j, k in enumerate(want_sel)}
        nq = 644
        rows_g, rows_w = list(range(nq)), list(range(nq))
        disc_g, disc_w = [], []
        common = discrete = 0
        for j, k in enumerate(got["sel"]):
            if k in pos:
                common += 1
                # the same 2D peak can still be a DIFFERENT query: its depth bin (an argmax over 51 bins, farhead.py:736-766) or its 3x3 peak
                # test (an equality on scores, yolox_head.py:429-438) fell the other way on the two sides -- a whole-row discrete difference,
                # not an arithmetic error.  Seen in the reference point (a bin is metres along the ray) / in the 2D score being zero or not
                if got.get("ref") is not None and o.get("ref") is not None:
                    jw = pos[k]
                    if (got["ref"][j] - o["ref"][jw]).abs().max().item() > 1e-3 or (got["score2d"][j].item() > 0) != (o["score2d"][jw].item() > 0):
                        discrete += 1
                        disc_g.append(nq + j); disc_w.append(nq + jw)
                        continue
                rows_g.append(nq + j); rows_w.append(nq + pos[k])
        A = got["logits"].shape[2]
        M = len(want_sel)
        rows_g += list(range(nq + len(got["sel"]), A)); rows_w += list(range(nq + M, A))
        d = (got["logits"][:, 0, rows_g] - o["logits"][:, 0, rows_w]).abs()
        d_disc = (got["logits"][:, 0, disc_g] - o["logits"][:, 0, disc_w]).abs().max().item() if disc_g else 0.0
        qs = torch.quantile(d.flatten()[::2].double(), torch.tensor([0.5, 0.99], dtype=torch.float64))
        par.append(dict(frame=fi, logit_max_abs=d.max().item(), logit_mean_abs=d.mean().item(), logit_p50_abs=qs[0].item(), logit_p99_abs=qs[1].item(),
                        last_layer_logit_max_abs=d[-1].max().item(), proposals_in_common=common, proposals=M,
                        rows_compared=len(rows_g), rows_excluded=(A - len(rows_g)), rows_excluded_same_peak_other_bin_or_peak_test=discrete,
                        logit_max_abs_incl_same_peak_rows=max(d.max().item(), d_disc),
                        rows_excluded_what="adaptive-query rows whose 2D peak (camera, cell) the other side did not select, or selected with another "
                                           "depth bin / peak-test outcome: different queries, not compared",
                        engine_path=got["path"]))
    # rows left out as "a different query on the two sides" must stay a handful: a regression of the depth decode / un-projection /
    # peak test that moves MANY reference points would otherwise be filtered out of the comparison and pass
    cap = max(4, 644 // 100)
    excluded_ok = all(p["rows_excluded_same_peak_other_bin_or_peak_test"] <= cap for p in par)
    # Frame 0 is the clean comparison.  From frame 1 on the two sides also differ through the streaming memory: which 256 queries
    # are kept is a discrete top-k on scores ~1e-4 apart, so any rounding difference (let alone bf16) changes the memory contents
    # and the frames stop being the same computation (tests/test_engine_full_gpu.py quantifies this with an fp64 oracle).
    return dict(precision=precision, tolerance_north_star=1e-3, checker="oracle (fp32 CPU port of the reference path), same seeded weights/inputs",
                logit_max_abs=par[0]["logit_max_abs"], logit_mean_abs=par[0]["logit_mean_abs"],
                # every frame this block reports must be inside the bar, not only the first
                meets_tolerance=bool(excluded_ok and all(p["logit_max_abs"] < 1e-3 for p in par)),
                meets_tolerance_frame0=bool(excluded_ok and par[0]["logit_max_abs"] < 1e-3),
                excluded_rows_cap=cap, excluded_rows_within_cap=bool(excluded_ok),
                headline_frame=0, frames=par,
                see="tests/test_engine_full_gpu.py (per-stage budget, fp64 yardstick), DESIGN.md section 4")


def agg_traffic(fp32_rows=False):
    """HBM bytes per launch of the aggregation kernel from the committed in-frame PMC passes (profiles/<round>/aggregate_pmc.json for
    bf16 value rows, aggregate_pmc_fp32rows.json for the fp32 rows of the in-tolerance engine: FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    separate rocprofv3 --pmc passes over `bench.py --eager`, tools/evidence_run.sh with PMC=1; the newest round that holds the file).
    Counters cannot be read from inside this process; the file carries the kernel name and the commit it was taken at, and a figure
    for another kernel is not reported."""
    name = "aggregate_pmc_fp32rows.json" if fp32_rows else "aggregate_pmc.json"
    for rnd in ("r6", "r5", "r4"):
        p = os.path.join(ROOT, "profiles", rnd, name)
        try:
            with open(p) as f:
                j = json.load(f)
            if j.get("kernel") != AGG_KERNEL:
                continue
            return j["hbm_bytes_per_launch"], ("profiles/%s/%s @ %s (in-frame rocprofv3 --pmc passes over `bench.py --eager`; a committed "
                                               "figure, NOT measured by the run that prints this line)" % (rnd, name, j.get("commit", "?")))
        except Exception:   # noqa: BLE001
            continue
    return None, None


def agg_traffic_live(fp32_rows, timeout_s=150):
    """HBM bytes per launch of the aggregation kernel measured BY THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
    passes, --kernel-trace only -- the guide's HBM section) over tools/probe/run_agg_once.py, which launches the kernel eight times on
    benchmark-size operands (tests/cases.config2_aggregate_case: 7 cameras, 12750 tokens, A = 1544; value rows in this mode's dtype).
    FETCH_SIZE x2 on gfx950 (the counter tallies 128-byte requests at 64 bytes), WRITE_SIZE as reported.  Returns (bytes, source) or
    (None, reason): no rocprofv3 on PATH, a pass that fails or times out -- the caller then falls back to the committed in-frame figure."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself being profiled (no nested profiler runs)"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="far3d_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", N_LAUNCH="8")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable,
                   os.path.join(ROOT, "tools", "probe", "run_agg_once.py"), "--camsort"] + (["--fp32"] if fp32_rows else [])
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s pass timed out after %d s" % (ctr, timeout_s)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            got = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if AGG_KERNEL in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        got.append(float(row["Counter_Value"]))
            if len(got) < 4:
                return None, "rocprofv3 --pmc %s pass: %d launches of %s in the counter file" % (ctr, len(got), AGG_KERNEL)
            got = got[2:]                                  # the first launches fill the caches
            vals[ctr] = sum(got) / len(got)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return by, ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, x2 fetch correction of gfx950) over launches "
                "3-8 of tools/probe/run_agg_once.py on benchmark-size operands, %s value rows; FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB" %
                ("fp32" if fp32_rows else "bf16", vals["FETCH_SIZE"], vals["WRITE_SIZE"]))


def measure(args, precision, steps, warmup, world, rank, dev, sharded, dist, kernel_timings=True, proposals="topk", latency_runner=True,
            two_streams=False):
    """Build an engine of `precision` and run, in this order: the parity frames; region A -- the reference's protocol, K frames with a
    device sync (+ barrier) around every one, on the engine in its single-frame configuration (one hipGraph per frame, no frame
    overlap, tiles tuned for launches alone); region B -- K frames back to back with the frame pipeline on; the camera-group latency
    runner; the per-kernel roofline timings.  Returns a dict of raw figures."""
    from far3d_amd import engine, ops, synth, weights
    from far3d_amd import dist as fdist
    # proposals: "topk" = 92 best peaks per camera (static, SURVEY 8(d): 644 adaptive queries); "threshold" = the reference's rule
    # (every peak with score > 0.1, ref yolox_head.py:429-458) in fixed-capacity form: args.capacity rows, count on the device
    cfg = engine.default_cfg(proposal_topk=K_PROP) if proposals == "topk" else engine.default_cfg(proposal_topk=None, proposal_capacity=args.capacity)
    spec = weights.detector_spec(cfg["backbone"])
    sd = weights.init_state_dict(spec, seed=0)
    eng = engine.Far3DEngine(sd, cfg, device=dev, precision=precision)
    eng.agg_variant = args.agg_variant
    if args.agg_split is not None:
        eng.agg_split_extra = args.agg_split
    eng.cam_priority = args.cam_priority
    eng.fused_rows = args.fused_rows        # row-resident decoder chains (bf16 decoder only; csrc/rowchain.hip); --no-fused-rows for A/B
    eng2 = None
    if two_streams and world == 1 and not sharded and proposals == "topk" and not args.eager and not args.no_pipeline:
        # a SECOND scene stream on the same device (configs[4]'s batch of independent streams, ref tools/test.py:229-234):
        # its own engine = its own buffers, memory queue, streams and hipGraphs, the same weights
        eng2 = engine.Far3DEngine(sd, cfg, device=dev, precision=precision)
        eng2.agg_variant, eng2.cam_priority, eng2.fused_rows = eng.agg_variant, args.cam_priority, args.fused_rows
        if args.agg_split is not None:
            eng2.agg_split_extra = args.agg_split
    del sd
    frames = []
    for fi in range(4):   # a few distinct frames (ego motion on), resident in HBM before the timed region
        frames.append(synth.make_frame(7, (640, 960), seed=0 if not (world > 1 and not sharded) else rank, frame_index=fi, device=dev,
                                       ego_motion=True))
    table = {"auto": None, "latency": "tuning_mi355x.json", "tput": "tuning_mi355x_tput.json"}[args.tile_table]
    eng.pipeline_sets = args.pipeline_sets
    eng.cam_streams = args.cam_streams

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---------------------------------------------------------------- region A: the reference's protocol (sync around every frame)
    # single-frame configuration: the whole steady-state frame as ONE hipGraph (per rank: camera graph, exchange, head graph), no frame
    # overlap, tiles tuned for a launch running alone
    eng.tile_table = table if table is not None else "tuning_mi355x.json"
    eng.pipeline = False
    if sharded:
        runner = fdist.ShardedFrame(eng, use_graph=not args.eager, pipeline=False)
        runner.record_stage_times = True
    else:
        runner = eng
        eng.use_graph = not args.eager
    # the first two frames of the sequence are kept for the parity check (frame 0: eager; frame 1: first graph replay)
    eng_frames = []
    for fi in range(2):
        o = runner.forward_frame(*frames[fi])
        runner.wait_outputs()
        cnt = o["sel_cnt"].cpu().numpy() if "sel_cnt" in o else None
        sel = [(n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()] if cnt is not None else []
        eng_frames.append(dict(logits=o["all_cls_scores"].float().cpu().clone(), sel=sel,
                               ref=o["reference_points"].float().reshape(-1, 3)[644:644 + len(sel)].cpu().clone() if "reference_points" in o else None,
                               score2d=o["bbox2d_scores"].float().reshape(-1)[:len(sel)].cpu().clone() if "bbox2d_scores" in o else None,
                               path="eager (first frame of the scene)" if fi == 0 or args.eager else "hipGraph replay"))
    step_i = 2
    for _ in range(max(1, warmup)):
        runner.forward_frame(*frames[step_i % len(frames)])
        step_i += 1
    if sharded:
        runner.stage_times.clear()
    sync()
    per_frame = []
    t0 = time.perf_counter()
    for k in range(steps):
        t1 = time.perf_counter()
        out = runner.forward_frame(*frames[step_i % len(frames)])
        sync()
        per_frame.append((time.perf_counter() - t1) * 1e3)
        step_i += 1
    dt_sync = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_sync], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_sync = t.item()
    per_frame.sort()
    stage_ms = None
    if sharded:       # where a rank's frame goes (DESIGN.md section 7's table): mean over the timed frames, gathered from every rank
        mine = runner.mean_stage_times()
        allr = [None] * world
        dist.all_gather_object(allr, dict(rank=rank, cameras=len(runner.cams), **mine))
        stage_ms = allr
    A = out["all_cls_scores"].shape[2]
    N, S, C = out["feat_flatten"].shape
    lg = out["all_cls_scores"]          # fixed-capacity proposal mode: the rows without a query carry -inf logits by construction
    assert not torch.isnan(lg).any() and not (lg == float("inf")).any(), "non-finite logits"

    # ---------------------------------------------------------------- region B: frames back to back, frame pipeline on
    # frames of one stream are software-pipelined: the per-camera stages of the next frames overlap the head of frame i (engine.py);
    # tiles tuned under that concurrency unless a table was asked for
    pipe = not args.no_pipeline and not args.eager
    eng.tile_table = table
    if sharded:
        runner = fdist.ShardedFrame(eng, use_graph=not args.eager, pipeline=pipe)
    else:
        eng.pipeline = pipe
    # every buffer set of the frame pipeline captures its hipGraphs on its first steady frame (a device sync each): keep that out
    # of the timed region whatever --warmup says
    for _ in range(eng.pipeline_sets + 1 if pipe else 1):
        runner.forward_frame(*frames[step_i % len(frames)])
        step_i += 1
    for _ in range(warmup):
        runner.forward_frame(*frames[step_i % len(frames)])
        step_i += 1
    sync()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(runner.output_stream())
    for k in range(steps):
        out = runner.forward_frame(*frames[step_i % len(frames)])
        evs[k + 1].record(runner.output_stream())   # completion of frame k (the head's stream in pipeline mode)
        step_i += 1
    sync()
    dt = time.perf_counter() - t0
    dev_ms = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))
    if world > 1:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    pipeline_on = bool(runner.pipeline)
    frames_in_flight = int(eng.pipeline_sets) if pipeline_on else 1
    camera_streams = len(runner._pipe["s_cams"]) if (pipeline_on and getattr(runner, "_pipe", None)) else 1
    tile_table_pipelined = eng.bf16_tile_table()
    sync()

    # ---- two independent scene streams interleaved on the one device (frames of stream A and stream B alternate; each stream is the
    # frame pipeline of region B with its own buffers and memory queue).  Reported BESIDE the single-stream figures, never instead.
    two = None
    if eng2 is not None and pipeline_on:
        try:
            eng2.pipeline_sets, eng2.cam_streams, eng2.tile_table, eng2.pipeline, eng2.use_graph = eng.pipeline_sets, eng.cam_streams, table, True, True
            frames_b = [(d, [dict(m, scene_token="bench-stream-b") for m in metas]) for d, metas in frames]
            sb = 0
            for _ in range(eng2.pipeline_sets + 3 + warmup):          # scene start (eager), the captures of every buffer set, warm-up
                eng2.forward_frame(*frames_b[sb % len(frames_b)])
                sb += 1
            sync()
            half = max(1, steps // 2)
            t0 = time.perf_counter()
            for k in range(half):
                oa = eng.forward_frame(*frames[step_i % len(frames)])
                ob = eng2.forward_frame(*frames_b[sb % len(frames_b)])
                step_i += 1
                sb += 1
            eng.wait_outputs()
            eng2.wait_outputs()
            sync()
            dt2 = time.perf_counter() - t0
            assert not torch.isnan(oa["all_cls_scores"]).any() and not torch.isnan(ob["all_cls_scores"]).any()
            two = {"streams": 2, "frames": 2 * half, "samples_per_s": 2 * half / dt2, "ms_per_frame": dt2 / (2 * half) * 1e3,
                   "single_stream_pipelined_samples_per_s": steps / dt,
                   "what": "two independent scene streams (two engines: own buffers, memory queues, streams, hipGraphs; same weights), "
                           "their frames enqueued alternately, each stream with the frame pipeline of protocol.pipelined"}
        except Exception as e:   # noqa: BLE001  (a side block must never cost the headline)
            two = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
            torch.cuda.synchronize(dev)
    if eng2 is not None:
        del eng2
        torch.cuda.empty_cache()

    # ---- the reference's protocol again with the frame's cameras split into groups that run side by side (far3d_amd.latency; --latency-groups 0 skips it)
    lat_groups = None
    if latency_runner and args.latency_groups > 1 and world == 1 and not sharded and proposals == "topk" and not args.eager:
        from far3d_amd.latency import CameraGroupFrame
        sync()
        was, was_table = eng.pipeline, eng.tile_table
        try:
            eng.pipeline, eng.tile_table = False, (table if table is not None else "tuning_mi355x.json")
            lat = CameraGroupFrame(eng, groups=args.latency_groups, use_graph=not args.eager)
            for _ in range(4):                                   # one eager frame (buffers), the captures, two replays
                lat.forward_frame(*frames[step_i % len(frames)])
                step_i += 1
            pf = []
            for _ in range(min(steps, 50)):
                sync()
                t1 = time.perf_counter()
                out_l = lat.forward_frame(*frames[step_i % len(frames)])
                sync()
                pf.append((time.perf_counter() - t1) * 1e3)
                step_i += 1
            pf.sort()
            assert not torch.isnan(out_l["all_cls_scores"]).any()
            lat_groups = {"groups": [list(b) for b in lat.blocks], "frames": len(pf), "mean_ms": sum(pf) / len(pf), "p50_ms": pf[len(pf) // 2],
                          "samples_per_s_mean": 1e3 * len(pf) / sum(pf),
                          "what": "sync before and after every frame; the frame's per-camera stages as camera groups on parallel streams "
                                  "(far3d_amd.latency.CameraGroupFrame), head after all groups"}
        except Exception as e:   # noqa: BLE001  (a side block must never cost the headline)
            lat_groups = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
            torch.cuda.synchronize(dev)
        eng.pipeline, eng.tile_table = was, was_table

    # Kernel durations for the rooflines: HIP events (on the launch stream) around hipGraph replays that contain ONLY
    # that kernel, fed with this run's live tensors (last decoder layer's operands / the staged images) -- a launch-gap-free
    # device time that agrees with rocprofv3's per-kernel average (profiles/).
    def device_time(fn, iters, reps=3):
        fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        # thread_local: with N > 1 the RCCL watchdog thread keeps polling its events while this thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(iters):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / (reps * iters)

    agg_ms, bb_ms, timing_error = [], [], None
    try:        # the per-kernel figures must never cost the headline line (e.g. a capture problem on one rank of N)
        img_in = eng._ins[0]["img"]
        img_local = img_in if not sharded else img_in[runner.cams[0]:runner.cams[-1] + 1] if runner.cams else img_in[:0]
        if kernel_timings:
            tokens, ref_, offs_, l2i_, U_, Vc_, hw_, st_, pad_, perm_, tab_, qbase_ = eng.last_agg
            agg_out = torch.empty(ref_.shape[0], 256, device=dev, dtype=eng.prec["dec"])
            agg_ms = [device_time(lambda: ops.aggregate_forward(tokens, ref_, offs_, l2i_, U_, Vc_, hw_, st_, cfg["pc_range"], pad_,
                                                                num_groups=cfg["num_groups"], perm=perm_, out=agg_out,
                                                                variant=args.agg_variant, tables=tab_, split=getattr(eng, "last_agg_split", None), qbase=qbase_), 24)]
        # the backbone ALONE, exactly the launch sequence of the frame (keep_stage2=False: the stage-2 map is consumed by the pooling
        # pass only): launches one after the other, so with the tile table tuned for that regime
        keep_table, eng.tile_table = eng.tile_table, "tuning_mi355x.json"
        bb_ms = [device_time(lambda: eng.backbone(img_local, keep_stage2=False), 2)] if img_local.shape[0] > 0 else []
        eng.tile_table = keep_table
    except Exception as e:   # noqa: BLE001
        timing_error = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
    n_adapt = int(out["num_adaptive_dev"].item()) if out.get("num_adaptive_dev") is not None else int(out["num_adaptive"])
    overflow = bool(int(out["proposal_overflow"].item())) if out.get("proposal_overflow") is not None else False
    res = dict(n_adaptive=n_adapt, proposal_overflow=overflow, steps=steps, dt=dt, dt_sync=dt_sync, dev_ms=dev_ms, per_frame=per_frame, lat_groups=lat_groups,
               two_streams=two,
               A=A, N=N, S=S, C=C, eng_frames=eng_frames, agg_ms=agg_ms, bb_ms=bb_ms, stage_ms=stage_ms,
               timing_error=timing_error, prec=dict(eng.prec), pipeline=pipeline_on, cfg=cfg, tile_table=tile_table_pipelined,
               ncam_local=len(runner.cams) if sharded else 7, agg_split_extra=int(getattr(eng, "agg_split_extra", 0)),
               # what the runner actually did (read back, not assumed): frames in flight and camera streams of its pipeline
               frames_in_flight=frames_in_flight, camera_streams=camera_streams)
    del eng, runner, frames, out
    torch.cuda.empty_cache()
    return res


def agg_roofline(args, res, traffic=None, traffic_src=None, live=False, committed=None, committed_src=None):
    """roofline block of the aggregation kernel: SURVEY.md 8(d)'s algorithmic bytes of one launch (all queries x all cameras of one
    decoder layer, value rows in the mode's value dtype) over the launch time measured live in this run."""
    cfg = res["cfg"]
    evb = 2 if res["prec"]["value"] == torch.bfloat16 else 4
    by = agg_algorithmic_bytes(res["N"], res["S"], res["C"], res["A"], cfg["num_pts"], cfg["num_groups"], cfg["num_levels"], evb)
    agg_t = (sum(res["agg_ms"]) / len(res["agg_ms"])) * 1e-3 if res["agg_ms"] else float("nan")
    return {"kernel": AGG_KERNEL + " (fused perspective-aware aggregation, one launch per decoder layer)" + (" + %d sibling workgroups for heavy queries (variant 9)" % res["agg_split_extra"] if res.get("agg_split_extra") else "") if args.agg_variant in (0, 8, 9)
            else "aggregate_v%d_kernel (A/B variant %d)" % (3 if args.agg_variant == 3 else 7, args.agg_variant),
            "bound": "hbm", "achieved": by / agg_t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": by / agg_t / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
            "traffic_measured_in_this_run": bool(live) if traffic is not None else None,
            "traffic_in_frame_committed": committed, "traffic_in_frame_committed_source": committed_src,
            "value_row_bytes": 256 * evb, "algorithmic_bytes_per_launch": by,
            "avg_launch_us": agg_t * 1e6, "timing": "HIP events around a hipGraph of 24 launches x 3 replays, live frame operands"}


def backbone_roofline(res):
    """roofline_backbone block: 2 * 201.87 GMAC per camera over the measured backbone time, against the dense
    MFMA peak of the mode's arithmetic."""
    prec = res["prec"]
    mode = prec.get("mma") or ("bf16" if prec["act"] == torch.bfloat16 else "fp32")
    peak = MFMA_PEAK[mode]
    bb_t = (sum(res["bb_ms"]) / len(res["bb_ms"])) * 1e-3 if res["bb_ms"] else float("nan")
    n = res["ncam_local"]
    kernels = {"bf16": "conv3x3_pipe_kernel + gemm1x1_pipe_kernel <NT=1> (bf16 MFMA)",
               "bf16x3": ("conv3x3_pipe_kernel + gemm1x1_pipe_kernel <NT=3, PAIR> (pair-stored activations, 3 bf16 MFMAs per product)"
                          if prec.get("pair") else "igemm_kernel<float, split_t> (register-staged, 3 bf16 MFMAs per product)"),
               "fp32": "igemm_kernel<float, float> (exact fp32 MFMA)"}[mode]
    return {"kernel": kernels + " -- VoV-99 backbone: all launches incl. eSE / pooling", "bound": "mfma",
            "achieved": n * BACKBONE_FLOP_PER_CAM / bb_t / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
            "regime": "the backbone's launches alone, one after the other (tile table tuned for launches alone)",
            "frac": n * BACKBONE_FLOP_PER_CAM / bb_t / peak, "backbone_ms": bb_t * 1e3, "cameras_on_this_rank": n,
            "peak_what": "dense bf16 MFMA 2.5 PF" + (" / 3 (three MFMAs per useful product)" if mode == "bf16x3" else "") if mode != "fp32" else "fp32 MFMA"}


def build_commit():
    """Commit the library was built from: git when available, else far3d_amd/_build_commit.txt (written by far3d_amd/build.py
    here, travels with the snapshot to boxes without .git), else FAR3D_COMMIT."""
    try:
        c = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if c:
            return c
    except Exception:   # noqa: BLE001
        pass
    try:
        with open(os.path.join(ROOT, "far3d_amd", "_build_commit.txt")) as f:
            return f.read().strip() or None
    except OSError:
        return os.environ.get("FAR3D_COMMIT")


def launch_plan(gpus, env, argv):
    """The command that starts the N ranks of `bench.py --gpus N`, or None when this process already is one of them (or N == 1).
    One process per GPU under torch.distributed.run on a free local port, rendezvous on 127.0.0.1 (the container hostname may
    not resolve).  A WORLD_SIZE that disagrees with --gpus is a caller error."""
    if "WORLD_SIZE" in env:
        if int(env["WORLD_SIZE"]) != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s (the launcher and the flag must agree)" % (gpus, env["WORLD_SIZE"]))
        return None
    if gpus <= 1:
        return None
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "fp32", "bf16x3", "bf16x3_all", "bf16x3_2d1", "bf16x3_f32act", "bf16_fp32dec", "bf16_fp32val"])
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicas"], help="N>1: shard one sample's cameras, or one scene stream per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-streams", dest="two_streams", action="store_false",
                    help="single GPU: skip protocol.two_streams_pipelined (a second engine = a second, independent scene stream interleaved with the first)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc passes (two short sub-runs); the "
                    "committed in-frame figure of the evidence set is reported instead")
    ap.add_argument("--no-fast-mode", "--no-in-tolerance", dest="fast_mode", action="store_false",
                    help="skip the second engine of the default run (the bf16 `fast_mode` block); --no-in-tolerance is the flag's old name")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying a hipGraph")
    ap.add_argument("--no-pipeline", action="store_true", help="single GPU: do not overlap the camera stages of frame i+1 with the head of frame i")
    ap.add_argument("--agg-variant", type=int, default=0, help="far3d_aggregate_forward kernel variant (A/B timing)")
    ap.add_argument("--agg-split", type=int, default=None, help="sibling workgroups for heavy queries in the aggregation kernel (variant 9; 0 = off; "
                    "default: the engine's setting)")
    ap.add_argument("--proposals", default="topk", choices=["topk", "threshold"],
                    help="adaptive queries: 92 best 2D peaks per camera (static 644), or the reference's score > 0.1 rule with a fixed capacity")
    ap.add_argument("--capacity", type=int, default=1024, help="--proposals threshold: rows reserved for the adaptive queries")
    ap.add_argument("--pipeline-sets", type=int, default=4, help="single GPU: frames in flight = buffer sets of the frame pipeline (4: the camera "
                    "stages of three frames run concurrently under the head of a fourth; 2: camera || head only)")
    ap.add_argument("--tile-table", default="auto", choices=["auto", "latency", "tput"],
                    help="bf16 conv tile table: tuned for a launch alone, or under the pipeline's 3-stream concurrency (auto: by mode; A/B)")
    ap.add_argument("--cam-streams", type=int, default=3, help="streams the camera stages of consecutive frames alternate between (A/B)")
    ap.add_argument("--cam-priority", type=int, default=-1, help="HIP stream priority of the camera-stage streams in pipeline mode (-1 = high, 0 = default; A/B)")
    ap.add_argument("--latency-groups", type=int, default=2, help="single GPU: also time the sync-per-frame protocol with the frame's cameras "
                    "split into this many groups on parallel streams (far3d_amd.latency; reported as protocol.sync_per_frame_groups beside the "
                    "plain engine's protocol.sync_per_frame = `value`; 0 = skip)")
    ap.add_argument("--no-fused-rows", dest="fused_rows", action="store_false", help="A/B: run the row-local parts of the decoder layers and the "
                    "cls / reg branches as separate GEMM / LayerNorm launches instead of the row-resident chains (engine.fused_rows, the default "
                    "since round 5; bf16 decoder only)")
    ap.add_argument("--fused-rows", dest="fused_rows", action="store_true", help="(default) kept so that round 4's command lines still parse")
    ap.set_defaults(fused_rows=True, fast_mode=True)
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="N ranks on fewer than N GPUs (test rig only): ranks share devices and exchange over gloo; the line says so and "
                         "is not a scaling measurement")
    args = ap.parse_args()

    import torch.distributed as dist

    # `python bench.py --gpus N` without a launcher starts its own N ranks (the reference's tools/dist_test.sh:11-23 wraps
    # torch.distributed.launch around tools/test.py the same way); under torch.distributed.run the environment already says N
    plan = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if plan is not None:
        raise SystemExit(subprocess.call(plan))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    shared_gpu = False
    if world > 1 and ndev < world:
        if not (args.allow_shared_gpu and ndev > 0):
            # not a crash: say what is missing in the one JSON line the caller parses, and on stderr
            msg = ("bench.py --gpus %d needs %d visible GPUs, this box has %d: no multi-GPU figure was measured "
                   "(--allow-shared-gpu runs the ranks on shared devices over gloo, for testing the launcher only)" % (world, world, ndev))
            if rank == 0:
                print(msg, file=sys.stderr)
                print(json.dumps({"metric": "samples/sec (7-cam frame) end-to-end @ VoV-99 640x960", "value": None, "unit": "samples/s",
                                  "n_gpus": world, "visible_gpus": ndev, "error": msg}))
            return
        shared_gpu = True
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared_gpu else "nccl"      # "nccl" IS RCCL on ROCm; RCCL refuses two ranks on one device
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    sharded = world > 1 and args.mode == "sharded"

    res = measure(args, args.precision, args.steps, args.warmup, world, rank, dev, sharded, dist, proposals=args.proposals,
                  two_streams=args.two_streams and world == 1)
    # The headline engine is the one that MEETS the north-star logit tolerance (bf16x3).  The same run also times the bf16 engine --
    # BASELINE configs[1]'s dtype, ~40x outside the tolerance -- same workload, same two protocols, and reports it as `fast_mode`.
    res_fast = None
    if world == 1 and args.precision == "bf16x3" and args.fast_mode and not args.eager and args.proposals == "topk":
        res_fast = measure(args, "bf16", args.steps, args.warmup, world, rank, dev, sharded, dist, kernel_timings=True)

    if rank == 0:
        def rates(r):
            pf = r["per_frame"]
            return {"pipelined": {"frames": r["steps"], "mean_ms": r["dt"] / r["steps"] * 1e3, "samples_per_s": r["steps"] * rep / r["dt"],
                                  "p50_ms_device_events": r["dev_ms"][len(r["dev_ms"]) // 2], "p50_samples_per_s": 1e3 / r["dev_ms"][len(r["dev_ms"]) // 2],
                                  "what": "frames issued back to back, ONE sync at each end: the throughput of a stream of frames (not `value`)",
                                  "frame_overlap": r["pipeline"], "frames_in_flight": r["frames_in_flight"], "camera_streams": r["camera_streams"],
                                  "tile_table": r["tile_table"],
                                  "frame_overlap_what": "the per-camera stages of the next frames (one frame per camera stream, high priority) run "
                                                        "concurrently while the head of frame i is in flight (one buffer set per frame in flight, head "
                                                        "graphs ordered on one stream: results identical to the unpipelined engine)"},
                    "sync_per_frame": {"frames": len(pf), "mean_ms": sum(pf) / len(pf), "p50_ms": pf[len(pf) // 2],
                                       "samples_per_s_mean": 1e3 * len(pf) / sum(pf),
                                       "what": "reference protocol (tools/analysis_tools/benchmark.py:84-111): device sync before and after every frame "
                                               "-> `value`; one hipGraph per frame, no frame overlap"}}
        dt, dev_ms, per_frame, A, N, S, C, cfg = (res[k] for k in ("dt", "dev_ms", "per_frame", "A", "N", "S", "C", "cfg"))
        rep = world if (world > 1 and not sharded) else 1        # replicas: every rank ran its own stream of frames
        samples = args.steps * rep
        dt_sync = res["dt_sync"]
        # HBM traffic of the dominant kernel: measured by this run when rocprofv3 is there (isolated launches on benchmark-size operands),
        # with the committed IN-FRAME figure of the evidence set beside it; without rocprofv3 the committed figure alone, labelled
        fp32_rows = res["prec"]["value"] != torch.bfloat16
        committed, committed_src = agg_traffic(fp32_rows=fp32_rows)
        traffic, traffic_src, live = committed, committed_src, False
        if world == 1 and not args.no_pmc:
            tl, src = agg_traffic_live(fp32_rows)
            if tl is not None:
                traffic, traffic_src, live = tl, src, True
            else:
                traffic_src = "%s; live measurement unavailable: %s" % (committed_src, src)
        line = {
            "metric": "samples/sec (7-cam frame) end-to-end @ VoV-99 640x960",
            "value": samples / dt_sync, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_sync / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded or world == 1 else "weak",
            "value_protocol": "sync_per_frame (the reference's: a device sync" + (" + barrier" if world > 1 else "") + " around every one of the K frames)",
            # read back from the process group, not from the flag: how many ranks actually took part, and over which backend
            "rccl_ranks": dist.get_world_size() if world > 1 and backend == "nccl" else (1 if world == 1 else 0),
            "ranks": dist.get_world_size() if world > 1 else 1, "backend": backend or "none (single process)",
            "visible_gpus": ndev, "shared_gpu": shared_gpu,
            # 6.4 samples/s (hardware not stated), taken with a device sync around every frame like `value`
            "vs_baseline": (samples / dt_sync) / 6.4 if world == 1 else None,
            "dtype": args.precision if args.precision in ("fp32", "bf16x3", "bf16x3_all", "bf16x3_2d1", "bf16x3_f32act") else "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: 7 cameras 3x640x960, VoV-99, A=%d queries (644 learned + 644 adaptive + 256 "
                                    "propagated), 2312 self-attn keys, 6 decoder layers, streaming memory on, ego motion on" % A) if args.proposals == "topk" else
                                   ("BASELINE configs[1] with the reference's threshold proposal rule: 7 cameras 3x640x960, VoV-99, %d query rows (644 "
                                    "learned + %d adaptive rows of which %d hold a proposal in the last frame + 256 propagated), fixed capacity, "
                                    "count on the device, overflow=%s" % (A, args.capacity, res["n_adaptive"], res["proposal_overflow"])),
                       "proposals": args.proposals, "fused_rows": bool(args.fused_rows),
                       "parallelism": "single GPU" if world == 1 else ("camera-sharded x%d + 1 all-gather" % world if sharded else
                                                                       "%d independent scene streams (replicas, no collective)" % world),
                       "precision_assignment": {k: str(v).replace("torch.", "") for k, v in res["prec"].items()},
                       "weights": "seeded random (far3d_amd.weights.init_state_dict, seed 0)", "commit": build_commit()},
            "protocol": rates(res),
            "roofline": agg_roofline(args, res, traffic, traffic_src, live, committed, committed_src),
            "roofline_backbone": backbone_roofline(res),
        }
        if res["timing_error"]:
            line["kernel_timing_error"] = res["timing_error"]
        if res.get("lat_groups"):
            line["protocol"]["sync_per_frame_groups"] = res["lat_groups"]
        if res.get("two_streams"):
            line["protocol"]["two_streams_pipelined"] = res["two_streams"]
        if res.get("stage_ms"):
            # per rank: where a (sync-per-frame) frame goes -- per-camera stages, the exchange, the replicated head
            line["per_rank_stage_ms"] = res["stage_ms"]

        def finite(o):           # strict JSON: no NaN / Infinity (a missing kernel timing becomes null)
            if isinstance(o, dict):
                return {k: finite(v) for k, v in o.items()}
            if isinstance(o, list):
                return [finite(v) for v in o]
            if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
                return None
            return o
        orc = None
        if world == 1 and not args.no_cpu_baseline and args.proposals == "topk":
            line["cpu_baseline"], orc = oracle_frames(2)
            line["parity"] = parity_block(args.precision, res["eng_frames"], orc)
            line["meets_tolerance"] = line["parity"]["meets_tolerance"]
        if res_fast is not None:
            r = rates(res_fast)
            blk = {"dtype": "bf16", "what": "BASELINE configs[1]'s dtype: bf16 activations / weights / value maps / decoder GEMM operands, ONE bf16 "
                                            "MFMA per product.  ~40x outside the north-star logit tolerance (parity below): reported beside the "
                                            "headline, never as `value`",
                   "value": r["sync_per_frame"]["samples_per_s_mean"], "unit": "samples/s", "ms_per_step": r["sync_per_frame"]["mean_ms"],
                   "value_protocol": "sync_per_frame", "steps": res_fast["steps"], "protocol": r,
                   "precision_assignment": {k: str(v).replace("torch.", "") for k, v in res_fast["prec"].items()},
                   "roofline": agg_roofline(args, res_fast, *agg_traffic(fp32_rows=False)), "roofline_backbone": backbone_roofline(res_fast)}
            if res_fast.get("lat_groups"):
                blk["protocol"]["sync_per_frame_groups"] = res_fast["lat_groups"]
            if orc is not None:
                blk["parity"] = parity_block("bf16", res_fast["eng_frames"], orc)
                blk["meets_tolerance"] = blk["parity"]["meets_tolerance"]
            line["fast_mode"] = blk
        print(json.dumps(finite(line)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
