"""Camera-sharded multi-GPU execution of ONE sample: one process per GPU, each rank runs the
per-camera stages (backbone, FPN+MLN, 2D head, proposals) for its cameras, ONE exchange (RCCL over xGMI; `nccl`
backend on ROCm) collects the token-major value maps and the adaptive-query records, and every rank runs the identical
cross-camera decoder (deterministic kernels -> replicated streaming memory, no further exchange).

The reference has no counterpart (it only replicates whole samples, tools/test.py:229-234); the decoder cannot be
sharded by camera because its softmax spans cameras (models/utils/detr3d_transformer.py:540).

What a frame costs on a rank (steady state, `use_graph`): one hipGraph for the per-camera stages, the exchange (the value
maps, the records and -- fixed-capacity threshold mode -- the per-rank proposal counts travel as ONE coalesced RCCL group,
issued eagerly: no collective is ever captured), one hipGraph for the replicated head.  With `pipeline` the two halves run on
two streams with two buffer sets, exactly like Far3DEngine.pipeline on one GPU: the camera graph and the exchange of frame i+1
overlap the head of frame i, so the frame time tends to max(camera stages + exchange, head) instead of their sum.

Host logic here is device-agnostic so the shard / gather / un-pad arithmetic is covered by 2-process gloo tests on CPU;
the compute itself has no CPU path.
"""
import torch
import torch.distributed as dist


def camera_shards(num_cams, world):
    """Contiguous camera blocks, padded with -1 so every rank owns `per` slots (7 cameras -> 8 slots on 8 GPUs)."""
    per = -(-num_cams // world)
    return per, [[c if c < num_cams else -1 for c in range(r * per, (r + 1) * per)] for r in range(world)]


def gather_camera_major(local, num_cams, group=None, async_op=False, out=None):
    """All-gather equally-shaped per-rank blocks whose leading dim is `per` camera slots and drop the padding slots.

    local: (per, ...) tensor (padding slots may hold anything).  out: optional preallocated (world*per, ...) destination (a
    persistent buffer that a captured head graph reads in place).  Returns (work_or_None, fn) where fn() -> (num_cams, ...)
    view of the gathered buffer; call work.wait() before fn() when async_op is set."""
    world = dist.get_world_size(group)
    shape = (world * local.shape[0],) + tuple(local.shape[1:])
    if out is None:
        out = torch.empty(shape, dtype=local.dtype, device=local.device)
    assert tuple(out.shape) == shape and out.dtype == local.dtype and out.is_contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # test rig only (several ranks sharing one GPU, tests/test_dist_gpu.py): gloo has no device collectives, stage via host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        out.copy_(host)
        return _Done(), (lambda: out[:num_cams])
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    return work, (lambda: out[:num_cams])


def gather_many(pairs, group=None):
    """All-gather several (local block, destination) pairs as ONE exchange.  On the RCCL backend the calls are coalesced into a
    single group (one launch moves the value maps, the records and the counts); elsewhere (gloo: CPU tests, the one-GPU test rig)
    they run one after the other.  Returns a list of work handles to wait() on (stream-ordered on the device backends)."""
    backend = dist.get_backend(group)
    if backend == "nccl" and len(pairs) > 1 and _COALESCE is not None and _coalesce_ok(group, pairs[0][0].device):
        with _COALESCE(group=group, device=pairs[0][0].device, async_ops=True) as c:
            for local, out in pairs:
                dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return [c]
    works = []
    for local, out in pairs:
        if local.is_cuda and backend == "gloo":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
            out.copy_(host)
        else:
            works.append(dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=True))
    return works


def _coalescing_manager():
    """torch.distributed's group-call manager if this build has it with the (group, device, async_ops) signature, else None.
    Decided once, by inspection: an exchange must never be retried in another form after part of it was issued."""
    import inspect
    cm = getattr(dist, "_coalescing_manager", None)
    if cm is None:
        return None
    try:
        params = inspect.signature(cm).parameters
    except (TypeError, ValueError):
        return None
    return cm if all(k in params for k in ("group", "device", "async_ops")) else None


_COALESCE = _coalescing_manager()
_COALESCE_CHECKED = {}


def _coalesce_ok(group, device):
    """One-time self-check of the coalesced exchange on this process group (it rides on a private torch API whose
    mixed-dtype fast path has only ever run here with a world of one).  Every rank gathers a small known pattern of the three
    dtypes the frame exchange mixes (bf16, f32, int32) through the coalescing manager and compares; a mismatch or an exception
    switches the group to sequential async gathers for good.  Collective: the first gather_many of a group runs it on every rank."""
    key = id(group)
    if key in _COALESCE_CHECKED:
        return _COALESCE_CHECKED[key]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ok = True
    try:
        srcs = [torch.full((5,), rank + 1, dtype=torch.bfloat16, device=device), torch.full((3,), 0.5 * (rank + 1), dtype=torch.float32, device=device),
                torch.full((1, 2), 7 * (rank + 1), dtype=torch.int32, device=device)]
        dsts = [torch.zeros((world,) + tuple(t.shape), dtype=t.dtype, device=device) for t in srcs]
        with _COALESCE(group=group, device=device, async_ops=True) as c:
            for t, d in zip(srcs, dsts):
                dist.all_gather_into_tensor(d.view(-1, *t.shape[1:]) if t.dim() > 1 else d.view(-1), t, group=group)
        c.wait()
        torch.cuda.synchronize(device)
        for t, d in zip(srcs, dsts):
            want = torch.stack([torch.full_like(t, float(v)) for v in ((r + 1) * (1 if t.dtype == torch.bfloat16 else 0.5 if t.dtype == torch.float32 else 7)
                                                                         for r in range(world))])
            ok = ok and bool(torch.equal(d, want))
    except Exception:   # noqa: BLE001  (any failure of the private API: use the public one)
        ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)     # all ranks take the same path
    _COALESCE_CHECKED[key] = bool(flag.item())
    return _COALESCE_CHECKED[key]


class _Done:
    def wait(self):
        return True


class QueryShard:
    """Row partition + per-layer exchange of the query-sharded decoder (Far3DEngine.decoder_query_sharded).  gather(src, dst) is a
    hook: ShardedFrame replaces it while it captures the head so that every exchange ends one hipGraph segment and starts the
    next (no collective is ever captured)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.gather = self.exchange

    def rows_per_rank(self, A):
        return -(-(-(-A // self.world)) // 4) * 4          # ceil(A / world), rounded up to a multiple of 4 rows

    def exchange(self, src, dst):
        for w in gather_many([(src, dst)], self.group):
            w.wait()


class _Replay:
    def __init__(self, g, qs=None):
        self.g, self.qs = g, qs

    def replay(self):
        if self.qs is None:
            self.g.replay()
        else:
            self.g.replay(self.qs)


class _SegmentedGraph:
    """A head captured as hipGraph segments separated by eager exchanges: replay() = g0, exchange, g1, exchange, ... gN."""

    def __init__(self):
        self.items = []

    def replay(self, qs):
        for it in self.items:
            if it[0] == "graph":
                it[1].replay()
            else:
                qs.exchange(it[1], it[2])


class ShardedFrame:
    """Drives a Far3DEngine in camera-sharded mode.  Needs static shapes: the top-K proposal mode (cfg['proposal_topk'] = K) or the
    fixed-capacity threshold mode (cfg['proposal_capacity']; each rank compacts its cameras' proposals into a block, the blocks
    and their counts are gathered and packed into the reference's camera-major order on every rank by far3d_compact_rows).

    use_graph: replay the steady-state frame as TWO hipGraphs per rank -- the per-camera stages and the replicated head --
    with the exchange issued eagerly between them (no collective is ever captured).  The graphs survive scene changes:
    the first frame of a scene runs eagerly and resets the engine's streaming memory in place.
    pipeline (with use_graph): frames of one stream are software-pipelined like on one GPU (engine.pipeline_sets buffer sets,
    engine.cam_streams high-priority camera streams): the camera graph + exchange of the next frames run side by side -- a rank
    with one or two cameras is launch-bound, three frames at once cost 1.08 ms per frame where one costs 2.33 (1 camera;
    profiles/r4/stage_times_bf16.txt) -- under the head of an earlier frame.  Exchanges are issued in frame order on every rank.
    Results are bit-identical to the unpipelined runner; outputs are ready on `output_stream()` (call `wait_outputs()` before
    reading them on another stream)."""

    def __init__(self, engine, group=None, use_graph=False, pipeline=False, decoder="replicated"):
        if engine.static_adaptive_rows() is None:
            raise ValueError("camera sharding needs static shapes: proposal_topk=K, or the fixed-capacity threshold mode "
                             "(proposal_capacity=rows); the legacy threshold mode syncs on a data-dependent M")
        self.eng, self.group = engine, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_cams = engine.cfg["num_cams"]
        self.per, shards = camera_shards(self.num_cams, self.world)
        self.slots = shards[self.rank]
        self.cams = [c for c in self.slots if c >= 0]
        self.use_graph = use_graph
        self.pipeline = bool(pipeline and use_graph)
        if self.pipeline and engine.tile_table is None:
            engine.tile_table = "tuning_mi355x_tput.json"      # tiles picked under the pipeline's concurrency (Far3DEngine.tile_table)
        # decoder="query_sharded": every rank runs A / world of the queries through the six layers (one small all-gather per layer)
        # instead of the whole replicated decoder; the graphs of the head are then captured as segments between the exchanges
        if decoder not in ("replicated", "query_sharded"):
            raise ValueError("decoder must be 'replicated' or 'query_sharded'")
        self.qshard = QueryShard(group) if (decoder == "query_sharded" and self.world > 1) else None
        self.capacity = engine.cfg.get("proposal_capacity") if engine.cfg["proposal_topk"] is None else None
        # fixed-capacity mode: rows of one rank's record block (its cameras cannot hold more proposals than their selection capacity)
        self.block_rows = min(self.capacity, self.per * engine.cfg["proposal_cap"]) if self.capacity is not None else None
        self._g_cam, self._g_head, self._st, self._head_out, self._sig = {}, {}, {}, {}, {}
        self._bufs = {}
        self._scene = None
        self._fidx = 0
        self._pipe = None
        # record_stage_times (unpipelined frames only; bench.py --gpus N): device events around the three parts of a rank's frame --
        # per-camera stages, exchange, replicated head -- so that a scaling line can be read against DESIGN.md section 7's table
        self.record_stage_times = False
        self.stage_times = []        # per frame: [event start, after the camera stages, after the exchange, after the head]

    def mean_stage_times(self):
        """Mean ms per part over the recorded frames (synchronises)."""
        if not self.stage_times:
            return dict(frames=0, camera_stage_ms=None, exchange_ms=None, head_ms=None)
        torch.cuda.synchronize(self.eng.dev)
        n = len(self.stage_times)
        cam = sum(e[0].elapsed_time(e[1]) for e in self.stage_times) / n
        exc = sum(e[1].elapsed_time(e[2]) for e in self.stage_times) / n
        head = sum(e[2].elapsed_time(e[3]) for e in self.stage_times) / n
        return dict(frames=n, camera_stage_ms=cam, exchange_ms=exc, head_ms=head)

    def _mark(self, evs):
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.eng.dev))
            evs.append(e)

    # ------------------------------------------------------------------------------------------ buffers
    def _gather_bufs(self, p, tok, rec):
        """Persistent exchange destinations of buffer set p (allocated once): the replicated head -- eager or captured -- reads the
        value maps and the adaptive-query records straight out of them, so a frame moves them exactly once."""
        b = self._bufs.get(p)
        if b is None or b["tok"].shape[1:] != tok.shape[1:] or b["tok"].dtype != tok.dtype or b["rec"].shape[1:] != rec.shape[1:]:
            dev = tok.device
            b = dict(tok=torch.empty((self.world * self.per,) + tuple(tok.shape[1:]), dtype=tok.dtype, device=dev),
                     rec=torch.empty((self.world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=dev))
            if self.capacity is not None:
                E = self.eng.cfg["embed_dims"]
                b.update(cnt=torch.zeros((self.world, 2), dtype=torch.int32, device=dev),       # per rank: (proposals, overflow flag)
                         rows=torch.empty((self.capacity, E + 4), dtype=torch.float32, device=dev),
                         m=torch.zeros((1,), dtype=torch.int32, device=dev), ovf=torch.zeros((1,), dtype=torch.int32, device=dev))
            self._bufs[p] = b
            self._g_head.pop(p, None)
        return b

    def _pad(self, t):
        if t.shape[0] < self.per:
            t = torch.cat([t, torch.zeros((self.per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)])
        return t

    # ------------------------------------------------------------------------------------------ the two halves of a frame
    def _camera_part(self, dd, pad_hw):
        """Per-camera stages for this rank's cameras -> dict(tok (per,S,E) padded, rec, cnt | None, hw, starts).
        Top-K mode: rec (per,K,E+4), one row [context (E+1) | normalised reference point (3)] per proposal (context first: 16-byte
        aligned GEMM operand).  Fixed-capacity mode: rec (block_rows, E+4) = this rank's proposals compacted in camera order and
        zero-padded, cnt (1,2) int32 = (their number, overflow flag)."""
        eng, cfg, dev = self.eng, self.eng.cfg, self.eng.dev
        K = cfg["proposal_topk"]
        E = cfg["embed_dims"]
        cnt = None
        if self.cams:
            img = dd["img"][self.cams[0]:self.cams[-1] + 1]      # contiguous camera block: a view (no index tensor, capture-safe)
            st = eng.camera_stage(img, dd, self.cams, pad_hw, block_rows=self.block_rows)
            tok = st["tokens"]
            if self.capacity is None:
                rec = self._pad(torch.cat([st["ctx"], st["ref2d"]], dim=1).view(len(self.cams), K, E + 4))
            else:
                rec = torch.cat([st["ctx"], st["ref2d"]], dim=1)
                cnt = torch.cat([st["m_dev"], st["overflow"]]).view(1, 2)
            hw, starts = st["hw"], st["starts"]
        else:           # idle rank (8 GPUs, 7 cameras): contributes padding only
            from .synth import level_shapes, level_starts
            hw = level_shapes(pad_hw, cfg["strides"])
            starts, S = level_starts(hw)
            tok = torch.zeros((0, S, E), dtype=eng.prec["value"], device=dev)
            if self.capacity is None:
                rec = torch.zeros((self.per, K, E + 4), dtype=torch.float32, device=dev)
            else:
                rec = torch.zeros((self.block_rows, E + 4), dtype=torch.float32, device=dev)
                cnt = torch.zeros((1, 2), dtype=torch.int32, device=dev)
        return dict(tok=self._pad(tok), rec=rec, cnt=cnt, hw=hw, starts=starts)

    def _exchange(self, p, st):
        """The ONE exchange of the frame: value maps + adaptive-query records (+ counts), never captured."""
        b = self._gather_bufs(p, st["tok"], st["rec"])
        pairs = [(st["tok"], b["tok"]), (st["rec"], b["rec"])]
        if self.capacity is not None:
            pairs.append((st["cnt"], b["cnt"]))
        return gather_many(pairs, self.group)

    def _head(self, p, dd, img_metas, hw, starts, pad_hw):
        from . import ops
        eng = self.eng
        K, E = eng.cfg["proposal_topk"], eng.cfg["embed_dims"]
        b = self._bufs[p]
        tok = b["tok"][:self.num_cams]
        if self.capacity is None:
            M = self.num_cams * K
            rec = b["rec"][:self.num_cams].view(M, E + 4)
            return eng.head_stage(tok, rec[:, E + 1:], rec[:, :E + 1], M, dd, img_metas, hw, starts, pad_hw, qshard=self.qshard)
        # fixed-capacity mode: pack the ranks' blocks into the reference's camera-major order (rank order = camera order)
        torch.amax(b["cnt"][:, 1:2], dim=0, out=b["ovf"])
        ops.compact_rows(b["rec"].view(self.world, self.block_rows, E + 4), b["cnt"][:, 0].contiguous(), b["rows"], b["m"], b["ovf"])
        eng._overflow = b["ovf"]
        out = eng.head_stage(tok, b["rows"][:, E + 1:], b["rows"][:, :E + 1], self.capacity, dd, img_metas, hw, starts, pad_hw, m_dev=b["m"],
                             qshard=self.qshard)
        out["proposal_overflow"] = b["ovf"]
        return out

    def _drop_stale_graphs(self, p, dd):
        """Graphs bake the addresses of the staged inputs: a re-allocation (new resolution / camera count) invalidates them."""
        sig = (dd["img"].data_ptr(), tuple(dd["img"].shape))
        if self._sig.get(p) != sig:
            if p in self._g_cam or p in self._g_head:
                torch.cuda.synchronize(self.eng.dev)
            for d in (self._g_cam, self._g_head, self._st, self._head_out):
                d.pop(p, None)
            self._sig[p] = sig

    def _capture(self, fn):
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of the process group keeps polling its events while we capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fn()
        return g, out

    def _capture_head(self, fn):
        """The head as ONE hipGraph (replicated decoder) or as segments between the per-layer exchanges (query-sharded decoder).
        Returns (object with .replay(), outputs)."""
        qs = self.qshard
        if qs is None:
            g, out = self._capture(fn)
            return _Replay(g), out
        seg = _SegmentedGraph()
        pool = torch.cuda.graph_pool_handle()
        state = {}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["ctx"] = torch.cuda.graph(state["g"], pool=pool, capture_error_mode="thread_local")
            state["ctx"].__enter__()

        def end():
            state["ctx"].__exit__(None, None, None)
            seg.items.append(("graph", state["g"]))

        def cut(src, dst):          # an exchange inside the head: close the segment, remember the exchange, open the next one
            end()
            seg.items.append(("gather", src, dst))
            begin()
        qs.gather = cut
        try:
            begin()
            try:
                out = fn()
            except BaseException as e:        # leave the stream-capture state clean before the error travels on
                state["ctx"].__exit__(type(e), e, e.__traceback__)
                raise
            end()
        finally:
            qs.gather = qs.exchange
        return _Replay(seg, qs), out

    # ------------------------------------------------------------------------------------------ frames
    def output_stream(self):
        """The stream the latest forward_frame's outputs are produced on."""
        return self._pipe["s_head"] if (self._pipe is not None and self.eng._ready is not None) else torch.cuda.current_stream(self.eng.dev)

    def wait_outputs(self):
        self.eng.wait_outputs()

    @torch.no_grad()
    def forward_frame(self, data, img_metas):
        eng = self.eng
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        scene = img_metas[0]["scene_token"]
        # an idle rank (no cameras) has no camera graph: it only pads the exchange and runs the head, and is never the slowest rank
        steady = self.use_graph and scene == self._scene and eng._mem_valid
        self._scene = scene
        p = eng._par = (self._fidx % max(2, int(eng.pipeline_sets))) if self.pipeline else 0
        self._fidx += 1
        if self.pipeline and steady:
            return self._pipelined_frame(data, img_metas, pad_hw)
        if self._pipe is not None:      # scene start in pipeline mode: eager on the caller's stream, after everything in flight
            cur = torch.cuda.current_stream(eng.dev)
            for sc in self._pipe["s_cams"]:
                cur.wait_stream(sc)
            cur.wait_stream(self._pipe["s_head"])
        eng._ready = None
        dd = eng._stage_inputs(data)      # every rank keeps the (small) calibration inputs; images are sliced per rank
        self._drop_stale_graphs(p, dd)
        evs = [] if (self.record_stage_times and steady and p in self._g_head) else None      # replayed frames only (a capture is not a frame)
        self._mark(evs)
        # ---- per-camera stages: eager, or one hipGraph per rank in steady state
        if steady and self.cams:
            if p not in self._g_cam:
                self._g_cam[p], self._st[p] = self._capture(lambda: self._camera_part(dd, pad_hw))
            self._g_cam[p].replay()
            st = self._st[p]
        else:
            st = self._camera_part(dd, pad_hw)
        self._mark(evs)
        for w in self._exchange(p, st):
            w.wait()
        self._mark(evs)
        # ---- replicated head on the gathered buffers: eager on the first frame of a scene (memory reset), else a hipGraph
        if not steady:
            return self._head(p, dd, img_metas, st["hw"], st["starts"], pad_hw)
        if p not in self._g_head:
            self._g_head[p], self._head_out[p] = self._capture_head(lambda: self._head(p, dd, img_metas, st["hw"], st["starts"], pad_hw))
        self._g_head[p].replay()
        self._mark(evs)
        if evs is not None:
            self.stage_times.append(evs)
        eng._overflow = self._head_out[p].get("proposal_overflow")
        return self._head_out[p]

    def _pipelined_frame(self, data, img_metas, pad_hw):
        """One steady-state frame in pipeline mode.  Stream s_cam: [wait until the head that last used this buffer set is done] ->
        input staging -> camera graph -> exchange.  Stream s_head: [wait for the exchange] -> head graph."""
        eng = self.eng
        if self._pipe is None:
            ncs = max(1, min(int(eng.pipeline_sets) - 1, int(eng.cam_streams)))
            self._pipe = dict(s_cams=[torch.cuda.Stream(eng.dev, priority=eng.cam_priority) for _ in range(ncs)], s_head=torch.cuda.Stream(eng.dev),
                              cam_done={}, head_done={}, n_issued=0)
        P = self._pipe
        p = eng._par
        cur = torch.cuda.current_stream(eng.dev)
        here = torch.cuda.Event()
        here.record(cur)
        img = data["img"][0] if data["img"].dim() == 5 else data["img"]
        cached = eng._ins.get(p)
        if cached is None or tuple(cached["img"].shape) != tuple(img.shape) or p not in self._g_head:
            # first steady frame on this buffer set (or a new input shape): capture its graphs with the device quiet.  The eager
            # exchange inside is a collective: every rank takes this branch on the same frame (same frame sequence on every rank)
            torch.cuda.synchronize(eng.dev)
            dd = eng._stage_inputs(data)
            self._drop_stale_graphs(p, dd)
            if (p, "tq") not in eng._bufs:
                eng._alloc_query_buffers(self.num_cams)
            if self.cams:
                self._g_cam[p], self._st[p] = self._capture(lambda: self._camera_part(dd, pad_hw))
                self._g_cam[p].replay()
            else:
                self._st[p] = self._camera_part(dd, pad_hw)
            st = self._st[p]
            for w in self._exchange(p, st):
                w.wait()
            self._g_head[p], self._head_out[p] = self._capture_head(lambda: self._head(p, dd, img_metas, st["hw"], st["starts"], pad_hw))
            P["cam_done"][p], P["head_done"][p] = torch.cuda.Event(), torch.cuda.Event()
            torch.cuda.synchronize(eng.dev)
            first = True
        else:
            first = False
        st = self._st[p]
        s_cam = P["s_cams"][P["n_issued"] % len(P["s_cams"])]      # consecutive frames alternate between the camera streams
        P["n_issued"] += 1
        s_cam.wait_event(here)
        for v in data.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(s_cam)
        with torch.cuda.stream(s_cam):
            if not first:
                s_cam.wait_event(P["head_done"][p])      # the head that last read this buffer set (inputs and exchange buffers included)
            eng._stage_inputs(data)
            if self.cams:
                self._g_cam[p].replay()
            works = self._exchange(p, st)
            P["cam_done"][p].record(s_cam)
        with torch.cuda.stream(P["s_head"]):
            P["s_head"].wait_event(here)
            P["s_head"].wait_event(P["cam_done"][p])
            for w in works:
                w.wait()                                         # stream-ordered on the RCCL backend: s_head waits, the host does not
            self._g_head[p].replay()
            P["head_done"][p].record(P["s_head"])
        eng._ready = P["head_done"][p]
        eng._overflow = self._head_out[p].get("proposal_overflow")      # this frame's flag (its buffer set), not the last captured one
        return self._head_out[p]
