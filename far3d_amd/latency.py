"""Single-GPU LATENCY mode for one frame: the per-camera stages as G camera groups side by side on G streams.

Why: a frame's per-camera stages are ~215 dependent launches, many of them one or two rounds of workgroups (VoVNet stages 4-5, FPN,
2D head); one frame alone leaves the chip half empty.  Far3DEngine.pipeline fills it with the stages of OTHER frames (throughput:
bench.py's `value`); this runner fills it with the other half of the SAME frame's cameras, which is what shortens a single frame --
the reference's own benchmark protocol syncs around every frame (tools/analysis_tools/benchmark.py:84-111; bench.py
`protocol.sync_per_frame`).  Measured basis (profiles/r4/stage_times_bf16.txt): 7 cameras in one batch 6.09 ms; two independent
4-camera jobs side by side 3.15 ms per job, i.e. 0.79 ms per camera against 0.87.

Mechanics (the engine's frame pipeline turned sideways): every camera group has its own buffer namespace (`engine._par = ("camgroup",
g)`) and, in steady state, its own hipGraph replayed on its own high-priority stream; the groups write their value maps and
adaptive-query records into ONE pair of persistent head-input buffers (camera-major, exactly what camera_stage of all cameras would
have produced), and the head -- eager on the first frame of a scene, a hipGraph afterwards -- runs on the caller's stream after all
groups.  Same kernels on the same per-camera data as the plain engine; results equal up to the tile choice of layers whose tile table
entry depends on the pixel count (fp32: identical accumulation order; tests/test_latency_gpu.py).  Static top-K proposal mode only.

Opt-in (`bench.py --latency-groups 2` reports it as protocol.sync_per_frame_groups beside the engine's own figure).  Measured in round 5
(profiles/r5): 6.90 ms per frame against 7.08 ms for the plain engine on the same box -- less than the 0.6 ms the stage times promised:
two camera groups side by side slow each other down more than two independent jobs did.
"""
import torch


class CameraGroupFrame:
    def __init__(self, engine, groups=2, use_graph=True, priority=-1):
        cfg = engine.cfg
        if cfg["proposal_topk"] is None:
            raise ValueError("CameraGroupFrame needs the static top-K proposal mode (cfg['proposal_topk'] = K)")
        if engine.pipeline:
            raise ValueError("CameraGroupFrame is the single-frame latency runner: use it on an engine without frame pipelining")
        self.eng = engine
        N = cfg["num_cams"]
        groups = max(1, min(int(groups), N))
        per = -(-N // groups)
        self.blocks = [(lo, min(lo + per, N)) for lo in range(0, N, per)]          # contiguous camera blocks, e.g. 7 -> (0,4), (4,7)
        self.use_graph = bool(use_graph)
        self.streams = [torch.cuda.Stream(engine.dev, priority=priority) for _ in self.blocks]
        self._b = None
        self._g_cam, self._g_head, self._head_out, self._meta = {}, None, None, None
        self._sig = None
        self._scene = None

    # ------------------------------------------------------------------------------------------ the halves of a frame
    def _head_inputs(self, tok, K):
        """Persistent head inputs (allocated once): value maps (N,S,E) and records (N,K,E+4) = [context (E+1) | reference point (3)]."""
        eng = self.eng
        N, E = eng.cfg["num_cams"], eng.cfg["embed_dims"]
        b = self._b
        if b is None or b["tok"].shape[1:] != tok.shape[1:] or b["tok"].dtype != tok.dtype:
            b = self._b = dict(tok=torch.empty((N,) + tuple(tok.shape[1:]), dtype=tok.dtype, device=tok.device),
                               rec=torch.empty((N, K, E + 4), dtype=torch.float32, device=tok.device))
            self._g_head = None
        return b

    def _group_part(self, g, dd, pad_hw):
        """Per-camera stages of camera block g on the CURRENT stream, results stored into the shared head inputs."""
        eng = self.eng
        K, E = eng.cfg["proposal_topk"], eng.cfg["embed_dims"]
        lo, hi = self.blocks[g]
        keep = eng._par
        eng._par = ("camgroup", g)
        try:
            st = eng.camera_stage(dd["img"][lo:hi], dd, range(lo, hi), pad_hw)
        finally:
            eng._par = keep
        b = self._head_inputs(st["tokens"], K)
        b["tok"][lo:hi].copy_(st["tokens"])
        b["rec"][lo:hi, :, :E + 1].copy_(st["ctx"].view(hi - lo, K, E + 1))
        b["rec"][lo:hi, :, E + 1:].copy_(st["ref2d"].view(hi - lo, K, 3))
        return st["hw"], st["starts"]

    def _head(self, dd, img_metas, hw, starts, pad_hw):
        eng = self.eng
        K, E, N = eng.cfg["proposal_topk"], eng.cfg["embed_dims"], eng.cfg["num_cams"]
        rec = self._b["rec"].view(N * K, E + 4)
        return eng.head_stage(self._b["tok"], rec[:, E + 1:], rec[:, :E + 1], N * K, dd, img_metas, hw, starts, pad_hw)

    def _fork(self, fn):
        """fn(g) for every group on its own stream, ordered after the caller's stream; the caller's stream then waits for all."""
        cur = torch.cuda.current_stream(self.eng.dev)
        start = torch.cuda.Event()
        start.record(cur)
        outs = []
        for g, s in enumerate(self.streams):
            s.wait_event(start)
            with torch.cuda.stream(s):
                outs.append(fn(g))
                done = torch.cuda.Event()
                done.record(s)
            cur.wait_event(done)
        return outs

    # ------------------------------------------------------------------------------------------ frames
    @torch.no_grad()
    def forward_frame(self, data, img_metas):
        eng = self.eng
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        scene = img_metas[0]["scene_token"]
        # graphs only after an eager frame of this runner has allocated every group's buffers and the head inputs outside a capture
        steady = self.use_graph and scene == self._scene and eng._mem_valid and self._b is not None
        self._scene = scene
        eng._par = 0
        eng._ready = None
        dd = eng._stage_inputs(data)
        sig = (dd["img"].data_ptr(), tuple(dd["img"].shape))
        if sig != self._sig:                 # graphs bake the addresses of the staged inputs
            if self._g_cam or self._g_head is not None:
                torch.cuda.synchronize(eng.dev)
            self._g_cam, self._g_head, self._head_out = {}, None, None
            self._sig = sig
        if not steady:
            # first frame of a scene (streaming memory reset in place) or graphs off: eager, the groups ONE AFTER THE OTHER on the caller's
            # stream.  (Until round 5 the eager groups also ran side by side on their streams; the full GPU suite then caught that form
            # differing from the plain engine on one run out of two -- eager launches allocate their temporaries while they are enqueued,
            # and nothing else in this code base enqueues eager work on several streams at once.  Only the graph replays, whose buffers all
            # exist beforehand, run concurrently; an eager frame is a scene start, where latency is not the point.)
            for g in range(len(self.blocks)):
                self._meta = self._group_part(g, dd, pad_hw)
            return self._head(dd, img_metas, *self._meta, pad_hw)
        if len(self._g_cam) < len(self.blocks) or self._g_head is None:
            # capture with the device quiet: one graph per camera group (each on the capturing stream), then the head's
            torch.cuda.synchronize(eng.dev)
            for g in range(len(self.blocks)):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    self._meta = self._group_part(g, dd, pad_hw)
                self._g_cam[g] = gr
            gh = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gh):
                self._head_out = self._head(dd, img_metas, *self._meta, pad_hw)
            self._g_head = gh
            torch.cuda.synchronize(eng.dev)
        self._fork(lambda g: self._g_cam[g].replay())
        self._g_head.replay()
        return self._head_out

    def wait_outputs(self):
        """Outputs are produced on the caller's stream (nothing to wait for); kept for interface parity with the other runners."""
