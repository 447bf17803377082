"""Camera-sharded multi-GPU execution of ONE sample: one process per GPU, each rank runs the
per-camera stages (backbone, FPN+MLN, 2D head, proposals) for its cameras, ONE all-gather (RCCL over xGMI; `nccl`
backend on ROCm) collects the token-major value maps and the adaptive-query records, and every rank runs the identical
cross-camera decoder (deterministic kernels -> replicated streaming memory, no further exchange).

The reference has no counterpart (it only replicates whole samples, tools/test.py:229-234); the decoder cannot be
sharded by camera because its softmax spans cameras (models/utils/detr3d_transformer.py:540).

Host logic here is device-agnostic so the shard / gather / un-pad arithmetic is covered by 2-process gloo tests on CPU;
the compute itself has no CPU path.
"""
import torch
import torch.distributed as dist


def camera_shards(num_cams, world):
    """Contiguous camera blocks, padded with -1 so every rank owns `per` slots (7 cameras -> 8 slots on 8 GPUs)."""
    per = -(-num_cams // world)
    return per, [[c if c < num_cams else -1 for c in range(r * per, (r + 1) * per)] for r in range(world)]


def gather_camera_major(local, num_cams, group=None, async_op=False):
    """All-gather equally-shaped per-rank blocks whose leading dim is `per` camera slots and drop the padding slots.

    local: (per, ...) tensor (padding slots may hold anything).  Returns (work_or_None, fn) where fn() -> (num_cams, ...)
    view of the gathered buffer; call work.wait() before fn() when async_op is set."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # test rig only (several ranks sharing one GPU, tests/test_dist_gpu.py): gloo has no device collectives, stage via host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        out.copy_(host)
        return _Done(), (lambda: out[:num_cams])
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    return work, (lambda: out[:num_cams])


class _Done:
    def wait(self):
        return True


class ShardedFrame:
    """Drives a Far3DEngine in camera-sharded mode.  Static-M proposal mode only (cfg['proposal_topk'] = K).

    use_graph: replay the steady-state frame as TWO hipGraphs per rank -- the per-camera stages and the replicated head --
    with the collectives issued eagerly between them (no collective is ever captured); the value-map gather then starts
    after the 2D head instead of overlapping it."""

    def __init__(self, engine, group=None, use_graph=False):
        if engine.cfg["proposal_topk"] is None:
            raise ValueError("camera sharding needs the static proposal mode (proposal_topk=K): a data-dependent M "
                             "would need a second, size-exchanging collective")
        self.eng, self.group = engine, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_cams = engine.cfg["num_cams"]
        self.per, shards = camera_shards(self.num_cams, self.world)
        self.slots = shards[self.rank]
        self.cams = [c for c in self.slots if c >= 0]
        self.use_graph = use_graph
        self._g_cam = self._g_head = None
        self._st = self._head_in = self._head_out = None
        self._scene = None

    def _camera_part(self, dd, pad_hw, overlap):
        """Per-camera stages for this rank's cameras -> (padded tokens (per,S,E), padded records (per,K,3+E+1), hw, starts).
        overlap: start the value-map gather from the after-FPN hook (eager mode) and return its handle instead of tokens."""
        eng, cfg, dev = self.eng, self.eng.cfg, self.eng.dev
        K = cfg["proposal_topk"]
        pending = {}

        def pad_tok(tok):
            if tok.shape[0] < self.per:
                tok = torch.cat([tok, torch.zeros((self.per - tok.shape[0],) + tuple(tok.shape[1:]), dtype=tok.dtype, device=dev)])
            return tok

        def start_gather(st):   # value maps are final right after the FPN: overlap their gather with the 2D head
            pending["tok"] = gather_camera_major(pad_tok(st["tokens"]), self.num_cams, self.group, async_op=True)

        if self.cams:
            img = dd["img"][self.cams[0]:self.cams[-1] + 1]      # contiguous camera block: a view (no index tensor, capture-safe)
            eng.after_fpn = start_gather if overlap else None
            try:
                st = eng.camera_stage(img, dd, self.cams, pad_hw)
            finally:
                eng.after_fpn = None
            tok = st["tokens"]
            rec = torch.cat([st["ref2d"], st["ctx"]], dim=1).view(len(self.cams), K, -1)
            hw, starts = st["hw"], st["starts"]
        else:           # idle rank (8 GPUs, 7 cameras): contributes padding only
            from .synth import level_shapes, level_starts
            hw = level_shapes(pad_hw, cfg["strides"])
            starts, S = level_starts(hw)
            tok = torch.zeros((0, S, cfg["embed_dims"]), dtype=eng.prec["value"], device=dev)
            if overlap:
                start_gather(dict(tokens=tok))
            rec = torch.zeros((0, K, 3 + cfg["embed_dims"] + 1), dtype=torch.float32, device=dev)
        if rec.shape[0] < self.per:
            rec = torch.cat([rec, torch.zeros((self.per - rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=dev)])
        return (pending.get("tok") if overlap else pad_tok(tok)), rec, hw, starts

    @torch.no_grad()
    def forward_frame(self, data, img_metas):
        eng = self.eng
        K = eng.cfg["proposal_topk"]
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        dd = eng._stage_inputs(data)      # every rank keeps the (small) calibration inputs; images are sliced per rank
        scene = img_metas[0]["scene_token"]
        # an idle rank (no cameras) stays eager: it only pads the gathers and runs the head, and is never the slowest rank
        steady = self.use_graph and bool(self.cams) and scene == self._scene and eng.mem is not None
        self._scene = scene
        if not steady:
            self._g_cam = self._g_head = None          # new scene: the captured branch decisions no longer hold
            pend, rec, hw, starts = self._camera_part(dd, pad_hw, overlap=True)
            _, rec_fn = gather_camera_major(rec, self.num_cams, self.group)
            work, tok_fn = pend
            if work is not None:
                work.wait()
            tokens = tok_fn()
            rec_all = rec_fn().reshape(self.num_cams * K, -1)
            ref2d, ctx = rec_all[:, :3].contiguous(), rec_all[:, 3:].contiguous()
            return eng.head_stage(tokens, ref2d, ctx, self.num_cams * K, dd, img_metas, hw, starts, pad_hw)
        # ---- steady state: graph(per-camera stages) -> eager gathers -> graph(replicated head)
        if self._g_cam is None:
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of the process group keeps polling its events while we capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._st = self._camera_part(dd, pad_hw, overlap=False)
            self._g_cam = g
        self._g_cam.replay()
        tok, rec, hw, starts = self._st
        _, tok_fn = gather_camera_major(tok, self.num_cams, self.group)
        _, rec_fn = gather_camera_major(rec, self.num_cams, self.group)
        rec_all = rec_fn().reshape(self.num_cams * K, -1)
        if self._head_in is None:
            self._head_in = (tok_fn().clone(), rec_all[:, :3].contiguous(), rec_all[:, 3:].contiguous())
        else:
            self._head_in[0].copy_(tok_fn())
            self._head_in[1].copy_(rec_all[:, :3])
            self._head_in[2].copy_(rec_all[:, 3:])
        if self._g_head is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._head_out = eng.head_stage(self._head_in[0], self._head_in[1], self._head_in[2], self.num_cams * K, dd, img_metas,
                                                hw, starts, pad_hw)
            self._g_head = g
        self._g_head.replay()
        return self._head_out
