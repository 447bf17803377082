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


class _Done:
    def wait(self):
        return True


class ShardedFrame:
    """Drives a Far3DEngine in camera-sharded mode.  Static-M proposal mode only (cfg['proposal_topk'] = K).

    use_graph: replay the steady-state frame as TWO hipGraphs per rank -- the per-camera stages and the replicated head --
    with the collectives issued eagerly between them (no collective is ever captured).  The graphs survive scene changes:
    the first frame of a scene runs eagerly and resets the engine's streaming memory in place."""

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
        self._st = self._head_out = None
        self._tok_all = self._rec_all = None
        self._scene = None

    def _gather_bufs(self, tok, rec):
        """Persistent all-gather destinations (allocated once): the replicated head -- eager or captured -- reads the value
        maps and the adaptive-query records straight out of them, so a frame moves them exactly once."""
        if self._tok_all is None or self._tok_all.shape[1:] != tok.shape[1:] or self._tok_all.dtype != tok.dtype:
            self._tok_all = torch.empty((self.world * self.per,) + tuple(tok.shape[1:]), dtype=tok.dtype, device=tok.device)
            self._rec_all = torch.empty((self.world * self.per,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
            self._g_head = None
        return self._tok_all, self._rec_all

    def _pad(self, t):
        if t.shape[0] < self.per:
            t = torch.cat([t, torch.zeros((self.per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)])
        return t

    def _camera_part(self, dd, pad_hw):
        """Per-camera stages for this rank's cameras -> (padded tokens (per,S,E), padded records (per,K,E+1+3), hw, starts).
        A record row is [context (E+1) | normalised reference point (3)] (context first: 16-byte aligned GEMM operand)."""
        eng, cfg, dev = self.eng, self.eng.cfg, self.eng.dev
        K = cfg["proposal_topk"]
        E = cfg["embed_dims"]
        if self.cams:
            img = dd["img"][self.cams[0]:self.cams[-1] + 1]      # contiguous camera block: a view (no index tensor, capture-safe)
            st = eng.camera_stage(img, dd, self.cams, pad_hw)
            tok = st["tokens"]
            rec = torch.cat([st["ctx"], st["ref2d"]], dim=1).view(len(self.cams), K, E + 4)
            hw, starts = st["hw"], st["starts"]
        else:           # idle rank (8 GPUs, 7 cameras): contributes padding only
            from .synth import level_shapes, level_starts
            hw = level_shapes(pad_hw, cfg["strides"])
            starts, S = level_starts(hw)
            tok = torch.zeros((0, S, E), dtype=eng.prec["value"], device=dev)
            rec = torch.zeros((0, K, E + 4), dtype=torch.float32, device=dev)
        return self._pad(tok), self._pad(rec), hw, starts

    def _head(self, dd, img_metas, hw, starts, pad_hw):
        K, E = self.eng.cfg["proposal_topk"], self.eng.cfg["embed_dims"]
        M = self.num_cams * K
        rec = self._rec_all[:self.num_cams].view(M, E + 4)
        return self.eng.head_stage(self._tok_all[:self.num_cams], rec[:, E + 1:], rec[:, :E + 1], M, dd, img_metas, hw, starts, pad_hw)

    @torch.no_grad()
    def forward_frame(self, data, img_metas):
        eng = self.eng
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        dd = eng._stage_inputs(data)      # every rank keeps the (small) calibration inputs; images are sliced per rank
        scene = img_metas[0]["scene_token"]
        # an idle rank (no cameras) stays eager: it only pads the gathers and runs the head, and is never the slowest rank
        steady = self.use_graph and scene == self._scene and eng._mem_valid
        self._scene = scene
        # ---- per-camera stages: eager, or one hipGraph per rank in steady state
        if steady and self.cams:
            if self._g_cam is None:
                g = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread of the process group keeps polling its events while we capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._st = self._camera_part(dd, pad_hw)
                self._g_cam = g
            self._g_cam.replay()
            tok, rec, hw, starts = self._st
        else:
            tok, rec, hw, starts = self._camera_part(dd, pad_hw)
        # ---- the ONE exchange of the frame: value maps + adaptive-query records, never captured
        tok_all, rec_all = self._gather_bufs(tok, rec)
        work, _ = gather_camera_major(tok, self.num_cams, self.group, async_op=True, out=tok_all)
        gather_camera_major(rec, self.num_cams, self.group, out=rec_all)
        if work is not None:
            work.wait()
        # ---- replicated head on the gathered buffers: eager on the first frame of a scene (memory reset), else a hipGraph
        if not steady:
            return self._head(dd, img_metas, hw, starts, pad_hw)
        if self._g_head is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._head_out = self._head(dd, img_metas, hw, starts, pad_hw)
            self._g_head = g
        self._g_head.replay()
        return self._head_out
