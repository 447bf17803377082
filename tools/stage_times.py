"""Device time of the stages a camera-sharded frame is made of, on ONE GPU (hipGraph replays, no launch gaps): the per-camera stages
for 7 / 4 / 2 / 1 cameras (what a rank of 1 / 2 / 4 / 8 GPUs runs), the replicated head, and the six decoder layers inside it for all
queries and for 1/2, 1/4, 1/8 of them (the query-sharded decoder's per-rank share).  DESIGN.md section 7 builds the expected 2 / 4 /
8-GPU frame times from these figures (the multi-GPU node is not available to this round's runs).  usage: stage_times.py [precision]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import engine, synth, weights  # noqa: E402

K_PROP = 92


def timeit(fn, iters=5, reps=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps)


class _FakeShard:
    """QueryShard stand-in: rank 0 of `world`, the exchange replaced by a device copy of the same size (its cost on xGMI is modelled
    separately in DESIGN.md)."""

    def __init__(self, world):
        self.rank, self.world = 0, world

    def rows_per_rank(self, A):
        return -(-(-(-A // self.world)) // 4) * 4

    def gather(self, src, dst):
        dst[:src.shape[0]].copy_(src)


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dev = torch.device("cuda", 0)
    cfg = engine.default_cfg(proposal_topk=K_PROP)
    sd = weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0)
    eng = engine.Far3DEngine(sd, cfg, device=dev, precision=precision)
    frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=dev, ego_motion=True) for fi in range(2)]
    for f in frames:                                     # frame 1 = a steady frame (memory in use)
        out = eng.forward_frame(*f)
    data, metas = frames[1]
    pad_hw = tuple(metas[0]["pad_shape"][0][:2])
    dd = eng._stage_inputs(data)
    res = {"precision": precision}
    for n in (7, 4, 2, 1):
        res["camera_stages_%dcam_ms" % n] = timeit(lambda: eng.camera_stage(dd["img"][:n], dd, range(n), pad_hw), iters=2)
    # the same stages with the frame pipeline's concurrency: K independent frames (one buffer set and one hipGraph each) replayed
    # side by side on K high-priority streams -- what a rank does in steady state; ms PER FRAME
    import time
    for n in (7, 4, 2, 1):
        graphs = []
        for j in range(3):
            eng._par = j
            eng.camera_stage(dd["img"][:n], dd, range(n), pad_hw)            # allocate this set's buffers outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.camera_stage(dd["img"][:n], dd, range(n), pad_hw)
            graphs.append(g)
        eng._par = 0
        streams = [torch.cuda.Stream(dev, priority=-1) for _ in range(3)]
        for k in (2, 3):
            def go():
                for j in range(k):
                    with torch.cuda.stream(streams[j]):
                        graphs[j].replay()
            go(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                go()
            torch.cuda.synchronize()
            res["camera_stages_%dcam_%dstreams_ms_per_frame" % (n, k)] = (time.perf_counter() - t0) / (10 * k) * 1e3
        del graphs
    st = eng._camera_part(dd, pad_hw)
    res["head_replicated_ms"] = timeit(lambda: eng._head_part(st, dd, metas, pad_hw), iters=2)
    # the decoder alone, on the live buffers of the head (same operands every replay: timing only)
    cfgE, nq, Lm, P_ = cfg["embed_dims"], cfg["num_query"], cfg["memory_len"], cfg["num_propagated"]
    M = 7 * K_PROP
    A = nq + M + P_
    TQ, QP, RF = (eng._bufs[(0, k)] for k in ("tq", "qp", "rf"))
    X2 = eng._bufs[(0, "x2op")]
    args = (X2, TQ[:A], QP[:A], st["tokens"], RF[:A], st["hw"], st["starts"], dd["lidar2img"][0], pad_hw, A)
    x2_keep = X2.clone()

    def dec(qs=None):
        X2.copy_(x2_keep)
        eng.decoder(*args, qshard=qs)
    res["decoder_all_queries_ms"] = timeit(dec, iters=2)
    for w in (2, 4, 8):
        qs = _FakeShard(w)
        res["decoder_query_share_1_of_%d_ms" % w] = timeit(lambda: dec(qs), iters=2)
    res["head_outside_decoder_ms"] = res["head_replicated_ms"] - res["decoder_all_queries_ms"]
    tok_bytes = st["tokens"][0].numel() * st["tokens"].element_size()
    res["exchange_bytes_per_camera"] = tok_bytes + K_PROP * (cfgE + 4) * 4
    res["decoder_exchange_bytes_per_layer"] = A * cfgE * 4
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stage_times_%s.json" % precision), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
