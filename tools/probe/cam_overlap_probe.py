"""Do two camera-stage graphs of DIFFERENT frames overlap usefully on one MI355X?  The per-camera stages of a frame are ~215
dependent launches, many of them one or two rounds of workgroups (stages 4-5, FPN, 2D head): a second, independent frame on another
stream could fill their tails.  Replays the two camera graphs of the pipelined engine (one per buffer set) back to back on one
stream and concurrently on two, and prints the time per pair."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far3d_amd import engine, synth, weights  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    cfg = engine.default_cfg(proposal_topk=92)
    eng = engine.Far3DEngine(weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0), cfg, device=dev, precision=prec)
    eng.use_graph = eng.pipeline = True
    frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=dev, ego_motion=True) for fi in range(4)]
    for i in range(6):
        eng.forward_frame(*frames[i % 4])
    torch.cuda.synchronize()
    P = eng._pipe
    g0, g1 = P["g_cam"][0], P["g_cam"][1]
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def timed(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def serial():
        with torch.cuda.stream(sa):
            g0.replay(); g1.replay()

    def concurrent():
        with torch.cuda.stream(sa):
            g0.replay()
        with torch.cuda.stream(sb):
            g1.replay()
    ts, tc = timed(serial), timed(concurrent)
    print("%s: two camera-stage graphs, one stream: %.3f ms per pair (%.3f per frame); two streams: %.3f ms per pair (%.3f per frame): x%.3f"
          % (prec, ts, ts / 2, tc, tc / 2, ts / tc))


if __name__ == "__main__":
    main()
