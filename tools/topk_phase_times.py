"""Where does the single-workgroup top-K (far3d_topk / far3d_decode_topk: csrc/post.hip block_topk_sorted) spend its time?
Profiling build (libfar3d_hip_prof.so, -DFAR3D_PROFILING: thread 0 stamps s_memtime at the stage boundaries) on the frame's two shapes:
the NMS-free decode (1544 queries x 26 classes, K = 300) and the memory update (1544 scores, K = 256).  Also times the shipped library.

  python tools/topk_phase_times.py            # on a GPU box
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_phase_times as cpt  # noqa: E402


def timeit(fn, iters=200):
    import torch
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    import torch
    from far3d_amd import lib as flib
    prof = "--shipped" not in sys.argv
    if prof:
        if not os.path.exists(cpt.PROF) or "--build" in sys.argv:
            cpt.build_prof()
        flib.LIB_PATH = cpt.PROF
    lib = flib.load()
    from far3d_amd import ops
    fn = None
    if prof:
        fn = lib.far3d_prof_set_topk_timestamps
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    A, ncls = 1544, 26
    cases = []
    for name, sd in (("normal(-4.6, 0.5)", 0.5), ("normal(-4.6, 0.02)", 0.02)):
        cls = (torch.randn(A, ncls, generator=g) * sd - 4.6).to(dev)
        box = torch.randn(A, 10, generator=g).to(dev)
        cases.append(("decode_topk n=%d K=300 %s" % (A * ncls, name), lambda cls=cls, box=box: ops.decode_topk(cls, box, 300, [-152.4, -152.4, -5.0, 152.4, 152.4, 5.0])))
        sc = cls.max(-1).values.contiguous()
        cases.append(("topk n=%d K=256 %s" % (A, name), lambda sc=sc: ops.topk(sc, 256)))
    names = ["load", "bisect", "compact", "rank stage 1", "rank stage 2 + place"]
    for name, f in cases:
        us = timeit(f)
        line = "%-44s %6.1f us per call (HIP events, back to back)" % (name, us)
        if prof:
            ts = torch.zeros(8, dtype=torch.int64, device=dev)
            fn(ctypes.c_void_p(ts.data_ptr()))
            f()
            torch.cuda.synchronize()
            fn(None)
            t = ts.cpu().numpy().astype(np.int64)
            d = np.diff(t[:6])
            line += " | ticks: " + "  ".join("%s %d" % (n, x) for n, x in zip(names, d)) + "  | passes %d  candidates %d  total %d" % (t[7], t[6], t[5] - t[0])
        print(line, flush=True)


if __name__ == "__main__":
    main()
