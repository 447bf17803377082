"""A/B of the persistent wave-specialised 3x3 kernel (csrc/conv_ws.hpp, tiles 400+) against the shipped tiles on the layers of the
VoV-99 640x960x7 frame: bitwise comparison of the outputs (same products in the same order) and interleaved hipGraph timings.
  python tools/probe/ws_conv_ab.py [pair|bf16] [rounds]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import lib as _flib
if os.environ.get("ABLATE"):      # the timing-only ablations exist in the profiling build only (tools/conv_phase_times.py builds it)
    _flib.LIB_PATH = os.path.join(os.path.dirname(_flib.LIB_PATH), "libfar3d_hip_prof.so")
from far3d_amd import ops

MODE = sys.argv[1] if len(sys.argv) > 1 else "pair"
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
pair = MODE == "pair"
# name, N, H, W, Cin, Cout, shipped tile(s), ws tiles
LAYERS = [
    ("stem2", 7, 320, 480, 64, 64, (163,), (406, 413, 453, 454, 451)),
    ("s2.c1", 7, 160, 240, 128, 128, (163,), (400, 406, 411, 450, 451, 452)),
    ("s3.c0", 7, 80, 120, 256, 160, (165,), (414, 451, 458)),
    ("s3.c1", 7, 80, 120, 160, 160, (152,), (414, 451, 458)),
    ("s3.c0b", 7, 80, 120, 512, 160, (192,), (414, 451, 458)),
    ("s4.c1", 7, 40, 60, 192, 192, (163,), (403, 404, 457, 458, 459, 451)),
    ("s4.c0", 7, 40, 60, 768, 192, (160,), (403, 404, 457, 458, 459, 451)),
    ("c256.l0", 7, 80, 120, 256, 256, (163,), (400, 406, 450, 451, 452)),
    ("c512.l0", 7, 80, 120, 256, 512, (163,), (400, 450, 451, 452)),
    ("c256.l1", 7, 40, 60, 256, 256, (163,), (400, 410, 450, 451, 457, 459)),
] if pair else [
    ("stem2", 7, 320, 480, 64, 64, (63,), (421,)),
    ("s2.c1", 7, 160, 240, 128, 128, (60,), (420,)),
    ("s3.c1", 7, 80, 120, 160, 160, (92,), (422,)),
    ("s4.c1", 7, 40, 60, 192, 192, (60, 100), (423,)),
]
if os.environ.get("ONLY"):
    LAYERS = [l for l in LAYERS if l[0].startswith(os.environ["ONLY"])]


def graph_of(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g, iters


def time_graph(g, iters, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * iters)      # us


for name, N, H, W, Cin, Cout, old, new in LAYERS:
    torch.manual_seed(1)
    x = torch.randn(N, H, W, Cin, device=dev)
    if os.environ.get("ZERO"):       # DVFS probe: the same launches on zero-filled operands (MI355X guide: a zero-filled GEMM clocks ~20 % higher)
        x.zero_()
    tdt = torch.float32 if pair else torch.bfloat16
    xs = ops.pair_from_float(x) if pair else x.to(tdt)
    pc = ops.PackedConv(torch.randn(Cout, Cin, 3, 3) * (0.0 if os.environ.get("ZERO") else 0.05), torch.randn(Cout) * (0.0 if os.environ.get("ZERO") else 1.0), stride=1, pad=1, dtype=tdt, device=dev, compute="bf16x3" if pair else None)
    cs = 2 if pair else 1
    # outputs are channel slices of a wider buffer (what the OSA concat buffers are)
    outs = {}
    for tl in old + new:
        buf = torch.zeros(N, H, W, (Cout + 64) * cs, device=dev, dtype=torch.bfloat16)
        outs[tl] = buf[..., 32 * cs:(32 + Cout) * cs]
    fl = 2.0 * N * H * W * Cout * Cin * 9
    ref = None
    graphs = {}
    for tl in old + new:
        try:
            ops.conv2d_nhwc(xs, pc, out=outs[tl], act="relu", tile=tl)
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            print("%-8s tile %d: %s" % (name, tl, str(e).splitlines()[0][:150]))
            continue
        if ref is None:
            ref = outs[tl].clone()
        same = torch.equal(outs[tl], ref)
        md = (ops.pair_to_float(outs[tl].contiguous()) - ops.pair_to_float(ref.contiguous())).abs().max().item() if pair else \
            (outs[tl].float() - ref.float()).abs().max().item()
        graphs[tl] = graph_of(lambda tl=tl: ops.conv2d_nhwc(xs, pc, out=outs[tl], act="relu", tile=tl)) + (same, md)
    res = {tl: [] for tl in graphs}
    for _ in range(ROUNDS):
        for tl, (g, it, _, _) in graphs.items():
            res[tl].append(time_graph(g, it))
    line = "%-8s %dx%dx%d %d->%d |" % (name, N, H, W, Cin, Cout)
    for tl in graphs:
        ts = sorted(res[tl])
        line += " t%d %6.1f us (min %6.1f) %5.0f TF/s%s |" % (tl, ts[len(ts) // 2], ts[0], fl / (ts[len(ts) // 2] * 1e-6) / 1e12,
                                                             "" if graphs[tl][2] else " DIFF %.2e" % graphs[tl][3])
    print(line, flush=True)
    if os.environ.get("ABLATE"):
        # timing-only ablations of the ws tiles (graphs re-captured: the mask is a launch argument)
        import ctypes
        h = _flib.load()
        for mask, what in ((1, "no DMA (consumers alone)"), (2, "no reads / MFMAs (producers alone)"), (3, "barriers only")):
            h.far3d_conv_ws_set_ablate(ctypes.c_int(mask))
            line = "   ablate %d %-36s |" % (mask, what)
            for tl in new:
                if tl not in graphs:
                    continue
                g, it = graph_of(lambda tl=tl: ops.conv2d_nhwc(xs, pc, out=outs[tl], act="relu", tile=tl))
                ts = sorted(time_graph(g, it) for _ in range(3))
                line += " t%d %6.1f us |" % (tl, ts[1])
            print(line, flush=True)
        h.far3d_conv_ws_set_ablate(ctypes.c_int(0))
