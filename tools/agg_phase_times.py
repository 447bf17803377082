"""Where does one query's time go inside the aggregation kernel?  Builds libfar3d_hip_prof.so (the same sources with
-DFAR3D_PROFILING: per-wave s_memtime stamps at the phase boundaries), runs the kernel on the config-2 case and prints the
median per-phase times.  Tools only -- the shipped libfar3d_hip.so has no such hooks."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import build as fbuild  # noqa: E402
from far3d_amd import lib as flib  # noqa: E402

PROF = os.path.join(ROOT, "far3d_amd", "libfar3d_hip_prof.so")


def build_prof():
    fbuild.build_profiling()


def main():
    """argv: [A] [variant] [operands.pt] [layer] [raw_out.npz] -- with an operands file (tools/dump_agg_operands.py) the kernel runs on
    the live operands of that decoder layer of a benchmark frame instead of the seeded test case."""
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 1544
    vs = sys.argv[2] if len(sys.argv) > 2 else "7"
    srt = vs.endswith("s")                     # "8s": kernel 8 in sorted mode (operands in launch order, hoisted projection)
    variant, nw = int(vs.rstrip("s")), 2
    opfile = sys.argv[3] if len(sys.argv) > 3 else None
    layer = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    raw_out = sys.argv[5] if len(sys.argv) > 5 else None
    if not os.path.exists(PROF) or os.path.getmtime(PROF) < os.path.getmtime(os.path.join(fbuild.CSRC, "sampling.hip")):
        build_prof()
    flib.LIB_PATH = PROF
    lib = flib.load()
    from far3d_amd import ops
    from tests import cases
    dev = "cuda:0"
    d = lambda t: t.to(dev).contiguous()
    if opfile:
        c = torch.load(opfile)["layers"][layer]
        A = c["ref"].shape[0]
        feat = torch.randn(c["feat_shape"], device=dev).to(torch.bfloat16)
        print("live operands: %s layer %d, A = %d" % (opfile, layer, A))
    else:
        c = cases.aggregate_case(num_cams=7, pad_hw=(640, 960), A=A, seed=0)
        feat = d(c["feat"].to(torch.bfloat16))
    args = [d(c[k]) for k in ("ref", "offsets", "lidar2img", "U", "Vc")]
    perm = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"])
    nblk = 8 * ((A + 7) // 8)
    ts = torch.zeros(nblk * 4 * 16, dtype=torch.int64, device=dev)
    tab = ops.agg_tables(args[4])
    qbase = None
    if srt:
        perm, (inv, qbase) = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"], sorted_operands=True)
        offs2, U2 = args[1].reshape(A, -1), args[3]
        Os, Us = torch.empty_like(offs2), torch.empty_like(U2)
        Os[inv.long()], Us[inv.long()] = offs2, U2
        args[1], args[3] = Os, Us
        print("sorted mode: operands in launch order")
    run = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, variant=variant, tables=tab,
                                        qbase=qbase)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    fn = lib.far3d_prof_set_agg_timestamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    assert fn(ctypes.c_void_p(ts.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    t = ts.cpu().numpy().reshape(nblk, 4, 16)[:, :nw]
    if variant in (8, 9):     # finer account of v8's build phase: weights of the first batch, then per-item times (they include list flushes = gathers)
        w = (t[:, :, 11] - t[:, :, 3]).astype(np.float64)
        w = w[t[:, :, 11] > 0]
        if w.size:       # (sorted mode forms the weights in front of the barrier: no such stamp)
            print("  v8 build: first batch's weights (eV wait + products)  median %6.0f p90 %6.0f" % (np.median(w), np.percentile(w, 90)))
        for nm, i in (("patch item", 12), ("per-corner item", 14)):
            n = t[:, :, i + 1].sum()
            print("  v8 build: %-16s count %6d  mean %7.0f ticks" % (nm, n, t[:, :, i].sum() / max(n, 1)))
        ld = t[:, :, 10].astype(np.float64)
        tot_ = (t[:, :, 5] - t[:, :, 3]).astype(np.float64)
        print("  v8 work estimate per wave: median %.0f p90 %.0f max %.0f; corr(estimate, build+gather ticks) = %.3f" %
              (np.median(ld), np.percentile(ld, 90), ld.max(), np.corrcoef(ld.flatten(), tot_.flatten())[0, 1]))
        print("  v8 |wave0 - wave1| build+gather ticks: median %.0f p90 %.0f" % (np.median(np.abs(tot_[:, 0] - tot_[:, 1])), np.percentile(np.abs(tot_[:, 0] - tot_[:, 1]), 90)))
    if raw_out:      # raw stamps + the workgroup -> query map, for offline analysis of the tail
        np.savez_compressed(raw_out, stamps=t, perm=perm.cpu().numpy(), q_per_xcd=(A + 7) // 8)
    live = t[:, :, 0] != 0
    t = t[live[:, 0]]
    if variant in (8, 9):      # v8's specialised front end, per wave (round 6 stamps): wave 0 projects and deals, wave 1 does the softmax statistics
        for w, names in ((0, ("issue loads", "wait + projection + bbox", "deal + descriptors -> LDS", "barrier + read back", "first eV issue")),
                         (1, ("issue loads", "wait + softmax statistics", "e^U (sorted: + the two hinted cameras' weights) -> LDS", "barrier + read back", "first eV issue"))):
            for nm, (i, j) in zip(names, ((0, 8), (8, 1), (1, 2), (2, 9), (9, 3))):
                dlt = (t[:, w, j] - t[:, w, i]).astype(np.float64)
                print("  front end wave %d: %-28s median %7.0f  p90 %7.0f  max %7.0f ticks" % (w, nm, np.median(dlt), np.percentile(dlt, 90), dlt.max()))
        fe = (t[:, :, 3] - t[:, :, 0]).astype(np.float64)
        print("  front end (start -> first eV issue), both waves: median %.0f p90 %.0f" % (np.median(fe), np.percentile(fe, 90)))
    if variant in (7, 11):
        for nm, i, j in (("  issue logit loads", 0, 8), ("  projection+bbox", 8, 9), ("  local max", 9, 1)):
            dlt = (t[:, :, j] - t[:, :, i]).astype(np.float64)
            print("  %-22s median %8.0f  p90 %8.0f  max %8.0f ticks" % (nm, np.median(dlt), np.percentile(dlt, 90), dlt.max()))
    t0 = t[:, :, 0].min()
    names = ["loads+proj+localmax", "B1 wait", "exp+sums+B2", "build", "gather", "reduce+B3"]
    if variant in (8, 9):
        names = ["loads + projection | softmax", "deal | e^U -> LDS", "barrier + exchange + eV issue", "build", "gather", "reduce + barrier"]
    print("queries %d, workgroups stamped %d; s_memtime ticks (shader-clock cycles, ~2.1 GHz under this kernel: a 23 us launch spans ~48k ticks)" % (A, len(t)))
    for i, nm in enumerate(names):
        dlt = (t[:, :, i + 1] - t[:, :, i]).astype(np.float64)
        print("  %-30s median %8.0f  p90 %8.0f  max %8.0f ticks" % (nm, np.median(dlt), np.percentile(dlt, 90), dlt.max()))
    tot = (t[:, :, 6] - t[:, :, 0]).astype(np.float64)
    print("  %-22s median %8.0f  p90 %8.0f  max %8.0f ticks" % ("whole wave", np.median(tot), np.percentile(tot, 90), tot.max()))
    print("  (stamps of different CUs are not mutually synchronised: only differences within one wave are meaningful)")
    busy = t[:, :, 7] > 0
    bld = (t[:, :, 4] - t[:, :, 3]).astype(np.float64)
    gth = (t[:, :, 5] - t[:, :, 4]).astype(np.float64)
    print("  waves with rows: %d of %d; their build median %.0f p90 %.0f; gather median %.0f p90 %.0f" %
          (busy.sum(), busy.size, np.median(bld[busy]), np.percentile(bld[busy], 90), np.median(gth[busy]), np.percentile(gth[busy], 90)))
    print("  list entries per wave: median %d  p90 %d  max %d; per query median %d" %
          (np.median(t[:, :, 7]), np.percentile(t[:, :, 7], 90), t[:, :, 7].max(), np.median(t[:, :, 7].sum(1))))


if __name__ == "__main__":
    main()
