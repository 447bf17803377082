"""Where does a workgroup of the pipelined conv / GEMM kernels spend its life?  Builds libfar3d_hip_prof.so (-DFAR3D_PROFILING: thread 0 of
every workgroup stamps s_memtime at the phase boundaries, csrc/igemm_kernels.hpp FAR3D_CONV_TS), runs single launches of the VoV-99 layers
at the benchmarked size and prints, per (layer, tile): the phase medians, the workgroup lifetime, how many workgroups a CU ran and how many
of them overlapped.  Tools only -- the shipped libfar3d_hip.so has no such hooks.

  python tools/conv_phase_times.py build            # here (no GPU needed)
  python tools/conv_phase_times.py [layer ...]      # on the GPU box
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import build as fbuild  # noqa: E402

PROF = os.path.join(ROOT, "far3d_amd", "libfar3d_hip_prof.so")
# (name, N, H, W, Cin, Cout, k, tiles)
LAYERS = [("s2.c1", 7, 160, 240, 128, 128, 3, (60, 130)), ("s3.c1", 7, 80, 120, 160, 160, 3, (92,)), ("s3.c0", 7, 80, 120, 512, 160, 3, (101,)),
          ("s4.c1", 7, 40, 60, 192, 192, 3, (60, 100)), ("s4.c0", 7, 40, 60, 768, 192, 3, (100,)), ("s5.c1", 7, 20, 30, 224, 224, 3, (102,)),
          ("s2.cat", 7, 160, 240, 768, 256, 1, (123, 145)), ("s3.cat", 7, 80, 120, 1312, 512, 1, (120,)), ("s4.cat", 7, 40, 60, 1728, 768, 1, (123, 145, 120))]


def build_prof():
    fbuild.build_profiling()


def main():
    import torch
    from far3d_amd import lib as flib
    flib.LIB_PATH = PROF
    lib = flib.load()
    from far3d_amd import ops
    fn = lib.far3d_prof_set_conv_timestamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    dev = "cuda:0"
    only = sys.argv[1:]
    for name, N, H, W, Cin, Cout, k, tiles in LAYERS:
        if only and name not in only:
            continue
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        pc = ops.PackedConv(torch.randn(Cout, Cin, k, k) * 0.05, torch.randn(Cout), stride=1, pad=k // 2, dtype=torch.bfloat16, device=dev)
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        sums = torch.zeros(N, Cout, dtype=torch.int64, device=dev) if k == 1 else None
        for tile in tiles:
            ts = torch.zeros(1 << 20, dtype=torch.int64, device=dev)       # 128 Ki workgroups x 8 stamps
            fn(None)
            for _ in range(3):
                ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile, sums=sums)
            torch.cuda.synchronize()
            fn(ctypes.c_void_p(ts.data_ptr()))
            ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile, sums=sums)
            torch.cuda.synchronize()
            fn(None)
            t = ts.cpu().numpy().astype(np.uint64).reshape(-1, 8)
            t = t[t[:, 1] != 0]
            if not len(t):
                print("%-7s tile %3d: no stamps (kernel without hooks)" % (name, tile))
                continue
            hw, xcc = (t[:, 0] & np.uint64(0xffffffff)).astype(np.int64), (t[:, 0] >> np.uint64(32)).astype(np.int64) & 0xf
            cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
            tt = t[:, 1:6].astype(np.int64)
            real = t[:, 6].astype(np.int64)
            # a kernel that does not take one of the stamps leaves the slot zero (tile 145 has no "first fill" hook: round 5's report
            # printed the difference to the zero as a negative phase): an unset stamp takes the previous stamp's value, its phase reads
            # 0 and is said to be unset; a stamp that runs backwards is an error of the hooks, not something to print
            unset = [int(k) for k in range(1, 5) if (tt[:, k] == 0).all()]
            for k in range(1, 5):
                z = tt[:, k] == 0
                tt[z, k] = tt[z, k - 1]
            assert (np.diff(tt, axis=1) >= 0).all(), "%s tile %d: phase stamps run backwards" % (name, tile)
            os.makedirs(os.path.join(ROOT, "gpurun_out", "conv_phases"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "conv_phases", "%s_t%d.npz" % (name, tile)), stamps=t)
            # s_memtime is a per-XCD counter: clock rate, spans and overlaps are evaluated per CU and then combined
            ph = np.diff(tt, axis=1)                    # set-up, first fill, K loop, epilogue
            life = tt[:, 4] - tt[:, 0]
            med = lambda a: float(np.median(a))
            cus = np.unique(cu)
            rates, spans, busy1, busy2, nper = [], [], 0.0, 0.0, []
            for c in cus:
                m = cu == c
                nper.append(int(m.sum()))
                sp = tt[m, 4].max() - tt[m, 0].min()
                spans.append(sp)
                dr = real[m].max() - real[m].min()
                if dr > 50:                              # >= 0.5 us between the first and the last entry stamp of this CU
                    rates.append(100.0 * (tt[m, 0].max() - tt[m, 0].min()) / dr)
                ev = sorted([(a, 1) for a in tt[m, 0]] + [(b, -1) for b in tt[m, 4]])
                alive, last = 0, ev[0][0]
                for when, d in ev:
                    if alive >= 1:
                        busy1 += when - last
                    if alive >= 2:
                        busy2 += when - last
                    alive += d
                    last = when
            if not rates:
                # one round of workgroups, one per CU: no two entry stamps on a CU.  s_memtime is a per-XCD counter, so the workgroups of one
                # XCD give the rate as well (their entries are spread over the dispatch time of the grid)
                for xc in np.unique(xcc):
                    m = xcc == xc
                    dr = real[m].max() - real[m].min()
                    if dr >= 5:
                        rates.append(100.0 * (tt[m, 0].max() - tt[m, 0].min()) / dr)
            mhz = float(np.median(rates)) if rates else 0.0          # 0: no two entry stamps far enough apart to measure the clock
            us = (lambda ticks: ticks / mhz) if rates else (lambda ticks: 0.0)
            tot = float(np.sum(spans))
            print("%-7s tile %3d: %5d workgroups on %3d CUs (max %d per CU) | s_memtime %5.0f MHz (0 = not measurable: times in us read 0) | per-CU span median %6.1f us (%d ticks) | median per "
                  "workgroup [ticks]: set-up %5.0f  first fill %5.0f  K loop %6.0f  epilogue %5.0f  life %6.0f (p10 %.0f p90 %.0f) = %.2f us | "
                  "a CU has >=1 workgroup alive %4.1f %% of its span, >=2 alive %4.1f %%%s" %
                  (name, tile, len(t), len(cus), max(nper), mhz, us(med(spans)), med(spans), med(ph[:, 0]), med(ph[:, 1]), med(ph[:, 2]), med(ph[:, 3]),
                   med(life), np.percentile(life, 10), np.percentile(life, 90), us(med(life)), 100.0 * busy1 / tot, 100.0 * busy2 / tot,
                   " | stamps this kernel does not take (phase reads 0): %s" % [("set-up", "first fill", "K loop", "epilogue")[k - 1] for k in unset] if unset else ""), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build_prof()
    else:
        main()
