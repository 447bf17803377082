"""Where does a persistent wave-specialised conv workgroup (csrc/conv_ws.hpp) spend its life?  Profiling build (libfar3d_hip_prof.so,
tools/conv_phase_times.py build_prof): per-workgroup record of consumer wave 0 -- cycles between arriving at a step barrier and leaving
it, cycles in the step bodies (fragment reads + MFMA issue), cycles in the epilogues -- and the shader clock (s_memtime / s_memrealtime).
  python tools/probe/ws_conv_prof.py [layer-prefix]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far3d_amd import lib as flib
flib.LIB_PATH = os.path.join(os.path.dirname(flib.LIB_PATH), "libfar3d_hip_prof.so")
from far3d_amd import ops

LAYERS = [("stem2", 7, 320, 480, 64, 64, (401, 413)), ("s2.c1", 7, 160, 240, 128, 128, (400, 411, 450, 451)),
          ("s3.c1", 7, 80, 120, 160, 160, (407, 414)), ("s4.c1", 7, 40, 60, 192, 192, (403, 404, 451)), ("c512.l0", 7, 80, 120, 256, 512, (400,))]
only = sys.argv[1] if len(sys.argv) > 1 else None
lib = flib.load()
setp = lib.far3d_prof_set_conv_timestamps
setp.restype, setp.argtypes = ctypes.c_int, [ctypes.c_void_p]
abl = lib.far3d_conv_ws_set_ablate
dev = "cuda:0"
for name, N, H, W, Cin, Cout, tiles in LAYERS:
    if only and not name.startswith(only):
        continue
    x = ops.pair_from_float(torch.randn(N, H, W, Cin, device=dev))
    pc = ops.PackedConv(torch.randn(Cout, Cin, 3, 3) * 0.05, torch.randn(Cout), stride=1, pad=1, dtype=torch.float32, device=dev, compute="bf16x3")
    y = torch.empty(N, H, W, 2 * Cout, device=dev, dtype=torch.bfloat16)
    for tile in tiles:
        for mask in (0, 1):
            abl(ctypes.c_int(mask))
            ts = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile)
            torch.cuda.synchronize()
            setp(ctypes.c_void_p(ts.data_ptr()))
            ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile)
            torch.cuda.synchronize()
            setp(ctypes.c_void_p(0))
            r = ts.cpu().numpy().reshape(-1, 16)
            r = r[r[:, 5] > 0]
            cyc = (r[:, 3] - r[:, 1]).astype(np.float64)
            us = (r[:, 7] - r[:, 6]).astype(np.float64) / 100.0
            steps = r[:, 5].astype(np.float64)
            span = (r[:, 7].max() - r[:, 6].min()) / 100.0
            print("%-8s t%d %s: %d workgroups, launch span %.1f us | per workgroup (median): %.0f steps, life %.1f us, clock %.0f MHz, "
                  "%.0f cycles / step = barrier wait %.0f + body %.0f, epilogues %.1f %% of life, first fill %.0f cycles" %
                  (name, tile, {0: "full          ", 1: "no-DMA        ", 5: "no-DMA no-LDS ", 13: "MFMA only      ", 9: "no-DMA no-barr"}[mask], len(r), span, np.median(steps), np.median(us), np.median(cyc / us),
                   np.median(cyc / steps), np.median(r[:, 8] / steps), np.median(r[:, 9] / steps), 100 * np.median(r[:, 10] / cyc),
                   np.median(r[:, 2] - r[:, 1])), flush=True)
abl(ctypes.c_int(0))
