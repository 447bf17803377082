"""Where does the time of the pipelined convolution / GEMM kernels go?  Timing-only ablations (far3d_amd/csrc/igemm_kernels.hpp,
FAR3D_ABLATE, a bit mask): the same launch with ingredients removed -- 1 the epilogue, 2 the LDS-DMA inside the K loop, 4 the MFMAs
(and the fragment reads that feed them), 8 the fragment reads only, 16 the barrier + vmcnt wait at the top of every step; 26 = MFMA
loop + epilogue only, 27 = the bare MFMA loop, 21 = the bare DMA stream.  The outputs of an ablated launch are wrong by construction;
only its duration is used.

  python tools/conv_ablation.py build        # here (no GPU needed): far3d_amd/libfar3d_hip_abl<k>.so for every variant
  python tools/conv_ablation.py run <k>      # on the GPU box: one line per layer for variant k (0 = the shipped library)
  python tools/conv_ablation.py all          # on the GPU box: every variant in its own process, one table
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {0: "shipped", 1: "no epilogue", 2: "no in-loop DMA", 4: "no MFMA / fragment reads", 8: "no fragment reads", 16: "no barrier / vmcnt wait",
            26: "MFMA loop + epilogue", 27: "bare MFMA loop", 21: "bare DMA stream", 5: "DMA + barriers, no epilogue", 11: "MFMA + barriers",
            # GEMM kernels with full-line DMA (gemm1x1_wide_kernel: the concat layers' shipped tiles) only -- which of the two streams is
            # the slow one: 32 = no weight stream (L2-resident, re-read by every workgroup), 64 = no activation stream (HBM / MALL, read once)
            37: "activation stream alone (bare DMA, no weights)", 69: "weight stream alone (bare DMA, no activations)",
            32: "no weight stream", 64: "no activation stream"}
# (name, N, H, W, Cin, Cout, k, tile): tile 0 = the shipped tile of the layer at the benchmarked size (far3d_amd/data/tuning_mi355x.json;
# FAR3D_ABL_TABLE=tuning_mi355x_tput.json for the table of the pipelined engines); a fixed tile id pins a kernel for an experiment
LAYERS = [("s2.c1", 7, 160, 240, 128, 128, 3, 0), ("s3.c1", 7, 80, 120, 160, 160, 3, 0), ("s3.c0", 7, 80, 120, 512, 160, 3, 0),
          ("s4.c1", 7, 40, 60, 192, 192, 3, 0), ("s4.c0", 7, 40, 60, 768, 192, 3, 0), ("s5.c1", 7, 20, 30, 224, 224, 3, 0),
          ("s2.cat", 7, 160, 240, 768, 256, 1, 0), ("s3.cat", 7, 80, 120, 1312, 512, 1, 0), ("s4.cat", 7, 40, 60, 1728, 768, 1, 0),
          ("s5.cat", 7, 20, 30, 2144, 1024, 1, 0)]


def lib_path(k):
    return os.path.join(ROOT, "far3d_amd", "libfar3d_hip.so" if k == 0 else "libfar3d_hip_abl%d.so" % k)


def build():
    from far3d_amd import build as fbuild
    fbuild.build(verbose=False)
    objs = [os.path.join(fbuild.OBJ, f) for f in os.listdir(fbuild.OBJ) if f.endswith(".o") and f != "igemm.o"]
    todo = [v for v in VARIANTS if v]
    for i in range(0, len(todo), 5):          # five compiles at a time (fourteen at once ran the container out of memory)
        procs = []
        for k in todo[i:i + 5]:
            o = os.path.join("/tmp", "igemm_abl%d.o" % k)
            procs.append((k, o, subprocess.Popen([fbuild.HIPCC] + fbuild.FLAGS + ["-DFAR3D_ABLATE=%d" % k, "-c", os.path.join(fbuild.CSRC, "igemm.hip"), "-o", o])))
        for k, o, p in procs:
            assert p.wait() == 0, "variant %d failed to compile" % k
            subprocess.run([fbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(k), o] + objs, check=True)
            print("built", lib_path(k), flush=True)


def run(k):
    import torch
    from far3d_amd import lib as flib
    flib.LIB_PATH = lib_path(k)
    from far3d_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tune_conv import timeit
    dev = "cuda:0"
    out = {}
    if os.environ.get("FAR3D_ABL_TABLE"):
        ops.BF16_TILE_TABLE = os.environ["FAR3D_ABL_TABLE"]      # the process default (no engine selection is active here)
    for name, N, H, W, Cin, Cout, ks, tile in LAYERS:
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        pc = ops.PackedConv(torch.randn(Cout, Cin, ks, ks) * 0.05, torch.randn(Cout), stride=1, pad=ks // 2, dtype=torch.bfloat16, device=dev)
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        out[name] = timeit(lambda: ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile)) * 1e6
    print(json.dumps({"variant": k, "us": out}))


def all_variants():
    rows = {}
    for k in VARIANTS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", str(k)], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("variant %d failed: %s" % (k, r.stderr[-300:]))
            continue
        rows[k] = json.loads(line[-1])["us"]
    names = [l[0] for l in LAYERS]
    fl = {l[0]: 2.0 * l[1] * l[2] * l[3] * l[4] * l[5] * l[6] * l[6] for l in LAYERS}
    print("%-28s" % "launch time, us" + "".join("%9s" % n for n in names))
    for k, us in rows.items():
        print("%-28s" % ("%d %s" % (k, VARIANTS[k])) + "".join("%9.1f" % us[n] for n in names))
    if 0 in rows:
        print("%-28s" % "shipped, TF/s" + "".join("%9.0f" % (fl[n] / rows[0][n] / 1e6) for n in names))
        print("%-28s" % "MFMA time at 2.5 PF, us" + "".join("%9.1f" % (fl[n] / 2.5e15 * 1e6) for n in names))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        all_variants()
