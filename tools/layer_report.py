"""Per-layer conv efficiency from a rocprofv3 kernel trace of bench.py (last frame in the trace)."""
import csv, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import weights
spec = weights.VOV_SPECS["V-99-eSE"]
N, H, W = 7, 640, 960
layers = []  # (name, npix, cout, K)
h, w = H // 2, W // 2
layers.append(("stem1", N*h*w, 64, 32)); layers.append(("stem2", N*h*w, 64, 64*9))
h, w = h // 2, w // 2
layers.append(("stem3", N*h*w, 128, 64*9))
in_ch = 128
for si in range(4):
    sc, oc = spec["stage_conv_ch"][si], spec["stage_out_ch"][si]
    if si > 0:
        h, w = -(-h // 2), -(-w // 2)
    for b in range(spec["block_per_stage"][si]):
        c = in_ch
        for i in range(5):
            layers.append(("s%d.b%d.c%d" % (si+2, b, i), N*h*w, sc, c*9)); c = sc
        layers.append(("s%d.b%d.cat" % (si+2, b), N*h*w, oc, in_ch + 5*sc)); in_ch = oc
hw = [(80,120),(40,60),(20,30),(10,15)]
for i, cin in zip((2,1,0), (1024,768,512)):
    layers.append(("fpn.lat%d" % i, N*hw[i][0]*hw[i][1], 256, cin))
for i in range(4):
    layers.append(("fpn.out%d" % i, N*hw[i][0]*hw[i][1], 256, 256*9))
for l in range(4):
    p = N*hw[l][0]*hw[l][1]
    # the engine's launch order (engine.roi_head): the two towers' first convs as ONE 256 -> 512 conv, then cls conv + 1x1 head, reg conv + 1x1 head
    # (until round 6 this list still held the reference's six-conv order and the report paired the launches with the wrong layers)
    layers += [("roi%d.tower0" % l, p, 512, 2304), ("roi%d.cls1" % l, p, 256, 2304), ("roi%d.clsh" % l, p, 26, 256),
               ("roi%d.reg1" % l, p, 256, 2304), ("roi%d.regh" % l, p, 5, 256)]
p = N*80*120
layers += [("depth.c0", p, 256, 2304), ("depth.c1", p, 256, 2304), ("depth.cls", p, 51, 256)]
rows = list(csv.DictReader(open(sys.argv[1])))
# rocprofv3 writes the trace in completion-record order, not in launch order: two neighbouring launches can come out swapped (round 5's
# report paired s4.b3.c4 with the concat GEMM's launch and printed 2 460.9 TF/s for the layer after it).  The layers of a frame run on
# ONE stream, so their start times are their order.
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
PEAK_TF = {"bf16": 2500.0, "x3": 2500.0 / 3}
# a frame starts with the stem: far3d_stem_im2col (+ a GEMM) in the fp32 / pair modes, stem_conv_kernel (the fused stem_1, itself layer 0) in bf16
is_conv = lambda r: "igemm" in r["Kernel_Name"] or "conv3x3_" in r["Kernel_Name"] or "gemm1x1" in r["Kernel_Name"] or "stem_conv_kernel" in r["Kernel_Name"]
stems = [i for i, r in enumerate(rows) if "stem_im2col" in r["Kernel_Name"] or "stem_conv_kernel" in r["Kernel_Name"]]
# the LAST COMPLETE frame of the trace: the trace ends with bench.py's backbone-alone replays (stem .. stage 5 only), which start with a
# stem launch too but hold fewer conv launches than a frame has layers
ig = None
for k in range(len(stems) - 1, -1, -1):
    seg = rows[stems[k]: stems[k + 1] if k + 1 < len(stems) else len(rows)]
    cand = [r for r in seg if is_conv(r)]
    if len(cand) >= len(layers):
        ig = cand[:len(layers)]
        break
assert ig is not None, "no complete frame (%d conv launches) in the trace" % len(layers)
tot_t = tot_f = 0
agg = {}
for (name, npix, cout, K), r in zip(layers, ig):
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    fl = 2.0 * npix * cout * K
    # sanity of the pairing: a 3x3 layer runs on a 3x3 / generic kernel, never on the 1x1 GEMM kernel, and the other way round; no
    # layer can exceed the matrix peak of its arithmetic (three MFMAs per product in the pair / split modes)
    kn = r["Kernel_Name"]
    is3 = K % 9 == 0 and not name.endswith((".cat", "clsh", "regh", ".cls")) and not name.startswith("fpn.lat")
    assert not (is3 and "gemm1x1" in kn) and not (not is3 and "conv3x3_" in kn), "layer %s (K=%d) paired with launch %s" % (name, K, kn[:60])
    peak = PEAK_TF["x3"] if ("true" in kn.split("<")[-1] or "pair_t" in kn or "split_t" in kn) else PEAK_TF["bf16"]
    assert t > 0 and fl / t / 1e12 <= peak, "layer %s: %.1f TF/s exceeds the %.0f TF/s peak (%s, %.1f us)" % (name, fl / t / 1e12, peak, kn[:60], t * 1e6)
    var = "S(fused stem)" if "stem_conv_kernel" in r["Kernel_Name"] else \
        ("P" if "conv3x3" in r["Kernel_Name"] else "G" if "gemm1x1" in r["Kernel_Name"] else "D" if "dma" in r["Kernel_Name"] else "R") + r["Kernel_Name"].split("<")[1].split(">")[0].replace("unsigned short", "bf16").replace(" ", "")
    blocks = int(r["Grid_Size_X"]) // max(1, int(r.get("Workgroup_Size_X", 256) or 256)) * int(r["Grid_Size_Y"])
    key = name.split(".")[0]
    a = agg.setdefault(key, [0.0, 0.0]); a[0] += t; a[1] += fl
    tot_t += t; tot_f += fl
    if len(sys.argv) > 2:
        print("%-12s npix=%7d cout=%4d K=%5d %-14s blocks=%5d %8.1f us %7.1f TF/s" % (name, npix, cout, K, var, blocks, t*1e6, fl/t/1e12))
for k, (t, f) in agg.items():
    print("%-8s %8.3f ms %8.1f GFLOP %7.1f TF/s" % (k, t*1e3, f/1e9, f/t/1e12))
print("total conv %.3f ms, %.1f GFLOP, %.1f TF/s" % (tot_t*1e3, tot_f/1e9, tot_f/tot_t/1e12))
bb = [v for k, v in agg.items() if k.startswith("stem") or k in ("s2", "s3", "s4", "s5")]
print("backbone conv %.3f ms, %.1f GFLOP, %.1f TF/s (stem + stages 2-5: the rows round 5's report covered)" %
      (sum(v[0] for v in bb) * 1e3, sum(v[1] for v in bb) / 1e9, sum(v[1] for v in bb) / sum(v[0] for v in bb) / 1e12))
