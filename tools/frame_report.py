"""Per-kernel totals of ONE frame (the last complete one) in a rocprofv3 kernel trace of bench.py, plus idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stem_im2col" in r["Kernel_Name"] or "stem_conv_kernel" in r["Kernel_Name"]]
segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
# whole benchmark frames only: a real frame has one aggregation launch per decoder layer (6); the segments at the end of
# a bench.py trace are polluted by the kernel-timing replays (24 aggregation launches each) and are skipped
nlayers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
segs = [sg for sg in segs if sum("aggregate" in r["Kernel_Name"] for r in rows[sg[0]:sg[1]]) == nlayers]
a, b = segs[-1]
fr = rows[a:b]
tot = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in fr:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    tot[k][0] += 1; tot[k][1] += d; busy += d
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) * 1e-3
foreign = sum(n for k, (n, d) in tot.items() if any(t in k for t in ("at::native", "rocprim", "rocsolver", "hipcub")))
print("frame wall %.1f us, kernel busy %.1f us, idle %.1f us, %d launches, %d aggregation launches, %d ATen/rocPRIM/rocSOLVER launches (%d clean frames in the trace)"
      % (wall, busy, wall - busy, len(fr), sum(n for k, (n, d) in tot.items() if "aggregate" in k), foreign, len(segs)))
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%8.1f us %5d  %s" % (d, n, k))
