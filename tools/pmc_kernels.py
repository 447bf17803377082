"""Per-kernel averages of rocprofv3 --pmc counter passes (one or more output directories), for the kernels whose name contains one
of the given substrings.  Used for the MFMA-vs-LDS diagnosis of the convolution kernels:

  python tools/pmc_kernels.py <out.txt> <pass dir> [<pass dir> ...] -- conv3x3_pipe gemm1x1_pipe

Derived columns (when their counters are present): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) is not
available without the CU-busy counter, so the ratios printed are per wave-cycle: X / SQ_WAVE_CYCLES (quad-cycle units as the
counters report them, MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys


def main():
    out = sys.argv[1]
    sep = sys.argv.index("--")
    dirs, subs = sys.argv[2:sep], sys.argv[sep + 1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if any(s in name for s in subs):
                    acc[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    for name in sorted(acc, key=lambda n: -sum(acc[n].get("SQ_WAVE_CYCLES", [0]))):
        c = {k: sum(v) / len(v) for k, v in acc[name].items()}
        n = max(len(v) for v in acc[name].values())
        wc = c.get("SQ_WAVE_CYCLES")
        row = "%-70s launches %4d" % (name[:70], n)
        for k in sorted(c):
            row += "  %s %.4g" % (k, c[k])
            if wc and k != "SQ_WAVE_CYCLES":
                row += " (%.3f/wave-cycle)" % (c[k] / wc)
        lines.append(row)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(l[:400] for l in lines[:12]))


if __name__ == "__main__":
    main()
