"""Turn the rocprofv3 --pmc passes of `bench.py --eager` (one counter per pass) into profiles/<round>/aggregate_pmc.json: HBM
bytes per launch of the aggregation kernel, measured INSIDE benchmark frames (value maps written by the FPN of the same frame,
other kernels between the six launches), stamped with the kernel name and the commit.

  python tools/pmc_to_json.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> [<dir TCC pass>] <kernel substring> <out.json>

FETCH_SIZE is doubled on gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: the counter tallies 128-byte requests at
64 bytes); WRITE_SIZE is taken as reported (the guide calls it uncalibrated)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def mean_counter(d, kernel, counters):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] in counters:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    args = sys.argv[1:]
    out, kernel = args[-1], args[-2]
    fetch = mean_counter(args[0], kernel, ("FETCH_SIZE",))
    write = mean_counter(args[1], kernel, ("WRITE_SIZE",))
    tcc = mean_counter(args[2], kernel, ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum")) if len(args) > 4 else {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    if not commit:      # the GPU box has no .git: the stamp far3d_amd/build.py wrote (the same source bench.py reports)
        try:
            commit = open(os.path.join(root, "far3d_amd", "_build_commit.txt")).read().strip()
        except OSError:
            commit = ""
    j = dict(kernel=kernel, commit=commit or os.environ.get("FAR3D_COMMIT", "?"),
             source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --eager`: every aggregation launch of "
                    "every benchmark frame in the trace (in-frame traffic, not an isolated micro-benchmark)",
             launches=fetch.get("FETCH_SIZE", (0, 0))[1],
             FETCH_SIZE_KB=fetch.get("FETCH_SIZE", (None, 0))[0], WRITE_SIZE_KB=write.get("WRITE_SIZE", (None, 0))[0],
             fetch_correction="x2 on gfx950 (MI355X_MICROARCH.md, HBM section)")
    if j["FETCH_SIZE_KB"] is not None and j["WRITE_SIZE_KB"] is not None:
        j["hbm_bytes_per_launch"] = int(j["FETCH_SIZE_KB"] * 1024 * 2 + j["WRITE_SIZE_KB"] * 1024)
    if "TCC_HIT_sum" in tcc and "TCC_MISS_sum" in tcc:
        h, m = tcc["TCC_HIT_sum"][0], tcc["TCC_MISS_sum"][0]
        j["l2_hit_rate"] = h / max(1.0, h + m)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
