"""Summarise rocprofv3 --pmc counter_collection.csv files: per (kernel, grid) mean of every counter."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "igemm" not in k and "conv3x3" not in k and "aggregate" not in k and "gemm" not in k:
                continue
            key = (k.split("(")[0].replace("void ", "")[:60], r["Grid_Size"], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print("    %-36s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
