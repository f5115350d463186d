import csv, glob, collections, os, sys
base = sys.argv[1] if len(sys.argv) > 1 else "prof_q"
f = sorted(glob.glob(f"gpurun_out/{base}/*/*kernel_stats.csv"), key=os.path.getmtime)
if f:
    for line in open(f[-1]):
        if "bm25" in line or "merge" in line or "Name" in line:
            print(line.strip()[:170])
for d in (base + "_sq", base + "_sq2"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "group" in k or "_wl" in k:
            print(k, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
