"""Summarise rocprofv3 --pmc counter_collection csv files: per kernel, mean of each counter over dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "mcica" not in k and len(sys.argv) > 1 and "--all" not in sys.argv:
        pass
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
