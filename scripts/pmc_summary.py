"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name (k_* kernels only)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if not name.startswith("void k_") and "k_" not in name[:12]:
                continue
            acc[name.split("(")[0][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
