import csv, glob, sys, collections
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "brick" in k or "exact" in k or "queue" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
