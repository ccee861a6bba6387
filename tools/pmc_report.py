"""Aggregate a rocprofv3 counter_collection.csv per kernel (mean per launch)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("gg::k_") and "fill" not in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
