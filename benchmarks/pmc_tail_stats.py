"""Sum the PMC counters of the LAST part of a rocprofv3 counter-collection csv, per kernel.
usage: pmc_tail_stats.py <counter_collection.csv> [fraction=0.3]"""
import csv, sys, collections, json
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = list(csv.DictReader(open(path)))
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
lo = ids[int(len(ids) * (1.0 - frac))]
agg = collections.OrderedDict()
for r in rows:
    if int(r["Dispatch_Id"]) < lo:
        continue
    k = r["Kernel_Name"]
    k = k[k.find("k_"):][:40] if "k_" in k else k[:40]
    a = agg.setdefault(k, collections.Counter())
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    a["_rows"] += 1
out = {}
for k, a in agg.items():
    n_c = len([c for c in a if c != "_rows"])
    d = {c: v for c, v in a.items() if c != "_rows"}
    d["dispatches"] = a["_rows"] / max(n_c, 1)
    out[k] = d
print(json.dumps(out, indent=1))
