"""Per-kernel launch statistics of the LAST part of a rocprofv3 kernel trace (csv): the steady regime of a run that
starts from the opening.  usage: trace_tail_stats.py <kernel_trace.csv> [fraction=0.3]"""
import csv, sys, collections, json
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
sel = [r for r in rows if r[0] >= t_lo]
agg = collections.OrderedDict()
for s, e, k in sel:
    k = k[k.find("k_"):][:28] if "k_" in k else k[:40]
    a = agg.setdefault(k, [0, 0, 0, []])
    a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s); a[3].append(e - s)
span = sel[-1][1] - sel[0][0]
busy = sum(a[1] for a in agg.values())
out = {"window_ms": span / 1e6, "gpu_busy_frac": busy / span, "kernels": {k: {"calls": a[0], "avg_us": a[1] / a[0] / 1e3, "max_us": a[2] / 1e3, "share_of_window": a[1] / span,
                                                                        "quantiles_us_10_50_90_99": [sorted(a[3])[min(len(a[3]) - 1, int(q * len(a[3])))] / 1e3 for q in (0.1, 0.5, 0.9, 0.99)]} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
