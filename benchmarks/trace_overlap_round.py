"""A round of the asynchronous loop WITH the opt-in second k_advance launch (qz_selfplay_set_overlap), from a rocprofv3 kernel
trace (csv) of `bench.py --overlap-us N`: for the last `fraction` of the trace, when each kernel of a round starts and ends
relative to the END of the round's first k_advance launch.  usage: trace_overlap_round.py <kernel_trace.csv> [fraction=0.1]"""
import collections
import csv
import json
import sys

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if "k_advance" in k:
            k = "k_advance second launch (beside the network)" if ", true>" in k else "k_advance first launch"
        elif "k_trunk" in k:
            k = "k_trunk persistent" if k[k.find("k_trunk"):].split("(")[0].rstrip(">").endswith("true") and k.count("true") >= 3 else "k_trunk"
        elif "k_" in k:
            k = k[k.find("k_"):].split("(")[0].split("<")[0]
        else:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
rel = collections.defaultdict(list)
dur = collections.defaultdict(list)
ref = None
rounds = 0
for s, e, k in rows:
    if s < t_lo:
        continue
    if k == "k_advance first launch":
        if ref is not None:
            rel["next first launch: start"].append((s - ref) / 1e3)
        ref = e
        rounds += 1
        dur[k].append((e - s) / 1e3)
        continue
    if ref is None or k in ("k_harvest_copy", "k_harvest_scan"):
        continue
    rel[k + ": start"].append((s - ref) / 1e3)
    rel[k + ": end"].append((e - ref) / 1e3)
    dur[k].append((e - s) / 1e3)


def st(v):
    v = sorted(v)
    return {"n": len(v), "mean_us": round(sum(v) / len(v), 1), "p10": round(v[len(v) // 10], 1), "p50": round(v[len(v) // 2], 1), "p90": round(v[(9 * len(v)) // 10], 1)}


print(json.dumps({"rounds": rounds, "relative_to_the_end_of_the_first_launch_us": {k: st(v) for k, v in sorted(rel.items()) if len(v) >= 4},
                  "durations_us": {k: st(v) for k, v in sorted(dur.items()) if len(v) >= 4}}, indent=1))
