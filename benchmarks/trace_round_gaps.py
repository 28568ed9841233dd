"""Where the time BETWEEN the kernels of a round goes: for the last `fraction` of a rocprofv3 kernel trace (csv) of the asynchronous
loop, the gap in front of every kernel of the main chain (k_advance -> k_trunk -> k_head_fc -> k_round_tail -> k_advance ...) =
its start minus the end of the chain's previous kernel, and when the side kernels (k_moves, k_wave_rules) start and end relative to
k_advance's end.  usage: trace_round_gaps.py <kernel_trace.csv> [fraction=0.2]"""
import collections
import csv
import json
import sys

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        k = k[k.find("k_"):].split("(")[0].split("<")[0] if "k_" in k else k[:30]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
sel = [r for r in rows if r[0] >= t_lo]
chain = ("k_advance", "k_trunk", "k_head_fc", "k_round_tail")
gaps = collections.defaultdict(list)
side = collections.defaultdict(list)
prev = None
adv_end = None
for s, e, k in sel:
    if k in chain:
        if prev is not None:
            gaps["%s -> %s" % (prev[2], k)].append((s - prev[1]) / 1e3)
        prev = (s, e, k)
        if k == "k_advance":
            adv_end = e
    elif k in ("k_moves", "k_wave_rules") and adv_end is not None:
        side[k + " start after k_advance's end"].append((s - adv_end) / 1e3)
        side[k + " end after k_advance's end"].append((e - adv_end) / 1e3)


def st(v):
    v = sorted(v)
    return {"n": len(v), "mean_us": sum(v) / len(v), "p10": v[len(v) // 10], "p50": v[len(v) // 2], "p90": v[(9 * len(v)) // 10]}


print(json.dumps({"gaps_in_front_of_the_chain's_kernels": {k: st(v) for k, v in gaps.items() if len(v) > 5},
                  "side_stream": {k: st(v) for k, v in side.items() if len(v) > 5}}, indent=1))
