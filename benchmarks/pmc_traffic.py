"""rocprofv3 counter_collection CSVs (one FETCH_SIZE pass, one WRITE_SIZE pass) -> the pmc_traffic*.json files bench.py reads.

    python benchmarks/pmc_traffic.py --fetch <dir> --write <dir> --kernels k_wave_rules --boards 4096 --bytes-per-board 8468 \
        --label "..." --out profiles/round2/pmc_traffic.json [--rows-out profiles/round2/pmc_rules_b4096.csv]

Per launch: HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE, in units of 1,024 B (MI355X_MICROARCH.md, HBM / rocprofv3
section: on gfx950 FETCH_SIZE reports half the bytes of wide streaming reads -- an upper bound for the small gathers of these
kernels --, WRITE_SIZE is exact; Infinity-Cache hits are counted).  Kernels named in --kernels are summed per launch
(the pooled pipeline is two launches); only the last `--last` launches of every kernel count (the script's timed loop of
mask + planes calls; what precedes it is warm-up and mask-only calls)."""
import argparse, csv, glob, json, os, collections, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_support import tree_sha  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--kernels", required=True, help="comma-separated substrings of the kernel names to sum")
ap.add_argument("--boards", type=int, required=True)
ap.add_argument("--bytes-per-board", type=float, required=True)
ap.add_argument("--label", required=True)
ap.add_argument("--planes", type=int, default=1)
ap.add_argument("--last", type=int, default=20, help="use the last N launches of every kernel (the timed loop of the bench script; earlier ones are warm-up and mask-only calls)")
ap.add_argument("--out", required=True)
ap.add_argument("--rows-out")
a = ap.parse_args()
subs = a.kernels.split(",")


def per_launch(d, counter):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert fs, "no counter_collection.csv under " + d
    by = collections.defaultdict(list)
    rows = []
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        for sub in subs:
            if sub in r["Kernel_Name"]:
                by[sub].append(float(r["Counter_Value"]))
                rows.append((r["Kernel_Name"][:80], counter, r["Counter_Value"]))
    assert all(len(by[s]) >= a.last for s in subs), {s: len(by[s]) for s in subs}
    return sum(sum(by[s][-a.last:]) / a.last for s in subs), rows, {s: a.last for s in subs}


f_kb, rows_f, n_f = per_launch(a.fetch, "FETCH_SIZE")
w_kb, rows_w, n_w = per_launch(a.write, "WRITE_SIZE")
traffic = (2.0 * f_kb + w_kb) * 1024.0
alg = a.boards * a.bytes_per_board
out = {"boards": a.boards, "planes": bool(a.planes), "kernel": a.label, "fetch_size_kb": f_kb, "write_size_kb": w_kb,
       "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide streaming reads; an upper bound here: the reads are small gathers), "
                     "WRITE_SIZE as is; x1024 B (MI355X_MICROARCH.md, HBM section); summed over the launches of one call",
       "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio": traffic / alg,
       "launches_averaged": {"fetch": n_f, "write": n_w}, **tree_sha(),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), benchmarks/r6_final_job.sh; parsed by benchmarks/pmc_traffic.py"}
json.dump(out, open(a.out, "w"), indent=1)
if a.rows_out:
    with open(a.rows_out, "w") as g:
        g.write("kernel,counter,value\n")
        for r in rows_f + rows_w:
            g.write("\"%s\",%s,%s\n" % r)
print(json.dumps(out))
