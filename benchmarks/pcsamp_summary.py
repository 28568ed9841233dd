"""Aggregate a rocprofv3 PC-sampling CSV (host_trap or stochastic) by kernel, by source line (Instruction_Comment of a library built
with line tables) and by instruction class.  stdout: JSON."""
import collections
import csv
import json
import re
import sys


def klass(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main(path):
    rows = csv.DictReader(open(path, newline=""))
    cols = rows.fieldnames
    icol = next((c for c in cols if c.lower() == "instruction"), None)
    ccol = next((c for c in cols if "comment" in c.lower()), None)
    by_line, by_class, by_op, n = collections.Counter(), collections.Counter(), collections.Counter(), 0
    line_class = collections.defaultdict(collections.Counter)
    for r in rows:
        n += 1
        ins = (r.get(icol) or "").strip()
        op = ins.split()[0] if ins else "?"
        cm = (r.get(ccol) or "").strip()
        m = re.search(r"([A-Za-z0-9_./-]+\.(?:hip|h|hpp)):(\d+)", cm)
        key = "%s:%s" % (m.group(1).split("/")[-1], m.group(2)) if m else (cm[:60] or "?")
        by_line[key] += 1
        k = klass(op)
        by_class[k] += 1
        by_op[op] += 1
        line_class[key][k] += 1
    out = {"file": path, "columns": cols, "samples": n, "kernels": None,
           "by_class": {k: v / max(n, 1) for k, v in by_class.most_common()},
           "top_ops": [[k, v / max(n, 1)] for k, v in by_op.most_common(40)],
           "top_lines": [[k, round(v / max(n, 1), 5), dict(line_class[k])] for k, v in by_line.most_common(150)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
