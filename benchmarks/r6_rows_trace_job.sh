#!/bin/bash
# rocprofv3 kernel trace of a bench run with k_rows: how long its launches last at a given number of wavefronts in the grid
O=gpurun_out/${OUT:-r6rowstrace}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  IFS=: read name boards waves <<< "$c"
  QZ_ROWS_WAVES=$waves timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$name -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --boards $boards --no-cpu-baseline --no-c3 --second-line-seconds 0 --select-opts 40 > $R/$O/tr_$name.json 2> $R/$O/tr_$name.err
  t=$(find $R/$O/tr_$name -name "*kernel_trace.csv" | head -1)
  python3 $R/benchmarks/trace_tail_stats.py "$t" 0.1 > $R/$O/trace_$name.json
  python3 - <<PY
import json
d=json.load(open("$R/$O/trace_$name.json"))
print("$name", {k:(round(v["avg_us"]), [round(x) for x in v["quantiles_us_10_50_90_99"]]) for k,v in d["kernels"].items() if k.startswith("k_rows") or k.startswith("k_advance")})
PY
  rm -rf $R/$O/tr_$name
done
