#!/bin/bash
# round-2 job N: rocprofv3 kernel stats of the bench on the current code
R=$PWD; O=$R/gpurun_out/r2n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 0 --desync-plies 700 --no-cpu-baseline --no-c3 > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
f = glob.glob('$O/prof_bench/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-70s %7s %10.1f us %6s%%" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY
