#!/bin/bash
# which vector-memory-path counters this box offers, then k_advance<8> under them (late game, 8,192 boards): is the wavefronts' gather
# traffic (one cache line per lane) what saturates?  PMCS="A B C" = one rocprofv3 pass per group (comma separated inside a group; at most
# three counters of one hardware block per pass: five TA / TCP counters at once = "exceeds the capabilities of the hardware to collect")
O=gpurun_out/${OUT:-r4cnt}; mkdir -p $O
R=$GRAFT_REPO_ROOT
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $R/$O/counters_available.txt; wc -l $R/$O/counters_available.txt
k=0
for g in $PMCS; do
  k=$((k+1))
  BOARDS=${BOARDS:-8192} PLAYOUTS=400 MAXP=4096 BUDGET=1000 FIX=0 MAXD=992 ITERS=${ITERS:-60} ROUNDS=64 EVERY=50 SEL=${SEL:-0} timeout 300 rocprofv3 --kernel-trace --pmc ${g//,/ } --output-format csv -d $R/$O/pmc_$k -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/pmc_$k.log 2>&1
  c=$(find $R/$O/pmc_$k -name "*counter_collection.csv" | head -1)
  python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.3 > $R/$O/pmc_group$k.json; python3 - <<PY
import json
d=json.load(open("$R/$O/pmc_group$k.json"))
for kn,v in d.items():
    if "k_advance" in kn or "k_trunk" in kn: print(kn[:24], {a:(b if a=="dispatches" else "%.4g"%b) for a,b in v.items()})
PY
  rm -rf $R/$O/pmc_$k
done
