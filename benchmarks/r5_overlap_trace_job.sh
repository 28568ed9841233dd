#!/bin/bash
# kernel trace of a short bench run with the opt-in second k_advance launch: where the pieces of a round lie (benchmarks/trace_overlap_round.py)
O=gpurun_out/${OUT:-r5ovtrace}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ov in ${OVERLAPS:-1000}; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 --overlap-us $ov $EXTRA > $R/$O/prof_$ov.json 2> $R/$O/prof_$ov.err
  t=$(find $R/$O/prof -name "*kernel_trace.csv" | head -1)
  python3 $R/benchmarks/trace_overlap_round.py "$t" 0.1 | tee $R/$O/overlap_round_${ov}us${TAG}.json | head -60
  rm -rf $R/$O/prof
done
