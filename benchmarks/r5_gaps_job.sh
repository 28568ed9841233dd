#!/bin/bash
# kernel trace of a short bench run: the gaps between the kernels of a round (benchmarks/trace_round_gaps.py)
O=gpurun_out/${OUT:-r5gaps}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 $EXTRA > $R/$O/prof.json 2> $R/$O/prof.err
cd $R
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 benchmarks/trace_round_gaps.py "$t" 0.1 | tee $O/round_gaps${TAG}.json
rm -rf $O/prof
