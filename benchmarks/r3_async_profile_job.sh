#!/bin/bash
# asynchronous loop at the headline size: interval statistics, then a rocprofv3 kernel trace whose LAST 30 % (the
# late-game regime) is summarised per kernel
O=gpurun_out/${OUT:-r3d}; mkdir -p $O
BOARDS=4096 PLAYOUTS=400 MAXP=${MAXP:-64} BUDGET=${BUDGET:-0} FIX=0 ITERS=${ITERS:-500} ROUNDS=64 EVERY=25 GRAPH=${GRAPH:-0} timeout 400 python benchmarks/async_debug.py > $O/async_stats.log 2>&1
tail -4 $O/async_stats.log | cut -c1-1000
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BOARDS=4096 PLAYOUTS=400 MAXP=${MAXP:-64} BUDGET=${BUDGET:-0} FIX=0 ITERS=${PITERS:-300} ROUNDS=64 EVERY=100 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/prof.log 2>&1
cd $R
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_tail_stats.py "$t" 0.3 > $O/trace_tail_stats.json; cat $O/trace_tail_stats.json | head -60
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$s" $O/kernel_stats_whole_run.csv
rm -rf $O/prof
