#!/bin/bash
# SQ counters of the asynchronous loop's kernels in the late-game regime (last 30 % of the dispatches of a run from the opening)
O=gpurun_out/${OUT:-r3e}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BOARDS=4096 PLAYOUTS=400 MAXP=${MAXP:-64} BUDGET=${BUDGET:-0} FIX=0 ITERS=${PITERS:-180} ROUNDS=64 EVERY=60 timeout 800 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$O/pmc -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/pmc.log 2>&1
cd $R
tail -2 $O/pmc.log | cut -c1-600
c=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python benchmarks/pmc_tail_stats.py "$c" 0.3 > $O/pmc_tail_stats.json; cat $O/pmc_tail_stats.json
rm -rf $O/pmc
