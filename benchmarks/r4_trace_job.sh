#!/bin/bash
# rocprofv3 kernel trace of the asynchronous loop at BOARDS boards (late-game regime): per-kernel statistics of its last part
O=gpurun_out/${OUT:-r4t}; mkdir -p $O
R=$GRAFT_REPO_ROOT
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
cd /tmp && export TMPDIR=/tmp
BOARDS=${BOARDS:-8192} PLAYOUTS=400 MAXP=4096 BUDGET=${BUDGET:-1000} FIX=0 MAXD=992 SKIP_ROUNDS=${WARM:-9600} ITERS=20 ROUNDS=64 EVERY=20 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/prof.log 2>&1
cd $R
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_tail_stats.py "$t" 0.1 > $O/trace_tail_stats_${BOARDS:-8192}.json; python - <<PY
import json; d=json.load(open("$O/trace_tail_stats_${BOARDS:-8192}.json")); print(round(d["window_ms"]), round(d["gpu_busy_frac"],3)); [print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()}) for k,v in list(d["kernels"].items())[:8]]
PY
tail -2 $O/prof.log | cut -c1-400
rm -rf $O/prof
