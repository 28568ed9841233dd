#!/bin/bash
# asynchronous loop at the headline size: s_memtime stamps inside k_advance, then a rocprofv3 kernel trace whose LAST 25 %
# (the late-game regime) is summarised per kernel
O=gpurun_out/${OUT:-r4p}; mkdir -p $O
R=$GRAFT_REPO_ROOT
BUDGET=${BUDGET:-1000} MAXP=4096 WARM_ROUNDS=${WARM:-9600} MEAS_ROUNDS=1280 timeout 300 python benchmarks/advance_stamps.py > $O/advance_stamps.json 2>/dev/null
python - <<PY
import json; d=json.load(open("$O/advance_stamps.json")); print({k:round(d[k],3) if isinstance(d[k],float) else d[k] for k in ("playouts_per_s","cycles_per_playout","cycles_per_launch_per_wave","rounds","wall_s")}); print({k:round(v) for k,v in d["cycles_per_playout_by_phase"].items()}); print(d["longest_launch_cycles_quantiles"], d["overrunning_launches"]); print({k: round(d[k], 3) for k in ("mean_depth", "levels_replayed_frac", "replay_rounds_per_playout", "replay_rounds_failed_frac", "levels_per_replay_round", "cycles_per_replay_round", "cycles_per_walked_level")}, {k: round(v) for k, v in d["descent_cycles_per_playout"].items()})
PY
cd /tmp && export TMPDIR=/tmp
BOARDS=4096 PLAYOUTS=400 MAXP=4096 BUDGET=${BUDGET:-1000} FIX=0 SKIP_ROUNDS=${WARM:-9600} ITERS=20 ROUNDS=64 EVERY=20 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/prof.log 2>&1
cd $R
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_tail_stats.py "$t" 0.1 > $O/trace_tail_stats.json; python - <<PY
import json; d=json.load(open("$O/trace_tail_stats.json")); print(round(d["window_ms"]), round(d["gpu_busy_frac"],3)); [print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()}) for k,v in list(d["kernels"].items())[:8]]
PY
rm -rf $O/prof
