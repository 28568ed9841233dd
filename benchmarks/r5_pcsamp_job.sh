#!/bin/bash
# rocprofv3 PC sampling of the asynchronous loop in its late-game regime: where k_advance's issue slots go, by source line
# (library built with line tables: tests/hip/libqzero_hip_lines.so).  METHOD=host_trap|stochastic, INTERVAL in UNIT.
O=gpurun_out/${OUT:-r5pc}; mkdir -p $O
R=$GRAFT_REPO_ROOT
make -C tests/hip -s libqzero_hip_lines.so 2>&1 | grep -E "error"
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L 2>/dev/null | grep -i -A6 "pc sampl" | head -30 > $R/$O/pc_sampling_configs.txt
QZ_BENCH_LIB=$R/tests/hip/libqzero_hip_lines.so BOARDS=${BOARDS:-10240} PLAYOUTS=400 MAXP=4096 BUDGET=${BUDGET:-2400} FIX=0 MAXD=992 SKIP_ROUNDS=${WARM:-6400} ITERS=${ITERS:-6} ROUNDS=64 EVERY=3 \
  timeout ${TMO:-420} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${METHOD:-host_trap} --pc-sampling-unit ${UNIT:-time} --pc-sampling-interval ${INTERVAL:-1} \
  --kernel-trace --output-format csv -d $R/$O/prof -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/prof.log 2>&1
echo "rocprofv3 rc=$?"
tail -3 $R/$O/prof.log | cut -c1-300
cd $R
find $O/prof -type f | head -20
f=$(find $O/prof -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && { head -3 "$f"; wc -l "$f"; python3 benchmarks/pcsamp_summary.py "$f" > $O/pcsamp_summary.json; python3 - <<PY
import json; d=json.load(open("$O/pcsamp_summary.json")); print(d["samples"], d["kernels"]); [print(x) for x in d["top_lines"][:40]]
PY
}
rm -rf $O/prof
