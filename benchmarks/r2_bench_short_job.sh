#!/bin/bash
# short bench run (no CPU baseline, no C3 microbench): step time, engine telemetry, tree-kernel durations
mkdir -p gpurun_out/r2j
python bench.py --steps ${STEPS:-6} --no-cpu-baseline --no-c3 $EXTRA > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2j/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["plies_per_s"], d["engine_stats"])
for t in d["roofline_tree"]: print(t["kernel"], t["avg_launch_us"])
print(d["ms_per_step_series"])
PY
tail -3 gpurun_out/r2j/bench.err
