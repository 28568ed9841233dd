#!/bin/bash
# descent records per board: 16 (product) vs 32 / 64 (tests/hip/libqzero_hip_recs*.so), short bench on the same box
mkdir -p gpurun_out/recs
for lib in "" recs32 recs64; do
  if [ -n "$lib" ]; then export QZ_BENCH_LIB=$PWD/tests/hip/libqzero_hip_$lib.so; else unset QZ_BENCH_LIB; fi
  python bench.py --steps 6 --no-cpu-baseline --no-c3 > gpurun_out/recs/bench_${lib:-recs16}.json 2> gpurun_out/recs/bench_${lib:-recs16}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/recs/bench_${lib:-recs16}.json").read().strip().splitlines()[-1])
print("${lib:-recs16}", d["ms_per_step"], d["plies_per_s"], "rules", round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), [(t["kernel"], round(t["avg_launch_us"],1)) for t in d["roofline_tree"]], "nn", d["roofline_nn"]["avg_launch_us"])
print("   ", {k: d["engine_stats"][k] for k in ("max_edges","descent_levels","playouts") if k in d["engine_stats"]})
PY
done
