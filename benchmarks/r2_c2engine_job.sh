#!/bin/bash
# BASELINE configs[2] as an ENGINE run: 32,768 boards x n_playout=400 on one GPU; the leaf rules op in situ with the
# pooled pipeline (the library's choice at this size) and, as A/B, with k_wave_rules (variant 3); rocprofv3 kernel stats
R=$PWD; O=$R/gpurun_out/c2e; mkdir -p $O
timeout 1500 python bench.py --boards 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-c3 > $O/bench_c3_engine_b32768.json 2> $O/bench_c3_engine_b32768.err
timeout 1500 python bench.py --boards 32768 --steps 2 --warmup 0 --no-cpu-baseline --no-c3 --rules-variant 3 > $O/bench_c3_engine_b32768_wave_rules.json 2> $O/bench_c3_engine_b32768_wave_rules.err
python - <<PY
import json
for f in ("bench_c3_engine_b32768", "bench_c3_engine_b32768_wave_rules"):
    d=json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["plies_per_s"], d["playouts_per_s"], "rules", round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), [(t["kernel"], round(t["avg_launch_us"],1)) for t in d["roofline_tree"]], "nn", d["roofline_nn"]["avg_launch_us"], d["engine_stats"]["arena_bytes"], d["engine_stats"]["node_overflow"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- /usr/bin/python3 $R/bench.py --boards 32768 --steps 1 --warmup 0 --desync-plies 300 --no-cpu-baseline --no-c3 > $O/prof.log 2>&1
find $O/prof -name '*kernel_trace.csv' -delete
head -9 $(find $O/prof -name '*kernel_stats.csv' | head -1) | cut -c1-140
