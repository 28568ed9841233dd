#!/bin/bash
# round-2 final job: everything profiles/round2 cites, measured on the committed code in ONE gpurun call
# (PMC passes are separate rocprofv3 runs with --kernel-trace only, program directly after `--`)
set -x
R=$PWD; O=$R/gpurun_out/r2final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py --steps 8 --clock-log $O/clock_log_bench_default.json > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 6 --groups 2 --no-cpu-baseline --no-c3 > $O/bench_groups2.json 2> $O/bench_groups2.err
timeout 600 python bench.py --steps 2 --library-trunk --no-cpu-baseline --no-c3 > $O/bench_library_trunk.json 2> $O/bench_library_trunk.err
# the other BASELINE configurations that fit one GPU: configs[1] (n_playout=100) and configs[4] per GPU (n_playout=800)
timeout 600 python bench.py --steps 12 --playouts 100 --no-cpu-baseline --no-c3 > $O/bench_c2_playouts100.json 2> $O/bench_c2_playouts100.err
timeout 600 python bench.py --steps 3 --playouts 800 --no-cpu-baseline --no-c3 > $O/bench_c5_playouts800_1gpu.json 2> $O/bench_c5_playouts800_1gpu.err
python - <<PY
import json
for f in ("bench_default", "bench_groups2", "bench_library_trunk", "bench_c2_playouts100", "bench_c5_playouts800_1gpu"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, {k: d[k] for k in ("value", "ms_per_step", "plies_per_s", "playouts_per_s")}, d["games_per_s_steady_state"]["value"] if d.get("games_per_s_steady_state") else None)
    print("   rules %.1f us frac %.3f | tree %s | nn %s" % (d["roofline"]["avg_launch_us"], d["roofline"]["frac"],
          [(t["kernel"], round(t["avg_launch_us"], 1)) for t in d["roofline_tree"]], (d.get("roofline_nn") or {}).get("avg_launch_us")))
PY
timeout 300 python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null; cut -c1-150 $O/movegen_c3.jsonl
timeout 300 python benchmarks/movegen_bench.py --boards 4096 > $O/movegen_b4096.jsonl 2>/dev/null; cut -c1-150 $O/movegen_b4096.jsonl
timeout 300 python benchmarks/conv_bench.py 2>&1 | grep -v amdgpu.ids > $O/conv_bench.txt; cat $O/conv_bench.txt
timeout 300 python benchmarks/trunk_stamps.py 2>&1 | grep -v amdgpu.ids > $O/trunk_stamps.txt; cat $O/trunk_stamps.txt
timeout 600 python benchmarks/select_stamps.py 3 2>&1 | grep -v amdgpu.ids > $O/select_stamps.txt; tail -4 $O/select_stamps.txt
timeout 300 python benchmarks/rules_stamps.py 3 2>&1 | grep -v -e amdgpu.ids -e Warning > $O/rules_stamps.txt; cat $O/rules_stamps.txt
timeout 300 python benchmarks/insitu_rules_timing.py 2>&1 | grep variant > $O/insitu_rules_variants.txt; cat $O/insitu_rules_variants.txt
for v in 3 5 6; do timeout 200 python benchmarks/movegen_bench.py --boards 4096 --variant $v 2>/dev/null | grep '^{' >> $O/movegen_b4096_variants.jsonl; done; cut -c1-150 $O/movegen_b4096_variants.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 0 --desync-plies 700 --no-cpu-baseline --no-c3 > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 4096 > $O/pmc_fetch_b4096.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 4096 > $O/pmc_write_b4096.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_fetch_b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_write_b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_trunk -- /usr/bin/python3 $R/benchmarks/conv_bench.py --iters 5 --what heads_staged > $O/pmc_sq_trunk.log 2>&1
find $O -name '*kernel_trace.csv' -delete
cd $R
python benchmarks/pmc_traffic.py --fetch $O/pmc_fetch_b4096 --write $O/pmc_write_b4096 --kernels k_wave_rules --boards 4096 --bytes-per-board 8468 \
  --label "k_wave_rules<16,1> (mask + planes, one launch: 1,024 mask workgroups + 256 encoder groups), position set S-mid" \
  --out $O/pmc_traffic.json --rows-out $O/pmc_rules_b4096_rows.csv | cut -c1-300
python benchmarks/pmc_traffic.py --fetch $O/pmc_fetch_b32768 --write $O/pmc_write_b32768 --kernels k_pool_paths_enc,k_pool_masks_enc --boards 32768 --bytes-per-board 8468 \
  --label "k_pool_paths_enc + k_pool_masks_enc (mask + planes), position set S-mid" \
  --out $O/pmc_traffic_c3.json --rows-out $O/pmc_rules_b32768_rows.csv | cut -c1-300
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_sq_trunk/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r["Kernel_Name"]
    if "k_trunk" in k or "k_head_fc" in k:
        agg[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/pmc_sq_trunk.txt", "w") as g:
    for k, cs in agg.items():
        line = "%s %s" % (k, {c: round(sum(v[-4:]) / len(v[-4:])) for c, v in cs.items()})
        print(line); g.write(line + "\n")
PY
find $O -name '*counter_collection.csv' -size +3M -delete
