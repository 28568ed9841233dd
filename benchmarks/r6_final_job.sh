#!/bin/bash
# Round-6 evidence for profiles/round6/: GPU test tier + smoke, the bench line (the driver's command), A/B lines, rocprofv3 kernel stats
# of the same command, PMC traffic of k_advance (separate FETCH_SIZE / WRITE_SIZE passes, program directly after `--`), a rocprofv3
# stats file of the rules op at 32,768 boards (C3), the 2-rank line.  PARTS="tests pmc bench ab prof c3 ranks" selects (the PMC passes run BEFORE the bench line, which cites their file).
O=gpurun_out/${OUT:-r6final}; mkdir -p $O
R=$GRAFT_REPO_ROOT
PARTS=${PARTS:-"tests pmc bench ab prof c3 ranks"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
ulimit -c 0
if has tests; then
  echo "tree: ${TREE_SHA:-unknown} ($(date -u +%FT%TZ)); command: python -m pytest tests -m gpu -x -q" > $O/pytest_gpu.log
  timeout 2400 python -m pytest tests -m gpu -x -q --timeout=900 2>&1 | tail -15 >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log
fi
( cd /tmp && export TMPDIR=/tmp
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/pmc_$c.json 2> $R/$O/pmc_$c.err
  done
  bpb=$(python3 -c "import json; d=json.load(open('$R/$O/pmc_FETCH_SIZE.json')); print(d['roofline']['algorithmic_bytes_per_launch'] / d['config']['boards_per_gpu'])")
  cd $R && python3 $R/benchmarks/pmc_traffic.py --fetch $R/$O/pmc_FETCH_SIZE --write $R/$O/pmc_WRITE_SIZE --kernels k_advance --boards 13312 --bytes-per-board $bpb --last 200 \
     --label "k_advance<7> (13,312 boards, one 3,000-us deadline per launch, n_playout=400, last 200 launches of a bench run)" --out $R/$O/pmc_traffic_advance.json > /dev/null && cat $R/$O/pmc_traffic_advance.json | head -14
  # (the bench line below cites this file -- and says whether it was measured on the kernels it runs: it must be in place first)
  mkdir -p $R/profiles/round6 && cp $R/$O/pmc_traffic_advance.json $R/profiles/round6/pmc_traffic_advance.json
  rm -rf $R/$O/pmc_FETCH_SIZE $R/$O/pmc_WRITE_SIZE
fi
)
cd $R
if has bench; then
  t0=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --clock-log $O/clock_log_bench_default.json > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err | cut -c1-300
  echo "bench wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
fi
line() { timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-c3 --second-line-seconds 0 "$@"; }
if has ab; then
  line --boards 4096 --budget-us 1000 --select-opts 0 > $O/bench_boards4096.json 2> $O/bench_boards4096.err
  line --playouts 800 > $O/bench_c5_playouts800_1gpu.json 2> $O/bench_c5.err
  line --playouts 100 > $O/bench_c2_playouts100.json 2> $O/bench_c2.err
  line --boards 20480 --select-opts 40 > $O/bench_k_rows_20480boards.json 2> $O/bench_rows.err
  line --boards 20480 > $O/bench_k_advance_20480boards.json 2> $O/bench_adv20k.err
  for f in $O/bench_*.json; do python3 - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f".split("/")[-1], round(d["plies_per_s"]), round(d["playouts_per_s"]/1e6,1), "M playouts/s", round(d.get("ms_per_round",0),3), "ms/round")
except Exception as e: print("$f", "FAILED", e)
PY
  done
fi
if has ranks; then
  QZ_DIST_BACKEND=gloo QZ_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --boards 6656 --no-cpu-baseline > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err; head -c 300 $O/bench_2ranks_one_gpu_gloo.json; echo
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 4 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err
  s=$(find $R/$O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$s" $R/$O/bench_kernel_stats_rocprofv3.csv; head -8 $R/$O/bench_kernel_stats_rocprofv3.csv | cut -c1-160
  t=$(find $R/$O/prof_bench -name "*kernel_trace.csv" | head -1); python3 $R/benchmarks/trace_tail_stats.py "$t" 0.15 > $R/$O/bench_kernel_trace_timed_region.json
  rm -rf $R/$O/prof_bench
fi
if has c3; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --launches 100 > $R/$O/c3_three_sets.jsonl 2> $R/$O/prof_c3.err
  s=$(find $R/$O/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$s" $R/$O/c3_kernel_stats_rocprofv3.csv; head -5 $R/$O/c3_kernel_stats_rocprofv3.csv | cut -c1-160
  rm -rf $R/$O/prof_c3
  grep '^{' $R/$O/c3_three_sets.jsonl | cut -c1-160
fi
