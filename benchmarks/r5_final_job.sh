#!/bin/bash
# Round-5 evidence for profiles/round5/: GPU test tier, the bench line (the driver's command) + A/B lines, rocprofv3 kernel stats of
# the same command, PMC traffic of k_advance<8> (separate FETCH_SIZE / WRITE_SIZE passes, program directly after `--`), SQ
# counters, the s_memtime stamps inside k_advance.  PARTS="tests bench ab prof pmc sq stamps" selects.
O=gpurun_out/${OUT:-r5final}; mkdir -p $O
R=$GRAFT_REPO_ROOT
PARTS=${PARTS:-"tests bench ab prof pmc sq stamps"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"; make -C tests/hip -s 2>&1 | grep -E "error"; make -C oracle -s 2>&1 | grep -E "error"
ulimit -c 0
if has tests; then
  echo "tree: ${TREE_SHA:-unknown} ($(date -u +%FT%TZ)); command: python -m pytest tests -m gpu -x -q" > $O/pytest_gpu.log
  timeout 1800 python -m pytest tests -m gpu -x -q --timeout=900 2>&1 | tail -15 >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log
fi
if has bench; then
  t0=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --clock-log $O/clock_log_bench_default.json > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err | cut -c1-300
  echo "bench wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
fi
line() { timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-c3 --second-line-seconds 0 "$@"; }
if has ab; then
  line --boards 4096 --budget-us 1000 --select-opts 0 > $O/bench_boards4096.json 2> $O/bench_boards4096.err
  line --boards 10240 --budget-us 2400 --select-opts 0 > $O/bench_round4_shape_10240boards_2400us.json 2> $O/bench_round4_shape.err
  line --playouts 800 > $O/bench_c5_playouts800_1gpu.json 2> $O/bench_c5.err
  line --playouts 100 > $O/bench_c2_playouts100.json 2> $O/bench_c2.err
  line --graph-rounds 16 > $O/bench_graph_rounds16.json 2> $O/bench_graph16.err
  for f in $O/bench_*.json; do python3 - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f".split("/")[-1], round(d["plies_per_s"]), round(d["playouts_per_s"]/1e6,1), "M playouts/s", round(d.get("ms_per_round",0),3), "ms/round")
except Exception as e: print("$f", "FAILED", e)
PY
  done
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 4 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err
  s=$(find $R/$O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$s" $R/$O/bench_kernel_stats_rocprofv3.csv; head -8 $R/$O/bench_kernel_stats_rocprofv3.csv | cut -c1-160
  t=$(find $R/$O/prof_bench -name "*kernel_trace.csv" | head -1); python3 $R/benchmarks/trace_tail_stats.py "$t" 0.15 > $R/$O/bench_kernel_trace_timed_region.json
  rm -rf $R/$O/prof_bench
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/pmc_$c.json 2> $R/$O/pmc_$c.err
  done
  bpb=$(python3 -c "import json; d=json.load(open('$R/$O/pmc_FETCH_SIZE.json')); print(d['roofline']['algorithmic_bytes_per_launch'] / d['config']['boards_per_gpu'])")
  python3 $R/benchmarks/pmc_traffic.py --fetch $R/$O/pmc_FETCH_SIZE --write $R/$O/pmc_WRITE_SIZE --kernels k_advance --boards 13312 --bytes-per-board $bpb --last 200 \
     --label "k_advance<8> (13,312 boards, one 3,000-us deadline per launch, n_playout=400, last 200 launches of a bench run)" --out $R/$O/pmc_traffic_advance.json > /dev/null && cat $R/$O/pmc_traffic_advance.json | head -12
  rm -rf $R/$O/pmc_FETCH_SIZE $R/$O/pmc_WRITE_SIZE
fi
if has sq; then
  BOARDS=13312 SELECT_OPTS=8 PLAYOUTS=400 MAXP=4096 BUDGET=3000 FIX=0 MAXD=992 ITERS=150 ROUNDS=64 EVERY=50 timeout 800 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$O/pmc_sq -- /usr/bin/python3 $R/benchmarks/async_debug.py > $R/$O/pmc_sq.log 2>&1
  c=$(find $R/$O/pmc_sq -name "*counter_collection.csv" | head -1)
  python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.3 > $R/$O/pmc_sq_async_13312boards.json; head -c 1500 $R/$O/pmc_sq_async_13312boards.json
  rm -rf $R/$O/pmc_sq
fi
cd $R
if has stamps; then
  BUDGET=1000 MAXP=4096 WARM_ROUNDS=9600 MEAS_ROUNDS=1280 timeout 300 python benchmarks/advance_stamps.py > $O/advance_stamps.json 2>/dev/null; head -c 600 $O/advance_stamps.json
fi
