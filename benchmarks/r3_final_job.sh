#!/bin/bash
# Round-3 evidence for profiles/round3/: GPU test tier, the bench line (default route + A/B lines), rocprofv3 kernel stats of
# the same command, PMC traffic of k_advance (separate FETCH_SIZE / WRITE_SIZE passes, program directly after `--`), and the
# s_memtime stamps inside k_advance.  PARTS="tests bench ab prof pmc stamps" selects.
O=gpurun_out/r3final; mkdir -p $O
R=$GRAFT_REPO_ROOT
PARTS=${PARTS:-"tests bench ab prof pmc stamps"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
if has bench; then
  timeout 600 python bench.py --clock-log $O/clock_log_bench_default.json > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err | cut -c1-300
fi
if has ab; then
  timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-c3 --groups 2 > $O/bench_groups2.json 2> $O/bench_groups2.err
  timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-c3 --no-memo > $O/bench_no_memo.json 2> $O/bench_no_memo.err
  timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-c3 --max-depth 0 > $O/bench_no_depth_limit.json 2> $O/bench_no_depth_limit.err
  timeout 400 python bench.py --mode lockstep --steps 3 --no-cpu-baseline --no-c3 > $O/bench_lockstep.json 2> $O/bench_lockstep.err
  timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-c3 --playouts 800 > $O/bench_c5_playouts800_1gpu.json 2> $O/bench_c5.err
  timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-c3 --playouts 100 > $O/bench_c2_playouts100.json 2> $O/bench_c2.err
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 4 --no-cpu-baseline --no-c3 > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err
  s=$(find $R/$O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$s" $R/$O/bench_kernel_stats_rocprofv3.csv; head -8 $R/$O/bench_kernel_stats_rocprofv3.csv | cut -c1-160
  t=$(find $R/$O/prof_bench -name "*kernel_trace.csv" | head -1); python3 $R/benchmarks/trace_tail_stats.py "$t" 0.15 > $R/$O/bench_kernel_trace_timed_region.json
  rm -rf $R/$O/prof_bench
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 > $R/$O/pmc_$c.json 2> $R/$O/pmc_$c.err
  done
  bpb=$(python3 -c "import json; d=json.load(open('$R/$O/pmc_FETCH_SIZE.json')); print(d['roofline']['algorithmic_bytes_per_launch'] / d['config']['boards_per_gpu'])")
  python3 $R/benchmarks/pmc_traffic.py --fetch $R/$O/pmc_FETCH_SIZE --write $R/$O/pmc_WRITE_SIZE --kernels k_advance --boards 4096 --bytes-per-board $bpb --last 200 \
     --label "k_advance (4,096 boards, n_playout=400, last 200 launches of a bench run)" --out $R/$O/pmc_traffic_advance.json > /dev/null && cat $R/$O/pmc_traffic_advance.json | head -12
  rm -rf $R/$O/pmc_FETCH_SIZE $R/$O/pmc_WRITE_SIZE
fi
cd $R
if has stamps; then
  BUDGET=1000 MAXP=4096 WARM_ROUNDS=9600 MEAS_ROUNDS=1280 timeout 300 python benchmarks/advance_stamps.py > $O/advance_stamps.json 2>/dev/null; head -c 600 $O/advance_stamps.json
fi
