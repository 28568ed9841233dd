set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1h
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','games_in_timed_region','mean_plies_per_game','mean_descent_depth')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['playouts_per_s'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 20 --no-cpu-baseline > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -size +8M -delete
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 4096 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 4096 > $O/pmc_write.log 2>&1
find $O -name '*kernel_trace.csv' -size +4M -delete
cd $R
python benchmarks/movegen_bench.py --boards 4096 > $O/movegen_b4096.jsonl 2>/dev/null; cut -c1-150 $O/movegen_b4096.jsonl
python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null; cut -c1-150 $O/movegen_c3.jsonl
