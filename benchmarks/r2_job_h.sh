# gpurun job H (round 2): full GPU tier + bench + profile of the current code
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|Error|assert |engine-route|in-situ|evaluator vs|max \|err|step [0-9]:|median|position [0-9]" | head -60 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py --steps 8 > $O/bench_s8.json 2> $O/bench_s8.err; tail -2 $O/bench_s8.err
python - <<PY
import json
d=json.load(open('$O/bench_s8.json'))
print({k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','mean_descent_depth','ms_per_step_series','games_per_s_steady_state')})
print('  rules %.1f select %.1f expand %.1f maxdepth %s'%(d['roofline']['avg_launch_us'], d['roofline_tree'][0]['avg_launch_us'], d['roofline_tree'][1]['avg_launch_us'], d['engine_stats'].get('max_depth')))
print(d.get('roofline_nn')); print(d.get('clocks')); print(d.get('roofline_c3')); print(d.get('cpu_baseline'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 0 --desync-plies 700 --no-cpu-baseline --no-c3 > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
cat $O/prof_bench/*/*kernel_stats.csv | cut -c1-130 | head -12
