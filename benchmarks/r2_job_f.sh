# gpurun job F (round 2): determinism hunt per kernel, fused trunk, bench
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_conv.py -q -s 2>&1 | grep -v "^$" | grep -E "Error|assert|passed|failed|FAILED|differs|max \|err|evaluator vs|k_" | head -40 > $O/pytest_det.log; cat $O/pytest_det.log
timeout 600 python bench.py --steps 6 --no-c3 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; tail -2 $O/bench_fused.err
python - <<PY
import json
d=json.load(open('$O/bench_fused.json'))
print({k:d[k] for k in ('ms_per_step','plies_per_s','mean_descent_depth','ms_per_step_series')})
print('  rules %.1f select %.1f expand %.1f maxdepth %s'%(d['roofline']['avg_launch_us'], d['roofline_tree'][0]['avg_launch_us'], d['roofline_tree'][1]['avg_launch_us'], d['engine_stats'].get('max_depth')))
print(d.get('roofline_nn')); print(d.get('clocks'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 2 --warmup 0 --desync-plies 700 --no-cpu-baseline --no-c3 > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
cat $O/prof_bench/*/*kernel_stats.csv | cut -c1-130 | head -14
