# gpurun job D (round 2): conv kernel v1.1 + k_select A/B + failed tests re-run
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_train.py -x -q -s 2>&1 | grep -v "^$" | tail -30 > $O/pytest_a.log; cat $O/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_mcts.py -x -q 2>&1 | tail -8 > $O/pytest_b.log; cat $O/pytest_b.log
for so in 0 1 2 3; do
  timeout 600 python bench.py --steps 6 --no-c3 --no-cpu-baseline --select-opts $so > $O/bench_so$so.json 2> $O/bench_so$so.err
done
python - <<PY
import json
for so in (0,1,2,3):
    try:
        d=json.load(open('$O/bench_so%d.json'%so))
    except Exception as e:
        print(so,'FAILED',e); continue
    print('select_opts',so, {k:d[k] for k in ('ms_per_step','plies_per_s','mean_descent_depth','ms_per_step_series')})
    print('  rules %.1f select %.1f expand %.1f conv %.1f maxdepth %s'%(d['roofline']['avg_launch_us'], d['roofline_tree'][0]['avg_launch_us'], d['roofline_tree'][1]['avg_launch_us'], (d.get('roofline_nn') or {}).get('avg_launch_us',0), d['engine_stats'].get('max_depth')))
    print('  clocks', d.get('clocks'))
PY
