# gpurun job E (round 2): determinism hunt + speculative replay in k_select
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_determinism.py -q -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_det.log; cat $O/pytest_det.log
timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_train.py tests/test_gpu_api.py -q 2>&1 | tail -12 > $O/pytest_b.log; cat $O/pytest_b.log
for so in 0 1; do
  timeout 600 python bench.py --steps 6 --no-c3 --no-cpu-baseline --select-opts $so > $O/bench_so$so.json 2> $O/bench_so$so.err
done
python - <<PY
import json
for so in (0,1):
    try:
        d=json.load(open('$O/bench_so%d.json'%so))
    except Exception as e:
        print(so,'FAILED',e); continue
    print('select_opts',so, {k:d[k] for k in ('ms_per_step','plies_per_s','mean_descent_depth','ms_per_step_series')})
    print('  rules %.1f select %.1f expand %.1f conv %.1f maxdepth %s'%(d['roofline']['avg_launch_us'], d['roofline_tree'][0]['avg_launch_us'], d['roofline_tree'][1]['avg_launch_us'], (d.get('roofline_nn') or {}).get('avg_launch_us',0), d['engine_stats'].get('max_depth')))
PY
