#!/bin/bash
# round-2 job O: full GPU tier + default bench on the current code
R=$PWD; O=$R/gpurun_out/r2o; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py --steps 8 > $O/bench_s8.json 2> $O/bench_s8.err; tail -2 $O/bench_s8.err
python - <<PY
import json
d=json.loads(open('$O/bench_s8.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','mean_descent_depth','ms_per_step_series','games_per_s_steady_state')})
print('  rules %.1f select %.1f expand %.1f'%(d['roofline']['avg_launch_us'], d['roofline_tree'][0]['avg_launch_us'], d['roofline_tree'][1]['avg_launch_us']))
print(d.get('roofline_nn')); print(d.get('clocks')); print(d.get('roofline_c3')); print(d.get('cpu_baseline')); print(d.get('engine_stats'))
PY
