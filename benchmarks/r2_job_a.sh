# gpurun job A (round 2): GPU test tier on the paged engine + a short bench
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_gpu.log
cat $O/pytest_gpu.log | tail -40
timeout 900 python bench.py --steps 5 --no-c3 > $O/bench_s5.json 2> $O/bench_s5.err; tail -3 $O/bench_s5.err
python - <<PY
import json
d=json.load(open('$O/bench_s5.json'))
print({k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','games_in_timed_region','mean_descent_depth')}); print(d['roofline']); print(d['engine_stats'])
PY
