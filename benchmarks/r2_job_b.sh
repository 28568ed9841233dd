# gpurun job B (round 2): the new bench line (clocks, roofline_tree, all-core CPU baseline) + the long game-length run
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
ls /sys/class/drm/ > $O/sysfs_probe.txt 2>&1; ls /sys/class/drm/card*/device/ >> $O/sysfs_probe.txt 2>&1; rocm-smi --showclocks --showpower --showtemp --json >> $O/sysfs_probe.txt 2>&1
timeout 900 python bench.py --clock-log $O/clock_log.json > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
for k in ('value','ms_per_step','plies_per_s','playouts_per_s','games_in_timed_region','mean_descent_depth','games_per_s_steady_state','game_lengths_seen','ms_per_step_series','clocks','engine_stats'): print(k, d.get(k))
print(d['roofline']); print(d['roofline_tree']); print(d.get('roofline_c3')); print(d.get('cpu_baseline'))
PY
timeout 2400 python benchmarks/game_length.py --boards 512 --playouts 400 --seconds 2100 --out $O/game_length_400playouts.json > $O/game_length.log 2> $O/game_length.err
tail -c 1500 $O/game_length.log; tail -3 $O/game_length.err
