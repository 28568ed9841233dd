# gpurun job C (round 2): MFMA trunk layer + k_select v2 + training / replay / rollout tiers, then A/B bench
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -s 2>&1 | tail -25 > $O/pytest_conv.log; cat $O/pytest_conv.log
timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_train.py -x -q -s 2>&1 | tail -40 > $O/pytest_b.log; cat $O/pytest_b.log
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_rules.py -x -q 2>&1 | tail -15 > $O/pytest_c.log; cat $O/pytest_c.log
timeout 600 python bench.py --steps 5 --no-c3 --no-cpu-baseline > $O/bench_mfma.json 2> $O/bench_mfma.err; tail -3 $O/bench_mfma.err
timeout 600 python bench.py --steps 5 --no-c3 --no-cpu-baseline --library-trunk > $O/bench_lib.json 2> $O/bench_lib.err; tail -3 $O/bench_lib.err
python - <<PY
import json
for f in ('bench_mfma','bench_lib'):
    try:
        d=json.load(open('$O/'+f+'.json'))
    except Exception as e:
        print(f, 'FAILED', e); continue
    print(f, {k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','mean_descent_depth','ms_per_step_series')})
    print('  rules', d['roofline']['avg_launch_us'], 'select', d['roofline_tree'][0]['avg_launch_us'], 'expand', d['roofline_tree'][1]['avg_launch_us'], 'nn', d.get('roofline_nn'))
    print('  clocks', d.get('clocks'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 300 --no-cpu-baseline --no-c3 > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
cat $O/prof_bench/*/*kernel_stats.csv | cut -c1-160 | head -30
