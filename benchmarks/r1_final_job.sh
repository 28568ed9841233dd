# gpurun job: final round-1 numbers for the committed code
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1final
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --groups 2 --no-cpu-baseline > $O/bench_groups2.json 2> $O/bench_groups2.err
python -c "
import json
for f in ('bench_default','bench_groups2'):
    d=json.load(open('$O/'+f+'.json'))
    print(f,{k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','games_in_timed_region','mean_plies_per_game','mean_descent_depth')}); print(d['roofline'])
    if 'cpu_baseline' in d: print(d['cpu_baseline']['value'], d['cpu_baseline']['playouts_per_s'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 20 --no-cpu-baseline > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -delete
cd $R
timeout 300 python benchmarks/movegen_bench.py --boards 4096 > $O/movegen_b4096.jsonl 2>/dev/null; cut -c1-150 $O/movegen_b4096.jsonl
timeout 300 python benchmarks/probe.py 2>&1 | grep -E "B= 4096|B=32768|engine|select|net  |expand|full step" > $O/probe.txt; cat $O/probe.txt
timeout 300 python benchmarks/insitu_leaf_stats.py 2>&1 | tail -1 > $O/insitu_leaf_stats.txt; cat $O/insitu_leaf_stats.txt
bash benchmarks/r1_c3_job.sh 2>&1 | tail -12
