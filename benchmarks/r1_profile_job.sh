set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1j
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s','games_in_timed_region','mean_plies_per_game','mean_descent_depth')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['playouts_per_s'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 20 --no-cpu-baseline > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -size +8M -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 30 > $O/kt_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 32768 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 4096 > $O/pmc_sq4096.log 2>&1
find $O -name '*kernel_trace.csv' -size +4M -delete
cd $R
python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null; cut -c1-150 $O/movegen_c3.jsonl
python benchmarks/movegen_bench.py --boards 4096 > $O/movegen_b4096.jsonl 2>/dev/null; cut -c1-150 $O/movegen_b4096.jsonl
grep -E "k_pool" $O/kt_c3/*/*kernel_stats.csv | cut -c1-70,200-300
python benchmarks/probe.py 2>&1 | grep -E "B= 4096|engine|select|net  |expand|full step" > $O/probe.txt; cat $O/probe.txt
python bench.py --bn eval --steps 5 --no-cpu-baseline --desync-plies 300 > $O/bench_eval_fp32_cl.json 2>/dev/null
python bench.py --bn eval --nn-dtype bf16 --steps 5 --no-cpu-baseline --desync-plies 300 > $O/bench_eval_bf16_cl.json 2>/dev/null
python -c "
import json
for f in ('bench_eval_fp32_cl','bench_eval_bf16_cl'):
    d=json.load(open('$O/'+f+'.json')); print(f, {k:d[k] for k in ('value','ms_per_step','plies_per_s','playouts_per_s')})"
