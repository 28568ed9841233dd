set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1f
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json
python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null
python benchmarks/movegen_bench.py --boards 4096 > $O/movegen_b4096.jsonl 2>/dev/null
cat $O/movegen_c3.jsonl $O/movegen_b4096.jsonl | cut -c1-140
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 20 --no-cpu-baseline > $O/prof_bench.log 2>&1
find $O/prof_bench -name '*kernel_trace.csv' -size +8M -delete
for B in 4096 32768; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b$B -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards $B > $O/pmc_fetch_b$B.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b$B -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards $B > $O/pmc_write_b$B.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 32768 > $O/pmc_sq.log 2>&1
find $O -name '*kernel_trace.csv' -size +4M -delete
cd $R
python benchmarks/probe.py 2>&1 | grep -E "per_leaf float32  cl=0|eval     float32  cl=1|eval     bfloat16 cl=1|engine|select|net  |expand|full step|finish" > $O/probe.txt
cat $O/probe.txt
ls -R $O | head -50
