set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1i
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 30 > $O/kt_c3.log 2>&1
for B in 32768; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b$B -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards $B > $O/pmc_fetch_b$B.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b$B -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards $B > $O/pmc_write_b$B.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 32768 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b4096 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 4096 > $O/pmc_sq4096.log 2>&1
find $O -name '*kernel_trace.csv' -size +4M -delete
cd $R
python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null; cut -c1-150 $O/movegen_c3.jsonl
grep -E "k_pool" $O/kt_c3/*/*kernel_stats.csv | cut -c1-70,200-300
