set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "8 4096" "16 32768" "8 32768"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_v$1_b$2 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 30 --variant $1 --boards $2 > $O/kt_v$1_b$2.log 2>&1
  grep -E "k_pool|k_movegen" $O/kt_v$1_b$2/*/*kernel_stats.csv | cut -c1-300
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_v16_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --variant 16 --boards 32768 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAVES --output-format csv -d $O/pmc_sq2_v16_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --variant 16 --boards 32768 > $O/pmc_sq2.log 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
find $O -name "*counter_collection.csv" | head
