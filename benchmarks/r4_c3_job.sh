#!/bin/bash
# the rules op after the hand-off record went from 1.5 KB to 184 B per board: parity of every kernel variant, the C3
# microbenchmark (32,768 boards, three position sets, mask-tile sweep), rocprofv3 kernel stats and the HBM traffic (PMC passes)
O=gpurun_out/${OUT:-r4c3}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rules.py -m gpu -q -x --timeout=500 2>&1 | tail -3 | tee $O/pytest_rules.log
for rep in 1 2; do python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | tee -a $O/c3.jsonl | cut -c1-260; done
for v in 16 24 32; do python benchmarks/movegen_bench.py --launches 100 --only S-mid --variant $v 2>&1 | grep '^{' | sed "s/^{/{\"mask_tile\": $v, /" | tee -a $O/c3_mask_tile_sweep.jsonl | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 50 --boards 32768 > $R/$O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $R/$O/pmc_write.log 2>&1
cd $R
python benchmarks/pmc_traffic.py --fetch $O/pmc_fetch --write $O/pmc_write --kernels k_pool_paths_enc,k_pool_masks_enc --boards 32768 --bytes-per-board 8468 \
  --label "k_pool_paths_enc + k_pool_masks_enc, 32,768 boards (S-mid), masks + planes, 184-byte hand-off record" \
  --out $O/pmc_traffic_c3.json --rows-out $O/pmc_rules_b32768_rows.csv | cut -c1-400
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3_kernel_stats_rocprofv3.csv 2>/dev/null; head -5 $O/c3_kernel_stats_rocprofv3.csv | cut -c1-200
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
