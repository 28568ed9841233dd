#!/bin/bash
# HBM-side traffic of the leaf rules op IN SITU on the 32,768-board engine (BASELINE configs[2]): separate FETCH_SIZE / WRITE_SIZE
# passes over a short bench run (80 desynchronisation plies, then one 20-playout ply whose launches are the ones counted)
R=$PWD; O=$R/gpurun_out/c2e_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- /usr/bin/python3 $R/bench.py --boards 32768 --steps 1 --warmup 0 --playouts 20 --desync-plies 80 --no-cpu-baseline --no-c3 > $O/$c.log 2>&1
  find $O/$c -name '*kernel_trace.csv' -delete
done
cd $R
python benchmarks/pmc_traffic.py --fetch $O/FETCH_SIZE --write $O/WRITE_SIZE --kernels k_pool_paths_enc,k_pool_masks_enc --boards 32768 --bytes-per-board 8468 \
  --label "k_pool_paths_enc + k_pool_masks_enc (mask + planes) on the leaf batches of a 32,768-board engine run (the last 20 launches: the timed 20-playout ply after 80 desynchronisation plies)" \
  --out $O/pmc_traffic_c3_engine.json --rows-out $O/pmc_rules_b32768_engine_rows.csv | cut -c1-400
find $O -name '*counter_collection.csv' -size +3M -delete
tail -3 $O/WRITE_SIZE.log
