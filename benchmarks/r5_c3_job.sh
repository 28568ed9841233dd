#!/bin/bash
# the rules op as ONE launch (k_pool_fused) against the two launches of rounds 1-4 (variant 7): parity of every variant + the T2
# sweep (10^6 positions through the fused kernel), the C3 microbenchmark on the three position sets, encoder-split and mask-tile sweeps
O=gpurun_out/${OUT:-r5c3}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rules.py tests/test_gpu_bench_shape.py::test_t2_one_million_positions_vs_the_oracle -m gpu -q -x -s --timeout=800 2>&1 | tail -4 | tee $O/pytest_rules.log
for rep in 1 2; do
  python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"form": "one launch", /' | tee -a $O/c3.jsonl | cut -c1-200
  python benchmarks/movegen_bench.py --launches 100 --variant 7 2>&1 | grep '^{' | sed 's/^{/{"form": "two launches", /' | tee -a $O/c3.jsonl | cut -c1-200
done
for sp in 30 40 50 60 70 100; do python benchmarks/movegen_bench.py --launches 100 --only S-mid,S-dense --enc-split $sp 2>&1 | grep '^{' | sed "s/^{/{\"enc_split\": $sp, /" | tee -a $O/c3_enc_split_sweep.jsonl | cut -c1-170; done
for v in 16 24 32; do python benchmarks/movegen_bench.py --launches 100 --only S-mid --variant $v 2>&1 | grep '^{' | sed "s/^{/{\"mask_tile\": $v, /" | tee -a $O/c3_mask_tile_sweep.jsonl | cut -c1-170; done
