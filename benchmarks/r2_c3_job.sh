#!/bin/bash
# C3 (32,768 boards): A/B of a pending change of the pooled pipeline against the previous build of the library
# (tests/hip/libqzero_hip_prev.so, built from HEAD by hand), same box, plus the encoder-split sweep
mkdir -p gpurun_out/c3
python -m pytest tests/test_gpu_rules.py -m gpu -q -x -k "c3_size or variant" 2>&1 | tail -3 | tee gpurun_out/c3/pytest.log
for rep in 1 2; do
  QZ_BENCH_LIB=$PWD/tests/hip/libqzero_hip_prev.so python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"lib": "prev", /' | tee -a gpurun_out/c3/ab.jsonl
  python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"lib": "new", /' | tee -a gpurun_out/c3/ab.jsonl
done
for sp in 30 40 50 60 80; do
  python benchmarks/movegen_bench.py --launches 100 --only S-mid,S-dense --enc-split $sp 2>&1 | grep '^{' | sed "s/^{/{\"enc_split\": $sp, /" | tee -a gpurun_out/c3/split.jsonl
done
