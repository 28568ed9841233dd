#!/bin/bash
# round-2 job P: fork / join form of the pooled pipeline: parity, then timings (C3 and 4,096 boards)
mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r2p/pytest.log
timeout 300 python benchmarks/movegen_bench.py 2>/dev/null | cut -c1-230 | tee gpurun_out/r2p/movegen_c3.jsonl
for v in 24 16; do echo "variant $v"; timeout 300 python benchmarks/movegen_bench.py --only S-mid --variant $v 2>/dev/null | cut -c1-160; done | tee gpurun_out/r2p/variants.log
timeout 300 python benchmarks/movegen_bench.py --boards 8192 2>/dev/null | cut -c1-160 | tee -a gpurun_out/r2p/variants.log
timeout 300 python benchmarks/movegen_bench.py --boards 16384 --only S-mid 2>/dev/null | cut -c1-160 | tee -a gpurun_out/r2p/variants.log
