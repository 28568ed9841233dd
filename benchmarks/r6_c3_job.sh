#!/bin/bash
# rules op at 32,768 boards (BASELINE configs[2]): the pooled pipeline's two launches SIDE BY SIDE (two streams + ready flags: the
# default) against one after the other (--dependent), over the share of the encoder tiles beside the path groups; after the parity
# of every variant (tests/test_gpu_rules.py: incl. the two forms against each other, the oracle and the first kernel)
O=gpurun_out/${OUT:-r6c3}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rules.py -m gpu -q -x --timeout=800 2>&1 | tail -3 | tee $O/pytest_rules.log
for sp in ${SPLITS:-1 25 50 75 100}; do
  python benchmarks/movegen_bench.py --launches 100 --enc-split $sp 2>&1 | grep '^{' | tee -a $O/c3_two_streams_vs_dependent.jsonl | cut -c1-230
  python benchmarks/movegen_bench.py --launches 100 --enc-split $sp --dependent 2>&1 | grep '^{' | tee -a $O/c3_two_streams_vs_dependent.jsonl | cut -c1-230
done
[ -n "$T2" ] && timeout 600 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "one_million" --timeout=500 2>&1 | tail -3 | tee $O/pytest_t2.log
