#!/bin/bash
# rules op at 32,768 boards: where the encoder tiles sit in the two launches' grids (share beside the path groups x share of the
# second launch's tiles in FRONT of its mask groups), after the parity of every variant
O=gpurun_out/${OUT:-r5c3}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rules.py -m gpu -q -x --timeout=800 2>&1 | tail -3 | tee $O/pytest_rules.log
python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"form": "default", /' | tee -a $O/c3.jsonl | cut -c1-200
for sp in ${SPLITS:-30 40 50}; do for ef in ${FIRSTS:-0 100 200 300 400}; do
  python benchmarks/movegen_bench.py --launches 100 --only S-mid,S-dense --enc-split $sp --enc-first $ef 2>&1 | grep '^{' | sed "s/^{/{\"enc_split\": $sp, \"enc_first\": $ef, /" | tee -a $O/c3_enc_order_sweep.jsonl | cut -c1-190
done; done
