#!/bin/bash
# head stage of k_trunk: which workgroups give the two-tile share to wave 1 (QZ_HEAD_ROLE_SHIFT; 31 = none)
mkdir -p gpurun_out/head
for rep in 1 2; do
for sh in 31 0 1 2 3 5 8; do
  echo "role_shift $sh: $(QZ_HEAD_ROLE_SHIFT=$sh python benchmarks/conv_bench.py --what heads_staged --iters 30 2>&1 | grep 'head stage')" | tee -a gpurun_out/head/role_shift.txt
done
done
QZ_HEAD_ROLE_SHIFT=0 python -m pytest tests/test_gpu_conv.py tests/test_gpu_api.py -m gpu -q -x -k "head or fixture or evaluat or trunk" 2>&1 | tail -3 | tee gpurun_out/head/pytest.log
