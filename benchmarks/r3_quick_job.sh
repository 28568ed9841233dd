#!/bin/bash
# quick check of kernel changes: the asynchronous-loop parity tests, the lock-step MCTS parity tests, then the loop's interval statistics
O=gpurun_out/${OUT:-r3q}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py tests/test_gpu_mcts.py -m gpu -x -q --timeout=300 2>&1 | tail -5 | tee $O/pytest.log
BOARDS=4096 PLAYOUTS=400 MAXP=${MAXP:-64} BUDGET=${BUDGET:-0} FIX=0 ITERS=${ITERS:-320} ROUNDS=${ROUNDS:-64} EVERY=${EVERY:-40} GRAPH=${GRAPH:-0} timeout 400 python benchmarks/async_debug.py > $O/async_stats.log 2>&1
tail -3 $O/async_stats.log | cut -c1-700
