#!/bin/bash
# one iteration of kernel work: parity tests of the tree kernels, stamps inside k_advance, A/B bench lines
O=gpurun_out/${OUT:-r4i}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error" ; make -C tests/hip -s libqzero_hip_astamps.so $VARLIBS 2>&1 | grep -E "error"
ulimit -c 0; timeout 900 python -m pytest tests/test_gpu_async_oracle.py tests/test_gpu_async.py tests/test_gpu_mcts.py -m gpu -x -q --timeout=600 2>&1 | tail -4 | tee $O/pytest.log
BUDGET=${BUDGET:-1000} MAXP=4096 WARM_ROUNDS=${WARM:-9600} MEAS_ROUNDS=1280 timeout 300 python benchmarks/advance_stamps.py > $O/advance_stamps.json 2>/dev/null
python - <<PY
import json; d=json.load(open("$O/advance_stamps.json")); print({k:round(d[k],3) if isinstance(d[k],float) else d[k] for k in ("playouts_per_s","cycles_per_playout","rounds","wall_s")}); print({k:round(v) for k,v in d["cycles_per_playout_by_phase"].items()}); print({k: round(d[k], 3) for k in ("mean_depth", "levels_replayed_frac", "replay_rounds_per_playout", "replay_rounds_failed_frac", "levels_per_replay_round", "cycles_per_replay_round", "cycles_per_walked_level")}, {k: round(v) for k, v in d["descent_cycles_per_playout"].items()})
PY
bash benchmarks/r4_ab_job.sh
