#!/bin/bash
# same-box A/B of builds of the product library / flags through bench.py (STEPS steps, no CPU leg, no C3 line, no second lines).
# usage: VARLIBS="libqzero_hip_vXX.so ..." CASES="name:boards:lib[:flag,flag...] ..." bash benchmarks/r6_ab_job.sh
#   lib = path relative to the repo root, or - for the product build; variant libraries are built on the box (tests/hip/Makefile)
O=gpurun_out/${OUT:-r6ab}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error" ; [ -n "$VARLIBS" ] && make -C tests/hip -s $VARLIBS 2>&1 | grep -E "error"
[ -n "$TESTS" ] && { ulimit -c 0; timeout 900 python -m pytest $TESTS -m gpu -x -q --timeout=600 2>&1 | tail -3 | tee $O/pytest.log; }
for c in $CASES; do
  IFS=: read name boards lib extra <<< "$c"
  L=""; [ "$lib" != "-" ] && L=$PWD/$lib
  QZ_BENCH_LIB=$L timeout 400 python bench.py --steps ${STEPS:-6} --warmup 2 --boards $boards --no-cpu-baseline --no-c3 --second-line-seconds 0 $EXTRA ${extra//,/ } > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$name", {k:round(d[k],1) for k in ("plies_per_s","playouts_per_s","ms_per_step","memo_hit_rate")}, "ms/round", round(d["ms_per_round"],3), "advance us", round(r["avg_launch_us"],1), "frac", round(r["frac"],4), "nn us", round(d["roofline_nn"]["avg_launch_us"],1))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-800:])
PY
done
