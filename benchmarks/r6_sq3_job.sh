#!/bin/bash
# one pass of SQ counters over a bench step of the product build (k_advance alone): instructions per playout and unit busy shares
O=gpurun_out/${OUT:-r6sq3}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:--}; do
  L=""; tag=product; [ "$lib" != "-" ] && { L=$R/$lib; tag=$(basename $lib .so); }
  QZ_BENCH_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$O/pmc_$tag -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 ${EXTRA} > $R/$O/bench_$tag.json 2> $R/$O/bench_$tag.err
  c=$(find $R/$O/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.15 > $R/$O/sq_$tag.json
  rm -rf $R/$O/pmc_$tag
  python3 - <<PY
import json
d=json.load(open("$R/$O/sq_$tag.json")); b=json.loads(open("$R/$O/bench_$tag.json").read().strip().splitlines()[-1])
ppr=b["playouts_per_s"]*b["ms_per_round"]/1e3
for k,v in d.items():
    if k.startswith(("k_advance","k_rows<")):
        n=v["dispatches"]; cyc=v["SQ_BUSY_CYCLES"]/n/32
        print("$tag", k[:14], "playouts/round %.0f VALU/playout %.0f SALU/playout %.0f; launch %.2f Mcycles; VALU busy %.3f SALU busy %.3f; waves resident avg %.0f; per wave: issuing %.2f waiting-to-issue %.2f" % (
            ppr, v["SQ_INSTS_VALU"]/n/ppr, v["SQ_INSTS_SALU"]/n/ppr, cyc/1e6, v["SQ_ACTIVE_INST_VALU"]/n*4/(1024*cyc), v["SQ_ACTIVE_INST_SCA"]/n*4/(1024*cyc), v["SQ_WAVE_CYCLES"]/n*4/cyc, v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"]))
PY
done
