#!/bin/bash
# SQ counters of the tree kernel of a bench run (SELECT_OPTS 8 = k_advance alone, 40 = k_rows for the boards without walls): instructions per playout
O=gpurun_out/${OUT:-r6sq}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in ${SOS:-40 8}; do
  QZ_ROWS_WEU=${WEU:-4} timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$O/pmc_sq_$so -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 --select-opts $so > $R/$O/pmc_sq_$so.json 2> $R/$O/pmc_sq_$so.err
  c=$(find $R/$O/pmc_sq_$so -name "*counter_collection.csv" | head -1)
  python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.15 > $R/$O/pmc_sq_select_opts_$so.json
  rm -rf $R/$O/pmc_sq_$so
  python3 - <<PY
import json
d=json.load(open("$R/$O/pmc_sq_select_opts_$so.json")); b=json.loads(open("$R/$O/pmc_sq_$so.json").read().strip().splitlines()[-1])
ppr = b["playouts_per_s"]*b["ms_per_round"]/1e3
for k,v in d.items():
    if k.startswith("k_rows") or k.startswith("k_advance") or k.startswith("k_lanes"):
        n=v["dispatches"]
        print("$so", k[:30], "dispatches", n, "VALU/launch %.0fM SALU/launch %.0fM" % (v["SQ_INSTS_VALU"]/n/1e6, v["SQ_INSTS_SALU"]/n/1e6), "waves/launch", v["SQ_WAVES"]/n,
              "busy%% issuing %.2f waiting-to-issue %.2f waiting %.2f" % (v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"]))
print("$so playouts per round (bench line, under the profiler)", round(ppr), "playouts/s", round(b["playouts_per_s"]/1e6,1))
PY
done
