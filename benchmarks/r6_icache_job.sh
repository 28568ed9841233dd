#!/bin/bash
# Instruction-cache and scalar-cache behaviour of the tree kernel over a bench step (one pass of eight counters)
O=gpurun_out/${OUT:-r6ic}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/bench.json 2> $R/$O/bench.err
c=$(find $R/$O/pmc -name "*counter_collection.csv" | head -1)
python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.15 > $R/$O/icache.json
rm -rf $R/$O/pmc
timeout 600 rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc2 -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 > $R/$O/bench2.json 2> $R/$O/bench2.err
c=$(find $R/$O/pmc2 -name "*counter_collection.csv" | head -1)
python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.15 > $R/$O/dcache.json
rm -rf $R/$O/pmc2
python3 - <<PY
import json
for f in ("icache","dcache"):
    d=json.load(open("$R/$O/%s.json"%f))
    for k,v in d.items():
        if k.startswith(("k_advance","k_trunk")):
            n=v["dispatches"]; print(f,k[:16],{c:round(x/n) for c,x in v.items() if c!="dispatches"})
PY
