#!/bin/bash
# What the tree kernels' issue slots go to: SQ counters by instruction class and by busy unit, in passes of eight counters
# (SELECT_OPTS 8 = k_advance alone, 40 = k_rows for the boards without walls).  One bench step per pass.
O=gpurun_out/${OUT:-r6sq2}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"
P3="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32"
P4="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
for so in ${SOS:-8 40}; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/$O/pmc_$so_$i -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --second-line-seconds 0 --select-opts $so > $R/$O/bench_${so}_$i.json 2> $R/$O/bench_${so}_$i.err
    c=$(find $R/$O/pmc_$so_$i -name "*counter_collection.csv" | head -1)
    python3 $R/benchmarks/pmc_tail_stats.py "$c" 0.15 > $R/$O/sq_select_opts_${so}_pass$i.json
    rm -rf $R/$O/pmc_$so_$i
  done
  python3 - <<PY
import json
tot={}
for i in (1,2,3,4):
    try:
        d=json.load(open("$R/$O/sq_select_opts_${so}_pass%d.json"%i))
    except Exception as e:
        print("pass",i,"failed",e); continue
    b=json.loads(open("$R/$O/bench_${so}_%d.json"%i).read().strip().splitlines()[-1])
    ppr=b["playouts_per_s"]*b["ms_per_round"]/1e3
    for k,v in d.items():
        if k.startswith(("k_rows<","k_advance","k_lanes")):
            n=v["dispatches"]
            t=tot.setdefault(k[:14],{})
            for c,x in v.items():
                if c!="dispatches": t[c]=x/n
            t["playouts_per_round_pass%d"%i]=ppr
print(json.dumps(tot,indent=1))
json.dump(tot,open("$R/$O/sq_by_class_select_opts_$so.json","w"),indent=1)
PY
done
