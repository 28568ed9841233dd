# gpurun job: C3 microbenchmark of the committed code + per-kernel durations + PMC traffic at 32,768 boards
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1c3
mkdir -p $O
cd $R
timeout 300 python benchmarks/movegen_bench.py > $O/movegen_c3.jsonl 2>/dev/null; cut -c1-140 $O/movegen_c3.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 30 > $O/kt_c3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/phase -- /usr/bin/python3 $R/benchmarks/pool_phase_times.py > $O/phase.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 --boards 32768 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b32768 -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 10 --boards 32768 > $O/pmc_sq.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f=glob.glob("$O/phase/*/*kernel_trace.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_pool" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][26:46], r.get("Grid_Size_X") or r.get("Grid_Size"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
with open("$O/phase_times.txt","w") as g:
    for k,v in sorted(agg.items()):
        v=v[-25:]
        line="%s grid_threads=%s calls=%d avg_us=%.1f"%(k[0],k[1],len(v),sum(v)/len(v)/1e3)
        print(line); g.write(line+"\n")
PY
find $O -name '*kernel_trace.csv' -size +2M -delete
