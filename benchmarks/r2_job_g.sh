# gpurun job G (round 2): counters of the trunk kernels
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
python benchmarks/conv_bench.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -- /usr/bin/python3 $R/benchmarks/conv_bench.py --iters 5 --what layered,fused > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc2 -- /usr/bin/python3 $R/benchmarks/conv_bench.py --iters 5 --what layered,fused > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc3 -- /usr/bin/python3 $R/benchmarks/conv_bench.py --iters 5 --what layered,fused > $O/pmc3.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    fs = glob.glob("$O/%s/*/*counter_collection.csv" % d)
    if not fs: print(d, "no counter file", glob.glob("$O/%s/*/*" % d)[:5]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_conv3x3" in k or "k_trunk" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v[-4:]) / len(v[-4:])) for c, v in cs.items()})
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -size +3M -delete
