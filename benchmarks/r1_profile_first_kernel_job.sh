set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1
python bench.py > $R/gpurun_out/r1/bench_default.json 2> $R/gpurun_out/r1/bench_default.err
tail -c 3000 $R/gpurun_out/r1/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1/prof_bench -- /usr/bin/python3 $R/bench.py --steps 1 --warmup 0 --desync-plies 20 --no-cpu-baseline > $R/gpurun_out/r1/prof_bench.log 2>&1
find $R/gpurun_out/r1/prof_bench -name "*kernel_trace.csv" -size +8M -delete
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1/pmc_fetch -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 > $R/gpurun_out/r1/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1/pmc_write -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 > $R/gpurun_out/r1/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r1/pmc_sq -- /usr/bin/python3 $R/benchmarks/movegen_bench.py --only S-mid --launches 20 > $R/gpurun_out/r1/pmc_sq.log 2>&1
cd $R
python benchmarks/movegen_bench.py > $R/gpurun_out/r1/movegen_c3.jsonl 2>/dev/null
python benchmarks/game_length.py --boards 512 --playouts 400 --plies 1500 --seconds 540 > $R/gpurun_out/r1/game_length_400.json 2>/dev/null
cat $R/gpurun_out/r1/game_length_400.json
ls -laR $R/gpurun_out/r1 | head -60
