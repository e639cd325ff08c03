# Runs on the GPU box: default bench (with cpu_baseline), rocprofv3 kernel stats of the same command, PMC traffic passes.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err; echo bench rc=$?; cat $R/gpurun_out/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --no-cpu-baseline --no-method-iteration > /dev/null 2>&1; echo stats rc=$?
for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-method-iteration > /dev/null 2>&1; echo $c rc=$?; done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_SQ -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-method-iteration > /dev/null 2>&1; echo SQ rc=$?
find $R/gpurun_out -name "*.csv" -newer $R/gpurun_out/bench_default.json | head -20
