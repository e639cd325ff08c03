# Runs on the GPU box (round 2 evidence): default bench (with cpu_baseline + parity_full_size + method_iteration), rocprofv3 kernel stats of the
# same command, and PMC passes (each in its own run, --kernel-trace only): HBM traffic (FETCH_SIZE, WRITE_SIZE), SQ issue / stall counters,
# LDS bank conflicts, L2 hit/miss and atomic counters (TCC_ATOMIC = atomics executed in L2, TCC_EA0_ATOMIC = forwarded to the memory side).
# Output under gpurun_out/prof_r02; tools/make_traffic.py r02 turns the PMC passes into profiles/r02_pmc_summary.json + profiles/traffic.json.
mkdir -p gpurun_out/prof_r02; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-method-iteration > /dev/null 2>&1; echo stats rc=$?
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-method-iteration"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
           "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r02_$i -- $B > /dev/null 2>&1; echo "pmc pass $i ($set) rc=$?"
done
# the pixel-parallel backward (GSR_BWD=px) for the A/B record, then the other two variants
for v in ewa plane; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r02_$v -- $B --variant $v > /dev/null 2>&1; echo "pmc $v rc=$?"
done
GSR_BWD=px timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r02_px1 -- $B > /dev/null 2>&1; echo "pmc px rc=$?"
GSR_BWD=px timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r02_px2 -- $B > /dev/null 2>&1; echo "pmc px2 rc=$?"
find $O/stats -name "*kernel_stats.csv" | head -2
