# Runs on the GPU box: SQ counter passes for the blend backward, both formulations (GSR_BWD=px|sp).  Output: gpurun_out/prof_bwd/<mode>_<set>/...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_bwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
VARIANT=${VARIANT:-surfel}
SETA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
SETB="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
for m in ${MODES:-px sp}; do
  i=0
  for set in "$SETA" "$SETB"; do
    i=$((i+1))
    GSR_BWD=$m timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${m}_$i -- python $R/bench.py --variant $VARIANT --steps 6 --warmup 2 --no-cpu-baseline --no-method-iteration > $O/${m}_$i.log 2>&1; echo $m set$i rc=$?
  done
done
python $R/tools/prof_bwd_summary.py
