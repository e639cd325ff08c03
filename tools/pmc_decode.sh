cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm$i -- python $GRAFT_REPO_ROOT/tools/bench_decode.py > /dev/null 2>&1; echo "pass $i rc=$?"
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_dec' in k or 'k_wgrad' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k[-18:], {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
