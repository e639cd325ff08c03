# blockIdx -> tile mapping: raster (0) / one band per XCD (1) / 4x4-tile blocks dealt out cyclically (2): it/s, blend times, HBM fetch of the blend kernels
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay"
for scene in "" "--skew-frac 0.25 --skew-scale 0.3" "--skew-frac 0.5 --skew-scale 0.15"; do for m in 0 1 2; do
GSR_XCD_REMAP=$m $B --steps 100 --warmup 10 $scene 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('[$scene]', 'xcd_remap=$m', d['value'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'])"
done; done
for v in surfel ewa plane; do for m in 0 1 2; do
rm -rf /tmp/pmc; GSR_XCD_REMAP=$m rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc -- $B --variant $v --steps 8 --warmup 2 --stage-steps 1 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE' and 'blend' in r['Kernel_Name']: acc[r['Kernel_Name'][:28]].append(float(r['Counter_Value']))
print('$v xcd_remap=$m', {k: round(2*sum(v)/len(v)*1024/1e6,1) for k,v in acc.items()}, 'MB fetched per launch (2 x FETCH_SIZE KB)')
PY
done; done
