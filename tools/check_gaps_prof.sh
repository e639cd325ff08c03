cd /tmp; export TMPDIR=/tmp
for prof in 0 1; do
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/tools/step_series.py --steps 100 --warmup 30 --prof $prof > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/ks | python -c "
import json,sys
t=json.load(sys.stdin); print('prof=$prof', t['period_us'], t['kernel_us'], t['gap_us'], [(k['name'][:16], k['gap_before_us']) for k in t['kernels'] if k['gap_before_us']>0.5])"
done
