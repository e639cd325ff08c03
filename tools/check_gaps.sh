cd /tmp; export TMPDIR=/tmp
python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -m gpu -q -x -k "forward_backward_parity or speculative or edge" 2>&1 | tail -2
for i in 1 2 3; do python $GRAFT_REPO_ROOT/bench.py --steps 200 --no-cpu-baseline --no-method-iteration --no-graph-replay 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"stage_ms\"][\"blend_bwd\"], d[\"stage_ms\"][\"blend_fwd\"], d[\"roofline\"][\"avg_launch_ms\"])"; done
GSR_MAILBOX_POLL=0 python $GRAFT_REPO_ROOT/bench.py --steps 200 --no-cpu-baseline --no-method-iteration --no-graph-replay 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nopoll', d[\"value\"])"
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/ks | python -c "
import json,sys
t=json.load(sys.stdin); print(t['period_us'], t['kernel_us'], t['gap_us'], [(k['name'][:22], k['dur_us'], k['gap_before_us']) for k in t['kernels']])"
grep -h "k_blend_bwd_sp" /tmp/ks/*/*kernel_stats.csv | cut -c1-120
