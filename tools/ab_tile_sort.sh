cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 60 --warmup 10"
for P in 100000 500000 750000; do for v in surfel ewa; do for m in fused kernel; do
GSR_TILE_SORT=$m $B --variant $v --P $P 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$v', $P, '$m', d['value'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'sum', round(s['binning']+s['blend_fwd'],4), d['config']['depth_order'][:8], d['config']['gaussians_per_tile_mean'], d['config']['gaussians_per_tile_max'])"
done; done; done
for v in ewa plane; do for m in fused kernel; do
GSR_TILE_SORT=$m $B --variant $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$v', 300000, '$m', d['value'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'sum', round(s['binning']+s['blend_fwd'],4))"
done; done
