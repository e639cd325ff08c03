# per-tile (fused into k_blend_fwd) against global depth order around the auto threshold P = 96 T:  bash tools/ab_depth_order.sh
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 40 --warmup 8"
for P in 600000 800000 1000000 1500000; do for m in tile global; do
GSR_DEPTH_ORDER=$m $B --variant surfel --P $P 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('surfel', $P, '$m', d['value'], 'order', s['depth_order'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'sum', round(s['depth_order']+s['binning']+s['blend_fwd'],4), d['config']['gaussians_per_tile_mean'], d['config']['gaussians_per_tile_max'])"
done; done
