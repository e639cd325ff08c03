# object-centric density (a few hundred tiles with very long lists): per-tile (fused / kernel) against global depth order.  bash tools/ab_skew.sh
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 30 --warmup 6 --skew-frac 0.5"
for sc in 0.3 0.15 0.08; do for m in "tile fused" "tile kernel" "global fused"; do set -- $m
GSR_DEPTH_ORDER=$1 GSR_TILE_SORT=$2 $B --skew-scale $sc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('scale $sc', '$1/$2', d['value'], 'order', s['depth_order'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'mean', d['config']['gaussians_per_tile_mean'], 'max', d['config']['gaussians_per_tile_max'])"
done; done
