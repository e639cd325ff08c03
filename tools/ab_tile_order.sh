# launch order of the blend workgroups: raster (default), XCD bands (GSR_XCD_REMAP=1), longest list first always (GSR_TILE_ORDER=1) / never (=0) / auto (feedback)
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
for scene in "" "--skew-frac 0.25 --skew-scale 0.3" "--skew-frac 0.5 --skew-scale 0.3" "--skew-frac 0.5 --skew-scale 0.15" "--skew-frac 0.5 --skew-scale 0.08"; do for m in "0 auto" "1 0" "0 0" "0 1"; do set -- $m
GSR_XCD_REMAP=$1 GSR_TILE_ORDER=$2 $B $scene 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('[$scene]', 'xcd_remap=$1 tile_order=$2', d['value'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'max', d['config']['gaussians_per_tile_max'])"
done; done
