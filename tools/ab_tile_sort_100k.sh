cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 300 --warmup 20"
for r in 1 2; do for v in surfel ewa; do for m in fused kernel; do
GSR_TILE_SORT=$m $B --variant $v --P 100000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$v', 100000, '$m', d['value'], 'binning', s['binning'], 'fwd', s['blend_fwd'], 'sum', round(s['binning']+s['blend_fwd'],4))"
done; done; done
