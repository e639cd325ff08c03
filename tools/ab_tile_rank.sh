# per-tile sort in the forward's prologue: rank inside depth buckets (default) against the all-pairs count / bitonic network (GSR_TILE_RANK=plain)
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
for a in "--P 300000" "--P 300000 --variant ewa" "--P 300000 --variant plane" "--P 100000" "--P 600000" "--P 1000000" "--P 1500000 --variant ewa"; do for m in buckets plain; do
GSR_TILE_RANK=$m $B $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('[$a]', '$m', d['value'], 'fwd', s['blend_fwd'], 'mean', d['config']['gaussians_per_tile_mean'], 'max', d['config']['gaussians_per_tile_max'])"
done; done
for m in buckets plain; do GSR_TILE_RANK=$m python $GRAFT_REPO_ROOT/tools/bench_pipeline.py --steps 200 --warmup 15 --loss full-hip --graph 2>&1 | tail -1 | grep -o "\"ms_per_iter.*"; GSR_TILE_RANK=$m python $GRAFT_REPO_ROOT/tools/bench_pipeline_octree_pgsr.py --steps 200 --warmup 15 --graph 2>&1 | tail -1 | grep -o "\"ms_per_iter.*"; done
