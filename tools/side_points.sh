# SURVEY §8d side points: P in {100k, 1M} at 1080p and 300k at 1600x900, all variants.  One JSON object per line.
cd /tmp
for cfg in "100000 1920 1080" "1000000 1920 1080" "300000 1600 900"; do set -- $cfg
  for v in surfel ewa plane; do
    python $GRAFT_REPO_ROOT/bench.py --variant $v --P $1 --W $2 --H $3 --steps 40 --warmup 5 --no-cpu-baseline --no-method-iteration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant':'$v','P':$1,'W':$2,'H':$3,'iters_per_s':d['value'],'R':d['config']['tile_instances_R'],'fwd_ms':d['rasterize_fwd_ms'],'bwd_ms':d['rasterize_bwd_ms'],'stage_ms':d['stage_ms']}))"
  done
done
