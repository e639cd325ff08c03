# Side points of §8(d) beside the headline workload, and the bench lines of the other variants:  bash tools/side_points.sh
# -> gpurun_out/side/{side_points.jsonl, bench_ewa.json, bench_plane.json, bench_ewa_sh.json}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/side; O=$GRAFT_REPO_ROOT/gpurun_out/side; cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-method-iteration"
$B --variant ewa 2>/dev/null | tail -1 > $O/bench_ewa.json
$B --variant plane 2>/dev/null | tail -1 > $O/bench_plane.json
$B --variant ewa --color-mode sh 2>/dev/null | tail -1 > $O/bench_ewa_sh.json
: > $O/side_points.jsonl
for P in 100000 1000000 3000000; do
  for v in surfel ewa plane; do
    $B --variant $v --P $P --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print(json.dumps({'variant':'$v','P':$P,'W':1920,'H':1080,'iters_per_s':d['value'],'R':d['config']['tile_instances_R'],'fwd_ms':round(s['preprocess']+s['depth_order']+s['binning']+s['blend_fwd'],4),'bwd_ms':round(s['bwd_memset']+s['blend_bwd']+s['preprocess_bwd'],4),'stage_ms':s}))" >> $O/side_points.jsonl
  done
done
cat $O/side_points.jsonl | cut -c1-160
