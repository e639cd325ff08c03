# A/B two builds of the library on the same box: GSR_LIB_PATH selects the .so.  Prints blend kernel stage times for 3 alternating runs.
cd /tmp
for i in 1 2 3; do
  for lib in libgsrast_old.so libgsrast_hip.so; do
    for v in ${VARIANTS:-surfel}; do
      GSR_LIB_PATH=$GRAFT_REPO_ROOT/gs-sr_amd/$lib python $GRAFT_REPO_ROOT/bench.py --variant $v --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$lib', '$v', 'it/s', d['value'], 'fwd', round(s['blend_fwd'],4), 'bwd', round(s['blend_bwd'],4))"
    done
  done
done
