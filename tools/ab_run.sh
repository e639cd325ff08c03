# A/B on one box: for each "name=libpath[:ENV=VAL]" argument prints it/s and blend stage times (3 alternating rounds).
cd /tmp
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}; lib=${rest%%:*}; envs=""
    if [ "$rest" != "$lib" ]; then envs=$(echo ${rest#*:} | tr ':' ' '); fi
    env $envs GSR_LIB_PATH=$GRAFT_REPO_ROOT/$lib python $GRAFT_REPO_ROOT/bench.py --variant ${VARIANT:-surfel} --steps 60 --warmup 10 --no-cpu-baseline --no-method-iteration 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$name', 'it/s', d['value'], 'fwd', round(s['blend_fwd'],4), 'bwd', round(s['blend_bwd'],4), 'pre', round(s['preprocess'],4), 'bin', round(s['binning'],4), 'pre_bwd', round(s['preprocess_bwd'],4))"
  done
done
