# SURFEL backward: shorter accumulation table + five waves per SIMD, same box
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
run() { $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1', d['value'], 'bwd', s['blend_bwd'])"; }
run default
for f in "-DSP_CAP_SURFEL=80 -DSP_WPE_SURFEL=5" "-DSP_CAP_SURFEL=64 -DSP_WPE_SURFEL=5" "-DSP_CAP_SURFEL=96"; do
  touch gs-sr_amd/csrc/gsr_blend_sp.hip
  make -C gs-sr_amd/csrc BLEND_EXTRA="$f" > /tmp/mk.log 2>&1 || { echo "$f: build failed"; continue; }
  run "$f"
done
touch gs-sr_amd/csrc/gsr_blend_sp.hip; make -C gs-sr_amd/csrc > /dev/null 2>&1; run default_again
