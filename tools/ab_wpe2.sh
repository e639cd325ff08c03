# PLANE / EWA backward with a shorter accumulation table (fewer LDS bytes -> one more workgroup per CU) and the matching register budget, same box
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
run() { $B --variant $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1', '$2', d['value'], 'bwd', s['blend_bwd'])"; }
run default plane; run default ewa
for f in "-DSP_CAP_PLANE=96 -DSP_WPE_PLANE=5" "-DSP_CAP_PLANE=88 -DSP_WPE_PLANE=5" "-DSP_CAP_EWA=96 -DSP_WPE_EWA=7" "-DSP_CAP_EWA=80 -DSP_WPE_EWA=8"; do
  touch gs-sr_amd/csrc/gsr_blend_sp.hip
  make -C gs-sr_amd/csrc BLEND_EXTRA="$f" > /tmp/mk.log 2>&1 || { echo "$f: build failed"; continue; }
  case "$f" in *PLANE*) run "$f" plane;; *) run "$f" ewa;; esac
done
touch gs-sr_amd/csrc/gsr_blend_sp.hip; make -C gs-sr_amd/csrc > /dev/null 2>&1; run default_again ewa
