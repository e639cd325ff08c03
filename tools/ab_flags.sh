# same-box trial of extra compiler flags on the two blend translation units (rebuilds on the GPU box):  bash tools/ab_flags.sh
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
run() { $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1', d['value'], 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'])"; }
run "baseline"
for f in "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-schedule-relaxed-occupancy=1" "-mllvm -greedy-reverse-local-assignment=1" "-mllvm -amdgpu-enable-rewrite-partial-reg-uses=1"; do
  touch gs-sr_amd/csrc/gsr_blend.hip gs-sr_amd/csrc/gsr_blend_sp.hip
  if make -C gs-sr_amd/csrc BLEND_EXTRA="$f" > /tmp/mk.log 2>&1; then run "[$f]"; else echo "[$f] does not build: $(grep -m1 -i 'error\|unknown' /tmp/mk.log | cut -c1-120)"; fi
done
touch gs-sr_amd/csrc/gsr_blend.hip gs-sr_amd/csrc/gsr_blend_sp.hip; make -C gs-sr_amd/csrc > /dev/null 2>&1; run "baseline again"
