# same-box A/B of the forward's LDS padding (20 KB for every variant against each variant's own staging size): rebuilds the library on the GPU box
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
run() { for a in "--variant ewa" "--variant plane" "--variant ewa --P 1500000" "--variant ewa --color-mode sh"; do $B $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1 [$a]', d['value'], 'fwd', s['blend_fwd'])"; done; }
run padded
touch gs-sr_amd/csrc/gsr_blend.hip; make -C gs-sr_amd/csrc BLEND_EXTRA=-DGSR_FWD_LDS_NOPAD=1 > /dev/null 2>&1
run natural
touch gs-sr_amd/csrc/gsr_blend.hip; make -C gs-sr_amd/csrc > /dev/null 2>&1
run padded_again
