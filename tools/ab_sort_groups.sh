# tile sort: up to how many 16-block groups a scatter block sums the group histograms itself (above: digit-major matrix + k_scan_rows); same box
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay --steps 100 --warmup 10"
run() { $B $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1', '[$2]', d['value'], 'binning', s['binning'])"; }
run default ""; run default "--P 200000"; run default "--variant ewa"
for g in 96 160; do
  touch gs-sr_amd/csrc/gsr_binning.hip gs-sr_amd/csrc/gsr_api.hip
  make -C gs-sr_amd/csrc COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -DGSR_SORT_MAX_GROUPS=$g" > /tmp/mk.log 2>&1 || { echo "$g: build failed"; tail -3 /tmp/mk.log; continue; }
  run "groups<=$g" ""; run "groups<=$g" "--P 200000"; run "groups<=$g" "--variant ewa"
done
make -C gs-sr_amd/csrc clean > /dev/null 2>&1; make -C gs-sr_amd/csrc -j8 > /dev/null 2>&1; run default_again ""
