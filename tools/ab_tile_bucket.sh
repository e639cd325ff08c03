#!/bin/bash
# same-box A/B of the one-pass bucket sort on the tile id against the two-pass radix sort (GSR_TILE_BUCKET=0): headline step and its stages
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-method-iteration --no-graph-replay"
for rep in 1 2 3; do
  for v in surfel ewa; do
    $B --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bucket $v', d['value'], d['stage_ms'])"
    GSR_TILE_BUCKET=0 $B --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('radix  $v', d['value'], d['stage_ms'])"
  done
done
