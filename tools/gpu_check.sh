mkdir -p gpurun_out; timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$?; tail -3 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for v in surfel ewa plane; do timeout 300 python $GRAFT_REPO_ROOT/bench.py --variant $v --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_$v.json 2>/dev/null; python -c "import json;d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/bench_$v.json'));print('$v',d['value'],d['stage_ms'])"; done
