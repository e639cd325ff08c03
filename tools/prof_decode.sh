# kernel times of the neural-Gaussian decode (tools/bench_decode.py under rocprofv3); GSR_LIB_PATH selects the library, e.g. an A/B build
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pd; GSR_LIB_PATH=$lib timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python $GRAFT_REPO_ROOT/tools/bench_decode.py > /tmp/pd.log 2>&1
  f=$(find /tmp/pd -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "$(basename ${lib:-default}): $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('k_dec_', 'k_wgrad')): print(r['Name'].split('(')[0][-16:], round(float(r['AverageNs'])/1e3,1), '|', end=' ')
")"; else echo "$lib: no stats"; tail -3 /tmp/pd.log; fi
done
