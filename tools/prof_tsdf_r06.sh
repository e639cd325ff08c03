# Round-6 TSDF evidence, run on the GPU box: counter calibration for sparse 16-byte accesses (tools/microbench/hbm_granule), then kernel stats +
# HBM-traffic counters of the sparse-TSDF kernels for the smooth-surface frame (tools/bench_tsdf_sparse.py) and config 5's tail (tools/bench_tile_tail.py).
#   -> gpurun_out/prof_r06_tsdf/{hbm_granule.json, *.json, *_kernel_stats.csv, tsdf_pmc.json}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06_tsdf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G=$R/tools/microbench/hbm_granule
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g/stats -- $G > /dev/null 2>&1; echo "granule stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/g/pmc_$c -- $G > /dev/null 2>&1; echo "granule pmc $c rc=$?"; done
python $R/tools/hbm_granule_report.py $O/g > $O/hbm_granule.json; cat $O/hbm_granule.json; rm -rf $O/g
cmd_of() { case $1 in tsdf_sparse) echo "$R/tools/bench_tsdf_sparse.py";; tile_tail) echo "$R/tools/bench_tile_tail.py";; tile_tail_terrain) echo "$R/tools/bench_tile_tail.py --surface terrain";; esac; }
timeout 300 python $R/tools/bench_tsdf_sparse.py 2>/dev/null | tail -1 > $O/tsdf_sparse.json
timeout 400 python $R/tools/bench_tile_tail.py --count-updates 2>/dev/null | tail -1 > $O/tile_tail.json
timeout 400 python $R/tools/bench_tile_tail.py --surface terrain --count-updates 2>/dev/null | tail -1 > $O/tile_tail_terrain.json
for t in tsdf_sparse tile_tail tile_tail_terrain; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$t -- python $(cmd_of $t) > /dev/null 2>&1; echo "$t stats rc=$?"
  f=$(find $O/st_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${t}_kernel_stats.csv; rm -rf $O/st_$t
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${t}_$c -- python $(cmd_of $t) > /dev/null 2>&1; echo "$t pmc $c rc=$?"
  done
done
python $R/tools/tsdf_pmc.py $O > $O/tsdf_pmc.json; cat $O/tsdf_pmc.json
rm -rf $O/pmc_*
cat $O/tsdf_sparse.json $O/tile_tail.json $O/tile_tail_terrain.json
