# Round-6 evidence, run on the GPU box:  bash tools/gpu_profile_r06.sh   -> gpurun_out/prof_r06 (copy the summaries into profiles/)
#   1. default bench (cpu_baseline, parity_full_size vs the float64 truth, method_iteration incl. graph replay)
#   2. rocprofv3 --kernel-trace --stats of the same timed loop, three variants; kernel timeline (gaps) of the surfel loop
#   3. PMC passes, each in its own run with --kernel-trace only: HBM traffic (FETCH_SIZE, WRITE_SIZE), SQ issue / wait counters, L2 hit / miss and
#      atomics -- for ALL THREE variants (round 2 had FETCH / WRITE / TCC for the surfel kernels only)
#   4. side points: P = 100k / 1M / 3M at 1080p, and the reference's default training resolution 1600x900 (gssr/cameras/utils.py:23-40)
#   5. sparse TSDF: throughput, kernel stats and HBM-traffic counters (smooth frame + config 5's tail)
#   6. BASELINE-size parity report with the float32-geometry floor (tools/full_parity_report.py, 14 cases)
mkdir -p gpurun_out/prof_r06; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06
cd /tmp && export TMPDIR=/tmp
timeout 1200 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
B="python $R/bench.py --no-cpu-baseline --no-method-iteration --no-graph-replay"
for v in surfel ewa plane; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- $B --variant $v > /dev/null 2>&1; echo stats $v rc=$?
  f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${v}_kernel_stats.csv
done
f=$(find $O/stats_surfel -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/timeline_gaps.py $O/stats_surfel > $O/timeline_surfel.json
rm -rf $O/stats_surfel $O/stats_ewa $O/stats_plane
Bs="$B --steps 8 --warmup 2 --stage-steps 1"
for v in surfel ewa plane; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
             "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r06_${v}_$i -- $Bs --variant $v > /dev/null 2>&1; echo "pmc $v pass $i rc=$?"
  done
done
python $R/tools/make_traffic.py r06 > /dev/null; cp $R/profiles/r06_pmc_summary.json $O/ 2>/dev/null; cp $R/profiles/traffic.json $O/ 2>/dev/null
rm -rf $R/gpurun_out/pmc_r06_*
: > $O/side_points.jsonl
side() { $B --variant $1 --P $2 --W $3 --H $4 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print(json.dumps({'variant':'$1','P':$2,'W':$3,'H':$4,'iters_per_s':d['value'],'R':d['config']['tile_instances_R'],'fwd_ms':round(s['preprocess']+s['depth_order']+s['binning']+s['blend_fwd'],4),'bwd_ms':round(s['bwd_memset']+s['blend_bwd']+s['preprocess_bwd'],4),'stage_ms':s}))" >> $O/side_points.jsonl; }
for v in surfel ewa plane; do side $v 300000 1600 900; done
for P in 100000 1000000 3000000; do for v in surfel ewa plane; do side $v $P 1920 1080; done; done
$B --variant ewa --color-mode sh 2>/dev/null | tail -1 > $O/bench_ewa_sh.json
$B --variant ewa 2>/dev/null | tail -1 > $O/bench_ewa.json
$B --variant plane 2>/dev/null | tail -1 > $O/bench_plane.json
# 5. sparse TSDF: both benchmarks, kernel stats and FETCH_SIZE / WRITE_SIZE of the TSDF kernels (tools/prof_tsdf_r06.sh -> gpurun_out/prof_r06_tsdf)
bash $R/tools/prof_tsdf_r06.sh > $O/tsdf_profile.log 2>&1; cp $R/gpurun_out/prof_r06_tsdf/*.json $R/gpurun_out/prof_r06_tsdf/*_kernel_stats.csv $O/ 2>/dev/null
cd /tmp
for m in scaffold-2dgs octree-pgsr; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/it_$m -- python $R/tools/iter_breakdown.py --method $m > /dev/null 2>&1
  f=$(find $O/it_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${m}_iteration_kernel_stats.csv; rm -rf $O/it_$m
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/it_losses -- python $R/tools/bench_losses.py > $O/bench_losses.json 2>/dev/null
f=$(find $O/it_losses -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/loss_kernel_stats.csv; rm -rf $O/it_losses
timeout 1500 python $R/tools/full_parity_report.py --big > $O/full_size_parity.jsonl 2> /dev/null; echo "parity report rc=$? (0 = every case inside its bars)"; wc -l $O/full_size_parity.jsonl
# 7. the ordering chain where it is not launch-bound (VERDICT r5 #4 i): P = 3 M surfels, kernel stats + HBM traffic of preprocess / depth sort / duplicate / tile sort
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st3m -- $B --variant surfel --P 3000000 --steps 20 --warmup 3 > /dev/null 2>&1; echo "3M stats rc=$?"
f=$(find $O/st3m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/surfel_3m_kernel_stats.csv; rm -rf $O/st3m
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3m_$c -- $B --variant surfel --P 3000000 --steps 6 --warmup 2 --stage-steps 1 > /dev/null 2>&1; echo "3M pmc $c rc=$?"
done
python $R/tools/chain_pmc.py $R/gpurun_out $O/surfel_3m_kernel_stats.csv > $O/chain_3m_pmc.json; rm -rf $R/gpurun_out/pmc3m_*
ls -la $O | head -40
