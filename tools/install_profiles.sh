# copy the summaries of gpurun_out/prof_r06 (tools/gpu_profile_r06.sh) into profiles/ and stamp the PMC summary with the measured revision:  bash tools/install_profiles.sh
# Order on the GPU box matters for one pair: bench.py quotes profiles/traffic.json + profiles/r06_pmc_summary.json in its line, so the committed
# r06_bench_default.json must come from a run made AFTER these two were installed (tools/gpu_profile_r06.sh run twice, or `python bench.py` once more).
O=gpurun_out/prof_r06; REV=$(git rev-parse --short HEAD)
for f in bench_default bench_ewa bench_plane bench_ewa_sh; do [ -s $O/$f.json ] && cp $O/$f.json profiles/r06_$f.json; done
for v in surfel ewa plane; do cp $O/${v}_kernel_stats.csv profiles/r06_${v}_kernel_stats.csv; done
cp $O/timeline_surfel.json profiles/r06_timeline_surfel.json; cp $O/side_points.jsonl profiles/r06_side_points.jsonl; cp $O/tsdf_sparse.json profiles/r06_tsdf_sparse.json
[ -s $O/tile_tail.json ] && cp $O/tile_tail.json profiles/r06_tile_tail.json
[ -s $O/tsdf_pmc.json ] && cp $O/tsdf_pmc.json profiles/r06_tsdf_pmc.json
[ -s $O/tile_tail_terrain.json ] && cp $O/tile_tail_terrain.json profiles/r06_tile_tail_terrain.json
[ -s $O/hbm_granule.json ] && cp $O/hbm_granule.json profiles/r06_hbm_granule.json
for t in tsdf_sparse tile_tail tile_tail_terrain; do [ -s $O/${t}_kernel_stats.csv ] && cp $O/${t}_kernel_stats.csv profiles/r06_${t}_kernel_stats.csv; done
[ -s $O/chain_3m_pmc.json ] && cp $O/chain_3m_pmc.json profiles/r06_chain_3m_pmc.json; [ -s $O/surfel_3m_kernel_stats.csv ] && cp $O/surfel_3m_kernel_stats.csv profiles/r06_surfel_3m_kernel_stats.csv
[ -s $O/full_size_parity.jsonl ] && cp $O/full_size_parity.jsonl profiles/r06_full_size_parity.jsonl
cp $O/scaffold-2dgs_iteration_kernel_stats.csv profiles/r06_scaffold2dgs_iteration_kernel_stats.csv; cp $O/octree-pgsr_iteration_kernel_stats.csv profiles/r06_octree_pgsr_iteration_kernel_stats.csv
cp $O/loss_kernel_stats.csv profiles/r06_loss_kernel_stats.csv 2>/dev/null; cp $O/bench_losses.json profiles/r06_bench_losses.json 2>/dev/null
cp $O/traffic.json profiles/traffic.json; python tools/kernel_resources.py > profiles/r06_kernel_resources.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/r06_pmc_summary.json'))
m={'what':'rocprofv3 --pmc passes of tools/gpu_profile_r06.sh (one counter set per run, --kernel-trace only), per-launch averages over the bench loop; '
          'hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B on gfx950)',
   'workload':'bench.py --steps 8 --warmup 2, P = 300000, 1920x1080, three variants', 'git_revision_of_the_measured_library':'$REV'}
out={'_meta':m}; out.update({k:v for k,v in d.items() if k!='_meta'})
json.dump(out, open('profiles/r06_pmc_summary.json','w'), indent=1)
b=json.loads(open('profiles/r06_bench_default.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['stage_ms'])
print({k:(v.get('iters_per_s'), (v.get('graph_replay') or {}).get('iters_per_s'), v.get('wall_over_kernel'), v.get('launches_per_iter')) for k,v in b['method_iteration'].items()})
print(b['graph_replay'].get('iters_per_s'), b['parity_full_size']['verdict'], b['cpu_baseline']['value'], b['roofline']['frac'], b['roofline']['traffic'])
PY
