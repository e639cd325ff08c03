# copy the summaries of gpurun_out/prof_r03 (tools/gpu_profile_r03.sh) into profiles/ and stamp the PMC summary with the measured revision:  bash tools/install_profiles.sh
O=gpurun_out/prof_r03; REV=$(git rev-parse --short HEAD)
cp $O/bench_default.json profiles/r03_bench_default.json; cp $O/bench_ewa.json profiles/r03_bench_ewa.json; cp $O/bench_plane.json profiles/r03_bench_plane.json; cp $O/bench_ewa_sh.json profiles/r03_bench_ewa_sh.json
for v in surfel ewa plane; do cp $O/${v}_kernel_stats.csv profiles/r03_${v}_kernel_stats.csv; done
cp $O/timeline_surfel.json profiles/r03_timeline_surfel.json; cp $O/side_points.jsonl profiles/r03_side_points.jsonl; cp $O/tsdf_sparse.json profiles/r03_tsdf_sparse.json
cp $O/scaffold-2dgs_iteration_kernel_stats.csv profiles/r03_scaffold2dgs_iteration_kernel_stats.csv; cp $O/octree-pgsr_iteration_kernel_stats.csv profiles/r03_octree_pgsr_iteration_kernel_stats.csv
cp $O/loss_kernel_stats.csv profiles/r03_loss_kernel_stats.csv 2>/dev/null; cp $O/bench_losses.json profiles/r03_bench_losses.json 2>/dev/null
cp $O/traffic.json profiles/traffic.json; python tools/kernel_resources.py > profiles/r03_kernel_resources.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/r03_pmc_summary.json')); old=json.load(open('profiles/r03_pmc_summary.json'))
m=old['_meta']; m['git_revision_of_the_measured_library']='$REV'
out={'_meta':m}; out.update({k:v for k,v in d.items() if k!='_meta'})
json.dump(out, open('profiles/r03_pmc_summary.json','w'), indent=1)
b=json.loads(open('profiles/r03_bench_default.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['stage_ms'])
print({k:(v.get('iters_per_s'), (v.get('graph_replay') or {}).get('iters_per_s'), v.get('wall_over_kernel'), v.get('launches_per_iter')) for k,v in b['method_iteration'].items()})
print(b['graph_replay'].get('iters_per_s'), b['parity_full_size']['verdict'], b['cpu_baseline']['value'], b['roofline']['frac'], b['roofline']['traffic'])
for f in ('ewa','plane','ewa_sh'):
    e=json.loads(open(f'profiles/r03_bench_{f}.json').read().strip().splitlines()[-1]); print(f, e['value'], e['stage_ms'])
t=json.load(open('profiles/r03_timeline_surfel.json')); print(t['period_us'], t['kernel_us'], t['gap_us'], t['launches_per_iter'])
for k in d:
    if k.startswith('k_blend_bwd_sp') or k.startswith('k_blend_fwd'):
        e=d[k]; print(k, e['SQ_INSTS_VALU'], e['hbm_bytes_per_launch'], round(e['TCC_HIT_sum']/(e['TCC_HIT_sum']+e['TCC_MISS_sum']),3))
PY
