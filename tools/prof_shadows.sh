cd /tmp; export TMPDIR=/tmp
for sh in 1 0; do
rm -rf /tmp/ks$sh
GSR_PIPE_SHADOWS=$sh rocprofv3 --kernel-trace --stats -d /tmp/ks$sh -o ks --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_pipeline_octree_pgsr.py --steps 40 --warmup 8 > /dev/null 2>&1
cp /tmp/ks$sh/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/octree_pgsr_shadows${sh}_kernel_stats.csv
done
