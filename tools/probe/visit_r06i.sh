cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
AB_SPECS="d ez1 ez2" bash tools/gpu_visit.sh r06i ab2
cd /tmp
for spec in d ez1 ez2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$spec -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 $spec > /dev/null 2>&1
  f=$(find /tmp/st_$spec -name "st_kernel_stats.csv" | head -1)
  echo "== $spec"; grep "k_surface_splat\|k_surface_resolve" $f | cut -d, -f1-8 | cut -c1-60,200-400
  cp $f $R/gpurun_out/r06i_kernel_stats_$spec.csv
done
