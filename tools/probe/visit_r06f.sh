cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for spec in d d+ov1; do
  tag=$(echo $spec | tr '+' '_')
  timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o tl --output-format csv -- python $R/tools/ab_bench.py --steps 40 --reps 1 $spec > $R/gpurun_out/r06f_tl_$tag.log 2>&1
  f=$(find /tmp/tl_$tag -name "tl_kernel_trace.csv" | head -1)
  head -2 $f | cut -c1-400
  python $R/tools/overlap_timeline.py $f --frames 3 > $R/gpurun_out/r06f_timeline_$tag.txt
  tail -3 $R/gpurun_out/r06f_tl_$tag.log | cut -c1-200
  tail -1 $R/gpurun_out/r06f_timeline_$tag.txt
done
cd $R
sed -n 1,60p gpurun_out/r06f_timeline_d_ov1.txt
