cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame and (pipelin or overlap or two_contexts or scripts_agree)" > gpurun_out/r06g_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06g_tests.log | cut -c1-300
AB_SPECS="d d+ov1 d+ov4" bash tools/gpu_visit.sh r06g ab2
cd /tmp
for spec in d d+ov1; do
  tag=$(echo $spec | tr '+' '_')
  timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o tl --output-format csv -- python $R/tools/ab_bench.py --steps 40 --reps 1 $spec > $R/gpurun_out/r06g_tl_$tag.log 2>&1
  f=$(find /tmp/tl_$tag -name "tl_kernel_trace.csv" | head -1)
  python $R/tools/overlap_timeline.py $f --frames 3 > $R/gpurun_out/r06g_timeline_$tag.txt
  tail -1 $R/gpurun_out/r06g_timeline_$tag.txt
done
cd $R
sed -n 1,32p gpurun_out/r06g_timeline_d_ov1.txt
