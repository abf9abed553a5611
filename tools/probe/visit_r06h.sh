cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or test_gpu_replay or reference_front_end" > gpurun_out/r06h_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06h_tests.log | cut -c1-300
AB_SPECS="d d+host d+ov1+host" bash tools/gpu_visit.sh r06h ab2
cd /tmp
for spec in d+host; do
  tag=$(echo $spec | tr '+' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_$tag -o tl --output-format csv -- python $R/tools/ab_bench.py --steps 40 --reps 1 $spec > $R/gpurun_out/r06h_tl_$tag.log 2>&1
  f=$(find /tmp/tl_$tag -name "tl_kernel_trace.csv" | head -1)
  python $R/tools/overlap_timeline.py $f --frames 2 > $R/gpurun_out/r06h_timeline_$tag.txt
  tail -1 $R/gpurun_out/r06h_timeline_$tag.txt
  find /tmp/tl_$tag -name "*memory_copy*" | head -2
done
cd $R
sed -n 1,28p gpurun_out/r06h_timeline_d_host.txt
