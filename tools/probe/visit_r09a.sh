cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash tools/gpu_visit.sh r09a prof clocks pmc pmcbig
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_d -o tl --output-format csv -- python $R/tools/ab_bench.py --steps 60 --reps 1 d > /dev/null 2>&1
f=$(find /tmp/tl_d -name "tl_kernel_trace.csv" | head -1)
python $R/tools/overlap_timeline.py $f --frames 3 > $R/gpurun_out/r09a_timeline_640x480.txt
sed -n 1,24p $R/gpurun_out/r09a_timeline_640x480.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_big -o tl --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 d > /dev/null 2>&1
f=$(find /tmp/tl_big -name "tl_kernel_trace.csv" | head -1)
python $R/tools/overlap_timeline.py $f --frames 2 > $R/gpurun_out/r09a_timeline_1280x960.txt
sed -n 1,22p $R/gpurun_out/r09a_timeline_1280x960.txt
