cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/st_big -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 60 --reps 1 d 2>&1 | grep "rep 0"
f=$(find /tmp/st_big -name "st_kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r06m_1280x960_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace -d /tmp/tl_big -o tl --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 d > /dev/null 2>&1
f=$(find /tmp/tl_big -name "tl_kernel_trace.csv" | head -1)
python $R/tools/overlap_timeline.py $f --frames 2 > $R/gpurun_out/r06m_timeline_1280x960.txt
sed -n 1,26p $R/gpurun_out/r06m_timeline_1280x960.txt
cd $R
bash tools/gpu_visit.sh r06m bench
