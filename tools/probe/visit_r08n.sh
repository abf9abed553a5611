cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in d nodense; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$v -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 $v 2>&1 | grep "rep 0"
cp $(find /tmp/st_$v -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r08n_${v}_640.csv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/stb_$v -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 $v 2>&1 | grep "rep 0"
cp $(find /tmp/stb_$v -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r08n_${v}_big.csv
done
