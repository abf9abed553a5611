cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# operator tier + frame parity first (golden bits of the compiled reduce.cu, compiled reference driver, oracle)
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_tracking or vs_reference or test_gpu_frame or bench_configs or test_gpu_fallback" > gpurun_out/r06t_tests_k.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06t_tests_k.log | cut -c1-300
AB_SPECS="d nopairs d@0 nopairs@0" bash tools/gpu_visit.sh r06t ab2
AB_ARGS="--big --steps 60" AB_SPECS="d nopairs" bash tools/gpu_visit.sh r06t_big ab2
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_nopairs_clocks.so elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06t_clocks.jsonl 2>gpurun_out/r06t_clocks.err; cut -c1-1500 gpurun_out/r06t_clocks.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 d d@0 2>&1 | grep "rep 0"
cp $(find /tmp/st_d -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r06t_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_n -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 nopairs@0 2>&1 | grep "rep 0"
cp $(find /tmp/st_n -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r06t_kernel_stats_nopairs_per_step.csv
head -8 $R/gpurun_out/r06t_kernel_stats.csv | cut -c1-150
grep se3_accum $R/gpurun_out/r06t_kernel_stats.csv $R/gpurun_out/r06t_kernel_stats_nopairs_per_step.csv | cut -c1-250
