cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/gpu_ab.sh r01j - "EF_ASSOC_ROWWALK=1"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/profj -o j --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
find /tmp/profj -name "j_kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r01j_kernel_stats.csv \;
EF_ASSOC_ROWWALK=1 rocprofv3 --kernel-trace --stats -d /tmp/profj2 -o j2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
find /tmp/profj2 -name "j2_kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r01j_rowwalk_kernel_stats.csv \;
