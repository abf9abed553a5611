cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_visit.sh r06w tests
AB_SPECS="d shfl" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r06w ab2
bash tools/gpu_visit.sh r06w clocks
