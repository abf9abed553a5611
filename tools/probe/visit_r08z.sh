cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_visit.sh r08z tests bench
