cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_visit.sh r06s tests
AB_SPECS="d" bash tools/gpu_visit.sh r06s ab2
bash tools/gpu_visit.sh r06s prof
