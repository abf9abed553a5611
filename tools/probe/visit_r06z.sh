cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_visit.sh r06z tests
AB_SPECS="d sepmerge" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r06z ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d sepmerge" bash tools/gpu_visit.sh r06z_big ab2
