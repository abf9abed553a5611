cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_visit.sh r07b tests
AB_SPECS="d sepinputs" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r07b ab2
AB_ARGS="--big --steps 60 --reps 2" AB_SPECS="d sepinputs" bash tools/gpu_visit.sh r07b_big ab2
