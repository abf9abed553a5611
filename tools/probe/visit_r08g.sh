cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d pre_pertap" bash tools/gpu_visit.sh r08g_big ab2
