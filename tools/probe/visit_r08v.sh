cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
AB_SPECS="d sb2_12 sb2_24 sa_8 sb1_8 sall" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08v ab2
