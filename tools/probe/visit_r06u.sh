cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
AB_SPECS="d nopairs p_icp p_search p_rgb p_none" bash tools/gpu_visit.sh r06u ab2
AB_ARGS="--big --steps 60" AB_SPECS="d nopairs p_none" bash tools/gpu_visit.sh r06u_big ab2
