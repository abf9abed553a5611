cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 2>/dev/null | tee gpurun_out/r08q_clocks.jsonl | cut -c1-1100
AB_SPECS="d a1 a4 a8" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08q ab2
