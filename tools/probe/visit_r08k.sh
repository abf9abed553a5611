cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame" > gpurun_out/r08k_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08k_tests_k.log | cut -c1-300
for v in clocks b8a1_clocks b32a1_clocks b64a2_clocks; do
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_$v.so 140 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['library'], d['whole_launch_us'], 'so3x', d['so3_per_iteration_us']['exchange'], 'searchA', d['se3_per_iteration_us']['search_publish_A'], 'A', d['resident_photometric_wavefront_us']['publish_A_sweep_sigma'], 'restB', d['se3_per_iteration_us']['rest_of_tasks_trees_publish_B'], 'B', d['se3_per_iteration_us']['exchange_B'], d['iteration_us_by_level'])" | tee -a gpurun_out/r08k_clocks.txt
done
AB_SPECS="d b8a1 b32a1 b64a2" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08k ab2
AB_ARGS="--big --steps 60 --reps 2" AB_SPECS="d b8a1 b32a1 b64a2" bash tools/gpu_visit.sh r08k_big ab2
