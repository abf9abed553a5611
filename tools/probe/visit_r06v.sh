cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or vs_reference or bench_configs or test_gpu_fallback" > gpurun_out/r06v_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06v_tests_k.log | cut -c1-300
AB_SPECS="d prep_late nopairs" bash tools/gpu_visit.sh r06v ab2
AB_ARGS="--big --steps 60" AB_SPECS="d p_nostream p_icp p_all nopairs" bash tools/gpu_visit.sh r06v_big ab2
bash tools/gpu_visit.sh r06v clocks
