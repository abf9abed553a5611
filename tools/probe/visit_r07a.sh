cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -k "test_gpu_fallback or test_gpu_frame or ops_map or ops_tracking or gpu_vs_reference or capi" > gpurun_out/r07a_tests_k.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r07a_tests_k.log | cut -c1-300
bash tools/gpu_visit.sh r07a bench
