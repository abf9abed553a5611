cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/probe/sincos_probe.hip -o /tmp/sincos_probe 2>/dev/null && /tmp/sincos_probe | tee gpurun_out/r06x_sincos_probe.json
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_sincos.so timeout 700 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or vs_reference or test_gpu_steady or ops_linalg or test_gpu_fallback or bench_configs" > gpurun_out/r06x_tests_sincos.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06x_tests_sincos.log | cut -c1-300
AB_SPECS="d sincos" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r06x ab2
