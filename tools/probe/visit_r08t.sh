cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or fallback" > gpurun_out/r08t_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08t_tests_k.log | cut -c1-300
for v in clocks modeone_clocks clocks modeone_clocks; do
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_$v.so 140 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['library'], d['whole_launch_us'], 'begin', d['begin_us'])" | tee -a gpurun_out/r08t_clocks.txt
done
AB_SPECS="d modeone" AB_ARGS="--reps 4" bash tools/gpu_visit.sh r08t ab2
