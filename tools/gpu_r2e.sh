#!/bin/bash
# round-2 visit E: tail rewrite check, streaming ceiling probe, A/B against the previous build, phase clocks, kernel trace
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=r02e
timeout 200 python -m pytest tests/test_gpu_ops_tracking.py::test_icp_step tests/test_gpu_ops_tracking.py::test_rgb_residual_and_step tests/test_gpu_frame.py::test_tracking_and_fusion_sequence "tests/test_gpu_frame.py::test_small_and_odd_resolutions_match_oracle" -m gpu -q --timeout=150 > $out/${tag}_quick.log 2>&1; echo "quick rc=$?"; tail -5 $out/${tag}_quick.log | cut -c1-300
timeout 60 tools/probe/stream_probe > $out/${tag}_stream_probe.json 2>&1; cat $out/${tag}_stream_probe.json
bash tools/gpu_ab.sh $tag - split
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_clocks.so timeout 120 python tools/accum_clocks.py 8 > $out/${tag}_accum_clocks.txt 2>&1; echo "clocks rc=$?"; tail -7 $out/${tag}_accum_clocks.txt | head -6
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --frames-cache /tmp/efframes > $out/${tag}_prof_stdout.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
head -30 $out/${tag}_kernel_stats.csv | cut -c1-150
