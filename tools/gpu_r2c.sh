#!/bin/bash
# round-2 visit C: new-feature tests, fused-vs-split A/B, kernel trace of the product build
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=r02c
timeout 300 python -m pytest tests/test_gpu_ops_tracking.py tests/test_gpu_frame.py::test_tracking_and_fusion_sequence "tests/test_gpu_frame.py::test_tracker_configurations_match_oracle" -m gpu -x -q > $out/${tag}_quick.log 2>&1; echo "quick rc=$?"; tail -3 $out/${tag}_quick.log
timeout 420 python -m pytest tests/test_gpu_global.py tests/test_gpu_replay.py tests/test_gpu_loop.py tests/test_gpu_frame.py::test_context_used_from_another_thread tests/test_gpu_frame.py::test_trajectory_log_grows_without_bound tests/test_gpu_frame.py::test_reference_download_mode_matches_the_reference_buffer_choice tests/test_gpu_frame.py::test_capacity_overflow_is_reported -m gpu -q > $out/${tag}_new.log 2>&1; echo "new rc=$?"; tail -40 $out/${tag}_new.log | cut -c1-300
bash tools/gpu_ab.sh $tag - split r02a
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --frames-cache /tmp/efframes > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
head -14 $out/${tag}_kernel_stats.csv | cut -c1-150
