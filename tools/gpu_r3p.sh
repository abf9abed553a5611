#!/bin/bash
# round 3, the last seconds: smoke() and the open-loop frame tests on the final library
mkdir -p gpurun_out
(timeout 20 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > gpurun_out/r03p_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03p_smoke.log
timeout 25 python -m pytest tests/test_gpu_frame.py -x -q --timeout=20 -k "persistent or first_frames or graph" > gpurun_out/r03p_tests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03p_tests.log
