#!/bin/bash
# round 3, last call: the C++ class's getGraph() through the headless replay front end + smoke() on the rebuilt libraries
mkdir -p gpurun_out
(timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > gpurun_out/r03l_smoke.log 2>&1; echo "smoke rc=$?"
(timeout 170 python -m pytest tests/test_gpu_replay.py -x -q) > gpurun_out/r03l_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03l_tests.log
