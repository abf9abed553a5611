#!/bin/bash
# round 3, last seconds of the budget: the global-closure and relocalisation tests on the library with eft::init_model_pair
mkdir -p gpurun_out
timeout 85 python -m pytest tests/test_gpu_global.py tests/test_gpu_reloc.py -x -q --timeout=80 > gpurun_out/r03o_tests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03o_tests.log
