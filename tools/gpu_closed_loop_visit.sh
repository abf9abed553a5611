out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_loop.py tests/test_gpu_closed_steady.py tests/test_gpu_global.py tests/test_gpu_reloc.py tests/test_gpu_replay.py tests/test_gpu_reference_front_end.py -m gpu -q --timeout=300 > $out/r04p_gpu_tests_closed_loop.log 2>&1; echo "pytest rc=$?" >> $out/r04p_gpu_tests_closed_loop.log
tail -8 $out/r04p_gpu_tests_closed_loop.log | cut -c1-300
timeout 150 python bench.py --no-cpu-baseline --no-side-legs --close-loops 2>/dev/null > $out/r04p_closeloops_bench.json; cut -c1-260 $out/r04p_closeloops_bench.json
