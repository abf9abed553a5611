#!/bin/bash
# round-2 visit D: global-closure tests, accum phase clocks, bench + kernel trace of the product build (every process on a short leash)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=r02d
timeout 200 python -m pytest tests/test_gpu_global.py tests/test_gpu_ops_tracking.py::test_icp_step tests/test_gpu_frame.py::test_tracking_and_fusion_sequence -m gpu -q --timeout=150 > $out/${tag}_new.log 2>&1; echo "new rc=$?"; tail -30 $out/${tag}_new.log | cut -c1-300
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_clocks.so timeout 120 python tools/accum_clocks.py 8 > $out/${tag}_accum_clocks.txt 2>&1; echo "clocks rc=$?"; cat $out/${tag}_accum_clocks.txt | tail -14
timeout 200 python bench.py --steps 200 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; cut -c1-1500 $out/${tag}_bench.json; tail -3 $out/${tag}_bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/${tag}_prof_stdout.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
head -30 $out/${tag}_kernel_stats.csv | cut -c1-150
tail -3 $out/${tag}_prof_stdout.log | cut -c1-300
