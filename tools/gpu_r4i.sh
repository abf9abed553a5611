#!/bin/bash
# rocprofv3 kernel stats of the bench command (short leash: round 4's first attempt sat in rocprofv3 until the limit)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
date +%s > $out/r04i_t0
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r04i --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --steps 20 --warmup 5 > $out/r04i_prof_stdout.log 2>&1; echo "rocprof rc=$?"
date +%s > $out/r04i_t1
find /tmp/prof -name "*kernel_stats.csv" | head; find /tmp/prof -name "r04i_kernel_stats.csv" -exec cp {} $out/r04i_bench_kernel_stats.csv \;
head -30 $out/r04i_bench_kernel_stats.csv | cut -c1-170
tail -3 $out/r04i_prof_stdout.log | cut -c1-400
