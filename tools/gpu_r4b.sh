#!/bin/bash
# round 4 visit b: fast order A/B on one box (default persistent / per-step / round-3 product) + kernel stats
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { tag=$1; shift; "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('[$tag]', d['value'], 'fps', r['avg_us'], 'us accum L0 (frac', r['frac'], ') splat', d['roofline_index_splat']['avg_us'], 'us')" | tee -a $out/r04b_ab.log; }
B="python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes"
for rep in 1 2; do
  run fast_persistent timeout 200 $B
  run fast_per_step timeout 200 $B --per-step-tracker
  EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_reforder.so run r3_product timeout 200 $B
done
run fast_close_loops timeout 200 $B --close-loops
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_reforder.so run r3_close_loops timeout 200 $B --close-loops
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r04b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes > /dev/null 2>&1
find /tmp/prof -name "r04b_kernel_stats.csv" -exec cp {} $out/r04b_bench_kernel_stats.csv \;
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o r04bp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes --per-step-tracker > /dev/null 2>&1
find /tmp/prof2 -name "r04bp_kernel_stats.csv" -exec cp {} $out/r04b_per_step_kernel_stats.csv \;
head -14 $out/r04b_bench_kernel_stats.csv | cut -c1-150
head -8 $out/r04b_per_step_kernel_stats.csv | cut -c1-150
