#!/bin/bash
# One GPU-box visit, every process on a short leash (a wedged runtime must not eat the GPU budget).
# usage (repo root on the GPU box): bash tools/gpu_round.sh <tag> tests | bench
#   tests : the whole -m gpu suite + smoke()
#   bench : the driver's command (python bench.py), rocprofv3 kernel stats of the same command, PMC traffic passes (launch-per-step script:
#           k_se3_accum_fast + k_index_splat; default: k_track_fast), phase clocks of the persistent tracker, several sequences on one GPU
tag=${1:-run}
what=${2:-tests}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$what" = "tests" ]; then
  timeout 700 python -m pytest tests -m gpu -q --timeout=300 --durations=8 > $out/${tag}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_gpu_tests.log
  tail -16 $out/${tag}_gpu_tests.log | cut -c1-300
  timeout 120 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
  tail -3 $out/${tag}_smoke.log
  exit 0
fi
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2> $out/${tag}_bench_driver_cmd.err; echo "bench (driver's command) rc=$?"
cut -c1-400 $out/${tag}_bench_driver_cmd.json
timeout 420 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cut -c1-1500 $out/${tag}_bench.json; tail -3 $out/${tag}_bench.err
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --steps 20 --warmup 5 > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh ${tag}_pmc_perstep --per-step-tracker 2>&1 | tail -8
PMC_SKIP_CAL=1 bash tools/pmc_traffic.sh ${tag}_pmc_default 2>&1 | tail -6
for c in FETCH_SIZE WRITE_SIZE; do cp $out/${tag}_pmc_perstep_${c}_calibration.txt $out/${tag}_pmc_default_${c}_calibration.txt 2>/dev/null; done
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/${tag}_fast_clocks.jsonl 2>$out/${tag}_fast_clocks.err; cat $out/${tag}_fast_clocks.jsonl
timeout 200 python tools/shared_gpu_bench.py --steps 150 --sequences 1,2,4 > $out/${tag}_shared_gpu.jsonl 2> $out/${tag}_shared_gpu.err; cat $out/${tag}_shared_gpu.jsonl
