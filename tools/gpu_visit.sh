#!/bin/bash
# One GPU-box visit (round 5): usage (repo root on the GPU box): bash tools/gpu_visit.sh <tag> <what...>
#   tests        the whole -m gpu suite (no -x: every failure is seen) + smoke()
#   ab           200-step bench of the default library and of the listed variants on THIS box (same frames): AB_LIBS="- fast nofma_fast"
#   pytest:<expr>   python -m pytest tests -m gpu -k <expr>
#   ab2          in-process A/B (tools/ab_bench.py): AB_SPECS="d d@2 fast"
#   prof         rocprofv3 --kernel-trace --stats of the driver's command (no CPU baseline, no side legs)
#   bench        the driver's command as the driver runs it (side legs + CPU baseline) -> <tag>_bench_driver_cmd.json
#   pmc          HBM-side traffic: FETCH_SIZE / WRITE_SIZE passes + calibration (tools/pmc_traffic.sh), then tools/pmc_json.py -> profiles/pmc_traffic.json
#   pmcbig       the same passes on BASELINE configs[2] (tools/pmc_traffic_big.sh) -> <tag>_pmc_traffic_1280x960.json
#   factorial    tools/parity_factorial.py -> <tag>_parity_factorial.json
#   shared       aggregate frames/s of 1, 2, 4 replays sharing the GPU -> <tag>_shared_gpu.jsonl
#   variant:<v>  libefusion_hip_<v>.so beside the default: phase clocks (<v>_clocks), the frame / reference / steady-state parity tests through it
#   clocks       phase clocks of the persistent tracker (libefusion_hip_clocks.so: python -m elasticfusion_amd.build --variant clocks)
tag=${1:-run}; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests)
      timeout 900 python -m pytest tests -m gpu -q --timeout=300 --durations=10 > $out/${tag}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_gpu_tests.log
      grep -E "passed|failed|error|FAILED|ERROR" $out/${tag}_gpu_tests.log | tail -30 | cut -c1-300
      timeout 120 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
      tail -3 $out/${tag}_smoke.log ;;
    pytest:*)
      timeout 600 python -m pytest tests -m gpu -q --timeout=300 -k "${what#pytest:}" > $out/${tag}_gpu_tests_k.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_gpu_tests_k.log
      tail -25 $out/${tag}_gpu_tests_k.log | cut -c1-300 ;;
    ab2)   # in-process A/B (tools/ab_bench.py): AB_SPECS="d d@2 fast"
      timeout 400 python tools/ab_bench.py ${AB_ARGS:-} ${AB_SPECS:-d fast} 2>&1 | grep -v "^$" | tee -a $out/${tag}_ab.log | tail -12 ;;
    prof)   # rocprofv3 kernel stats of the driver's command (no CPU baseline, no side legs)
      cd /tmp
      timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --steps 20 --warmup 5 > $out/${tag}_prof_stdout.log 2>&1
      find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
      head -30 $out/${tag}_bench_kernel_stats.csv | cut -c1-160; tail -2 $out/${tag}_prof_stdout.log | cut -c1-600
      cd $GRAFT_REPO_ROOT ;;
    bench)
      timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2> $out/${tag}_bench_driver_cmd.err; echo "bench (driver's command) rc=$?"
      cut -c1-700 $out/${tag}_bench_driver_cmd.json; tail -2 $out/${tag}_bench_driver_cmd.err ;;
    pmc)
      bash tools/pmc_traffic.sh ${tag}_pmc 2>&1 | tail -12
      python tools/pmc_json.py ${tag}_pmc --out $out/${tag}_pmc_traffic.json 2>&1 | tail -30 ;;
    pmcbig)   # BASELINE configs[2] (1280x960, ~1 M pre-seeded surfels); after `pmc` of the same tag (its calibration is reused)
      bash tools/pmc_traffic_big.sh ${tag}_pmcbig 2>&1 | tail -6
      for c in FETCH_SIZE WRITE_SIZE; do cp $out/${tag}_pmc_${c}_calibration.txt $out/${tag}_pmcbig_${c}_calibration.txt; done
      python tools/pmc_json.py ${tag}_pmcbig --out $out/${tag}_pmc_traffic_1280x960.json 2>&1 | tail -20 ;;
    factorial)
      timeout 300 python tools/parity_factorial.py $out/${tag}_parity_factorial.json > $out/${tag}_parity_factorial.log 2>&1; echo "factorial rc=$?"
      python -c "
import json; s = json.load(open('$out/${tag}_parity_factorial.json'))['summary']
for k in ('fma_reference_order', 'nofma_fast_order', 'fma_fast_order'): print(k, s[k]['over_the_bar'], s[k]['pose_difference_m'])
print(s['motion_error_against_the_generating_trajectory_m'])" ;;
    shared)   # several replays sharing this GPU (tools/shared_gpu_bench.py)
      timeout 300 python tools/shared_gpu_bench.py --steps 150 --sequences ${SHARED_SEQUENCES:-1,2,4} 2>/dev/null | tee $out/${tag}_shared_gpu.jsonl ;;
    variant:*)   # a development build (python -m elasticfusion_amd.build --variant <name> [and <name>_clocks]) beside the default: clocks, parity, A/B
      v=${what#variant:}
      [ -f elasticfusion_amd/libefusion_hip_${v}_clocks.so ] && timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so elasticfusion_amd/libefusion_hip_${v}_clocks.so 140 > $out/${tag}_clocks_${v}.jsonl 2>$out/${tag}_clocks_${v}.err
      cat $out/${tag}_clocks_${v}.jsonl | cut -c1-900
      EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_${v}.so timeout 500 python -m pytest tests -m gpu -q --timeout=300 -k "${VARIANT_TESTS:-test_gpu_frame or vs_reference or test_gpu_steady}" > $out/${tag}_gpu_tests_${v}.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_gpu_tests_${v}.log
      tail -6 $out/${tag}_gpu_tests_${v}.log | cut -c1-300 ;;
    clocks)
      timeout 120 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/${tag}_clocks.jsonl 2>$out/${tag}_clocks.err; cat $out/${tag}_clocks.jsonl; tail -2 $out/${tag}_clocks.err ;;
    ab)
      for rep in 1 2; do
        for v in ${AB_LIBS:-- fast}; do
          # "-" = the default library; "name" = libefusion_hip_<name>.so; "+flag" = the default library with bench.py --flag (e.g. +round3-tracker)
          lib=""
          case $v in
            -) ;;
            +*) lib="--${v#+}" ;;
            *) lib="--library $GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_$v.so" ;;
          esac
          timeout 300 python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes $lib 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$v]', d['value'], 'fps', d['ms_per_step'], 'ms', d.get('frame_time_ms'))" | tee -a $out/${tag}_ab.log
        done
      done ;;
  esac
done
