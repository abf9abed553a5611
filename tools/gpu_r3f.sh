#!/bin/bash
# round 3, sixth GPU visit: (1) level-0 update step fused into the search launch: parity (agreement test) and A/B through bench.py's flag;
# (2) k_se3_accum with CH = 10 at 1280x960 (two dependent rounds instead of four) against CH = 5, on the pre-seeded 1 M-surfel map.
tag=${1:-r03f}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_frame.py -m gpu -q --timeout=150 -k "persistent_and_per_step or tracking_and_fusion" > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -5 $out/${tag}_tests.log
show='import sys, json
d = json.loads(sys.stdin.read()); r = d["roofline"]; s = d["roofline_index_splat"]
print(sys.argv[1], d["value"], "fps accum L0", r["avg_us"], "us frac", r["frac"], "splat", s["avg_us"], "us frac", s["frac"], "surfels", d["config"]["surfels_end"])'
for rep in 1 2; do
  for f in "" "--fused-step"; do
    timeout 200 python bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes $f 2>/dev/null | python -c "$show" "[640x480 ${f:-three-launch}]" | tee -a $out/${tag}_ab.log
  done
done
for rep in 1 2; do
  for v in default bigch; do
    lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip.so
    [ $v = bigch ] && lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_bigch.so
    EF_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes --width 1280 --height 960 --preseed 1048576 --steps 60 --warmup 10 2>/dev/null | python -c "$show" "[1280x960 1M $v]" | tee -a $out/${tag}_ab.log
  done
done
