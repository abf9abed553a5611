cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 500 python tools/ab_bench.py --big --steps 60 --reps 2 d d+nores 2>&1 | grep -v "^$" | tee gpurun_out/r06c_ab_big.log | tail -8
