#!/bin/bash
# Last GPU-box visit of a round: the shared-GPU side measurement, the bench line of HEAD and rocprofv3 kernel stats of the same command.
# Every process on a short leash.  usage (repo root on the GPU box): bash tools/gpu_final.sh <tag>
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 260 python tools/shared_gpu_bench.py --steps 150 --sequences 1,2,4,8 > $out/${tag}_shared_gpu.jsonl 2> $out/${tag}_shared_gpu.err; echo "shared rc=$?"
cat $out/${tag}_shared_gpu.jsonl; tail -3 $out/${tag}_shared_gpu.err
timeout 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json; tail -3 $out/${tag}_bench.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-200
