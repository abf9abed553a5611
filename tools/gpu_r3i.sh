#!/bin/bash
# round 3: the rest of the -m gpu suite at HEAD (everything tools/gpu_r3g.sh did not run)
tag=${1:-r03i}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_global.py tests/test_gpu_loop.py tests/test_gpu_reloc.py tests/test_gpu_replay.py tests/test_gpu_closed_steady.py tests/test_gpu_one_frame.py tests/test_gpu_vs_reference.py tests/test_gpu_steady.py -m gpu -q --timeout=300 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -6 $out/${tag}_tests.log
