#!/bin/bash
# GPU box: a synthetic .klg through the REFERENCE's own front end over this library (oracle/_ref/reference_front_end) and through
# tools/efusion_replay.cpp; the two trajectories must be the same file.  Open loop (-o / default) and closed loop.
tag=${1:-rfe}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out /tmp/rfe_run
cd $GRAFT_REPO_ROOT
python - <<'PY'
from elasticfusion_amd import synth
s = synth.Sequence(0xEF0001)
synth.write_klg('/tmp/rfe_run/a.klg', [s.frame(k) for k in range(12)])
import shutil; shutil.copy('/tmp/rfe_run/a.klg', '/tmp/rfe_run/b.klg')
PY
cd /tmp/rfe_run
# MainController's defaults are -ic 40000 -ie 4e-05 (thresholds of the local closure); efusion_replay takes the same flags
timeout 120 $GRAFT_REPO_ROOT/oracle/_ref/reference_front_end -l /tmp/rfe_run/a.klg -q -o > $out/${tag}_reference_open.log 2>&1; echo "reference front end (open loop) rc=$?"
timeout 120 $GRAFT_REPO_ROOT/elasticfusion_amd/efusion_replay -l /tmp/rfe_run/b.klg -q > $out/${tag}_replay_open.log 2>&1; echo "efusion_replay (open loop) rc=$?"
cmp /tmp/rfe_run/a.klg.freiburg /tmp/rfe_run/b.klg.freiburg && echo "OPEN LOOP: identical trajectories ($(wc -l < /tmp/rfe_run/a.klg.freiburg) poses)"
cp /tmp/rfe_run/a.klg.freiburg $out/${tag}_reference_open.freiburg; cp /tmp/rfe_run/b.klg.freiburg $out/${tag}_replay_open.freiburg
rm -f /tmp/rfe_run/*.freiburg
timeout 120 $GRAFT_REPO_ROOT/oracle/_ref/reference_front_end -l /tmp/rfe_run/a.klg -q > $out/${tag}_reference_closed.log 2>&1; echo "reference front end (closed loop) rc=$?"
timeout 120 $GRAFT_REPO_ROOT/elasticfusion_amd/efusion_replay -l /tmp/rfe_run/b.klg -q -cl > $out/${tag}_replay_closed.log 2>&1; echo "efusion_replay (closed loop) rc=$?"
cmp /tmp/rfe_run/a.klg.freiburg /tmp/rfe_run/b.klg.freiburg && echo "CLOSED LOOP: identical trajectories"
cp /tmp/rfe_run/a.klg.freiburg $out/${tag}_reference_closed.freiburg; cp /tmp/rfe_run/b.klg.freiburg $out/${tag}_replay_closed.freiburg
