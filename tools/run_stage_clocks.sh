cd $GRAFT_REPO_ROOT && EF_HIPCC_FLAGS=-DEF_STAGE_CLOCKS python -m elasticfusion_amd.build --force > /dev/null 2>&1; python tools/stage_clocks.py 2>&1 | tail -8
