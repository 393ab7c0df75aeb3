cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/gate_r05.log
tail -15 gpurun_out/gate_r05.log
