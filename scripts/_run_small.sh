cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_small.py -q 2>&1 | grep -E "^E  +|passed|failed|FAILED|Error" | cut -c1-1500 > gpurun_out/small_tests.log
