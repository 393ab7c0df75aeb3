cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_small.py -q -k "training_run or evaluation" 2>&1 | grep -E "^E  |passed|failed|FAILED|Error|Mismatch|Greatest" | cut -c1-300 > gpurun_out/small_tests.log
cat gpurun_out/small_tests.log
