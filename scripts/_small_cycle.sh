cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_small.py -q 2>&1 | grep -E "^E  +|passed|failed|FAILED|Error" | cut -c1-600 > gpurun_out/small_tests.log
cat gpurun_out/small_tests.log
python scripts/bench_configs.py cora/acmgcn/auto squirrel/acmgcnp+A/auto chameleon/acmgcnp+A/auto > gpurun_out/r05_small_configs.jsonl 2> gpurun_out/r05_small_configs.err
cut -c1-200 gpurun_out/r05_small_configs.jsonl
bash scripts/configs_kernel_stats.sh cora/acmgcn/auto squirrel/acmgcnp+A/auto chameleon/acmgcnp+A/auto
grep -v "at::\|compute_cuda" gpurun_out/configs_kernel_stats.txt
