cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_oracle.py tests/test_gpu_golden.py tests/test_gpu_relabel.py tests/test_gpu_edge_cases.py -q -x > gpurun_out/r02o_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o_pytest.log
python bench.py > gpurun_out/r02o_bench.json 2> gpurun_out/r02o_bench.err
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 ACM_FORCE_SHARDED=1 python bench.py --steps 20 --no-cpu-baseline --no-extras > gpurun_out/r02o_bench_forced_sharded.json 2> gpurun_out/r02o_bench_forced_sharded.err
python scripts/bench_configs.py penn94/acmgcnp/csrX cora/acmgcn/csrX > gpurun_out/r02o_cfg.jsonl 2>&1
python scripts/profile_host.py > gpurun_out/r02o_host.txt 2>&1
