cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scale.py -q -k "aggregate_first_for_wide" 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300
python scripts/bench_configs.py arxiv-year/acmgcnp 2>/dev/null | cut -c1-900
ACM_TUNING=rewrites=6 python scripts/bench_configs.py arxiv-year/acmgcnp 2>/dev/null | cut -c1-900
python scripts/bench_scale.py pokec 2>/dev/null | cut -c1-1200
