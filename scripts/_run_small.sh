cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scale.py -q -k "aggregate_first_for_wide" 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300
python scripts/bench_scale.py pokec 2>/dev/null | cut -c1-1200
ACM_TUNING=rewrites=7 python - <<'PY'
import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import acm_gnn_amd.functional as AF
AF.AGG_WIDE_MIN_DEGREE = 1
import bench_configs as B
print(json.dumps(B.run("arxiv-year/acmgcnp", B.CONFIGS["arxiv-year/acmgcnp"]))[:900])
PY
