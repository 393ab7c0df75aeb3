for b in 1024 1280 1536 1792 2048 3584; do echo "epi blocks $b"; ACM_AGG_EPI_BLOCKS=$b ACM_AGG_BWD_ROLES=3 timeout 200 python - <<'PY'
import os, sys
sys.path.insert(0, "scripts")
from probe_pipeline import build
from acm_gnn_amd import data as D, functional as AF
wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
step, model = build(None, wl, use_graph=False)
for _ in range(5): step()
t = AF.KernelTimer(); AF.set_kernel_timer(t)
for _ in range(10): step()
AF.set_kernel_timer(None)
print({k: round(v[1] / v[0] * 1e3, 1) for k, v in t.summary().items() if "epi" in k})
PY
done
