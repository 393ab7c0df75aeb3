import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
import acm_gnn_amd
from acm_gnn_amd import data as D, distributed as DD, train as T, functional as AF
dev = torch.device("cuda:0")
wl = D.bench_workload("twitch-gamer")
n = wl["adj"].shape[0]
ops = DD.make_sharded_operators(wl["low"], wl["deg"], dev)
x, y = torch.from_numpy(wl["x"]).to(dev), torch.from_numpy(wl["y"]).to(dev)
model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
w = T.row_weights(torch.from_numpy(wl["splits"][0]).to(dev), n)
probe = AF.KernelTimer(only="conv_agg_bwd", external=True)
AF.set_kernel_timer(probe)
st = T.TrainStep(model, opt, x, ops, y, w, use_graph=True)
AF.set_kernel_timer(None)
print({k: len(v) for k, v in probe.captured.items()})
for _ in range(3): st()
vals = []
for _ in range(10):
    st(); vals += probe.captured_ms(list(probe.captured)[0])
print(vals)
