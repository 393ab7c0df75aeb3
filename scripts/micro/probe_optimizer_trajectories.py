import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch, torch.nn.functional as F, numpy as np
import acm_gnn_amd
from acm_gnn_amd import data as D, graph
import bench_dropin_route as B
DEV = torch.device("cuda:0")
import scipy.sparse as sp
wl = D.bench_workload("twitch-gamer", node_order="degree")
n = wl["adj"].shape[0]
low = B.coo(wl["low"]); high = B.coo(sp.identity(n, dtype=np.float32, format="csr") - wl["low"])
x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
idx = torch.from_numpy(wl["splits"][0]).to(DEV)
res = {}
for name, cls in (("torch", torch.optim.AdamW), ("fused", acm_gnn_amd.FusedAdamW), ("torch2", torch.optim.AdamW)):
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    opt = cls(model.parameters(), lr=0.05, weight_decay=1e-3)
    ls = []
    for _ in range(40):
        model.train(); opt.zero_grad()
        out = F.log_softmax(model(x, low, high, None), dim=1)
        loss = F.nll_loss(out[idx], y[idx]); loss.backward(); opt.step(); ls.append(float(loss))
    res[name] = ls
for i in (0, 1, 2, 5, 10, 20, 39):
    print(i, res["torch"][i], res["fused"][i], res["torch2"][i])
