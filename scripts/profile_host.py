"""Host-side (Python) cost of one eager train step: cProfile over 100 steps on the bench workload."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import acm_gnn_amd
from acm_gnn_amd import data as D, distributed as DD, train as T

dev = torch.device("cuda:0")
adj, x_np, y_np, (tr, va, te), n = D.synthetic_dataset("twitch-gamer")
x_np = D.row_normalize_features(x_np)
low, deg = D.build_filters(adj)
ops = DD.make_sharded_operators(low, deg, dev)
x, y = torch.from_numpy(x_np).to(dev), torch.from_numpy(y_np).to(dev)
model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
w = T.row_weights(torch.from_numpy(tr).to(dev), n)
step = T.TrainStep(model, opt, x, ops, y, w)
for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
