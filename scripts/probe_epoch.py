#!/usr/bin/env python3
"""Run ON THE GPU BOX: the reference's epoch (captured training step + captured evaluation pass) on the benchmark graph with
the input pipeline on and off."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd
from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T

dev = torch.device("cuda", 0)
wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
low, deg, x_np, y_np, splits = wl["low"], wl["deg"], wl["x"], wl["y"], wl["splits"]
n = low.shape[0]
x, y = torch.from_numpy(x_np).to(dev), torch.from_numpy(y_np).to(dev)
w = T.row_weights(torch.from_numpy(splits[0]).to(dev), n, device=dev)
sets = tuple(torch.from_numpy(np.asarray(s)).to(dev) for s in splits)
for pipe in (False, None):
    ops = DD.make_sharded_operators(low, deg, dev)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
    step = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, pipeline_input=pipe)
    ev = T.EvalStep(model, x, ops, y, sets, loss_set=1, use_graph=True)
    for _ in range(10):
        step(); ev()
    res = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            step(); ev()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
    only = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize(); only.append((time.perf_counter() - t0) / 50 * 1e3)
    print(json.dumps({"pipeline": step.pipe is not None, "epoch_ms": [round(r, 4) for r in sorted(res)], "step_ms": [round(r, 4) for r in sorted(only)]}), flush=True)
