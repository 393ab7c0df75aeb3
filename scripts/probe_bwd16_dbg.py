#!/usr/bin/env python3
"""Run ON THE GPU BOX: time the row-local backward of the benchmark's first layer under ACM_BWD16_* switches."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import data as D, distributed as DD, functional as AF
from acm_gnn_amd.layers import GraphConvolution

dev = torch.device("cuda", 0)
wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
low, deg, x_np = wl["low"], wl["deg"], wl["x"]
n = low.shape[0]
ops = DD.make_sharded_operators(low, deg, dev)
x8 = torch.zeros(n, 8, device=dev)
x8[:, :7] = torch.from_numpy(x_np).to(dev)
mt = sys.argv[1] if len(sys.argv) > 1 else "acmgcnp"
torch.manual_seed(1)
layer = GraphConvolution(7, 64, n, mt).to(dev)
layer.train()
go = torch.randn(n, 64, device=dev) * 1e-3
state = AF.DropoutState(dev, seed=5)
for cfg in sys.argv[2:] or [""]:
    keys = []
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        keys.append(k)
    for it in range(2):
        timer = AF.KernelTimer(only="conv_agg_bwd")
        AF.set_kernel_timer(timer)
        for _ in range(20):
            layer.zero_grad(set_to_none=True)
            out = layer(x8, ops, post_relu=True, post_drop=(0.1, 1, state))
            out.backward(go)
        AF.set_kernel_timer(None)
        s = timer.summary()
    print(json.dumps({"model": mt, "env": cfg, "us": {k: round(v[1] / v[0] * 1e3, 1) for k, v in s.items()}}), flush=True)
    for k in keys:
        os.environ.pop(k, None)
