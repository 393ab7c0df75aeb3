#!/usr/bin/env python3
"""Captured training step with the next step's table drawn on a side stream (ACM_PIPE_SIDE_STREAM=1) against the default
(between forward and backward on the main stream): ms per step and equality of the losses."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, train as T  # noqa: E402

DEV = torch.device("cuda:0")
wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
n = wl["adj"].shape[0]
x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
w = T.row_weights(torch.from_numpy(wl["splits"][0]).to(DEV), n, device=DEV)
ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV)
for side in ("0", "1", "0", "1"):
    os.environ["ACM_PIPE_SIDE_STREAM"] = side
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(x.shape[1], 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    g = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=True)
    assert g.pipe is not None and (g.pipe._side is not None) == (side == "1")
    losses = [float(g()) for _ in range(10)]
    wins = []
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            g()
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t) / 50 * 1e3)
    print(json.dumps({"side_stream": side, "ms": round(sorted(wins)[2], 4), "losses": losses[-3:]}), flush=True)
