#!/usr/bin/env python3
"""Evaluation forward of the arXiv-year-shaped ACM-GCN+ (wide aggregate-first first layer) with and without the layer's P cache
(layers.GraphConvolution.eval_agg_cache): ms per eval-mode forward pass over the static features."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD  # noqa: E402

dev = torch.device("cuda:0")
wl = D.bench_workload(sys.argv[1] if len(sys.argv) > 1 else "arxiv-year")
n = wl["adj"].shape[0]
ops = DD.make_sharded_operators(wl["low"], wl["deg"], dev)
x = torch.from_numpy(wl["x"]).to(dev)
model = acm_gnn_amd.GCN(x.shape[1], 64, int(wl["y"].max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(dev).eval()
for cache in (False, True):
    for layer in model.gcns:
        layer.eval_agg_cache = cache
    with torch.no_grad():
        for _ in range(5):
            model(x, ops)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            model(x, ops)
        torch.cuda.synchronize()
    print(f"P cache {'on ' if cache else 'off'}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per evaluation forward", flush=True)
