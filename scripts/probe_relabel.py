#!/usr/bin/env python3
"""Why is the captured step on random ids + in-operator relabelling faster than on the pre-sorted dataset?  Times both
several times in one process, alternating, with the per-kernel breakdown of each."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T  # noqa: E402

DEV = torch.device("cuda:0")


def setup(order, relabel):
    wl = D.bench_workload("twitch-gamer", seed=0, node_order=order)
    n = wl["adj"].shape[0]
    x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
    w = T.row_weights(torch.from_numpy(wl["splits"][0]).to(DEV), n, device=DEV)
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV, relabel=relabel)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(x.shape[1], 64, int(wl["y"].max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    return model, opt, x, ops, y, w


def kernels(args):
    step = T.TrainStep(*args, use_graph=False, fused_dropout=True)
    for _ in range(5):
        step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(5):
        step()
    AF.set_kernel_timer(None)
    return {k: round(v[1] / v[0] * 1e3, 1) for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])}


def timed(args, windows=5, steps=50):
    g = T.TrainStep(*args, use_graph=True, fused_dropout=True)
    for _ in range(10):
        g()
    out = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            g()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / steps * 1e3)
    return round(sorted(out)[windows // 2], 4)


if __name__ == "__main__":
    cases = {"degree-presorted": ("degree", False), "random+relabel": ("random", True), "degree+forced-relabel-noop": ("degree", True)}
    sets = {k: setup(*v) for k, v in cases.items()}
    for rnd in range(2):
        for k, a in sets.items():
            print(json.dumps({"case": k, "round": rnd, "graph_ms": timed(a), "perm": a[3].perm is not None}), flush=True)
    for k, a in sets.items():
        print(json.dumps({"case": k, "kernel_us": kernels(a)}), flush=True)
    a, b = sets["degree-presorted"][3], sets["random+relabel"][3]
    ia, ib = a.low.indptr_t if hasattr(a.low, "indptr_t") else None, None
    print(json.dumps({"chunk": [a.low.chunk, b.low.chunk], "nnz": [a.low.nnz, b.low.nnz], "items": [getattr(a.low, "n_items", None), getattr(b.low, "n_items", None)]}))
