#!/usr/bin/env python3
"""The zero-edit route, timed: the loop body of ACM-Geometric/train.py:119-137 (model.train(), zero_grad, forward on the
reference's sparse-COO adjacency TENSORS, F.log_softmax + nll_loss on the training rows, backward, torch.optim.AdamW)
on the twitch-shaped graph with the generator's RANDOM node ids -- what a user of the unmodified reference script gets
from the drop-in layer.  Three arms:
    relabel=auto   operators_for relabels by degree inside the operator (default for >= 32 768 nodes)
    relabel=off    tuning relabel=0: the operator keeps the random numbering
    pre-sorted     the dataset itself relabelled by degree beforehand (what bench.py does as data preparation)
Prints ms per eager step of the whole loop (dominated by torch's own launches: indexing + its sort-based backward,
F.dropout, ~80 AdamW kernels) and the time of the library's kernels inside it, which is what the relabelling changes.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, graph  # noqa: E402

DEV = torch.device("cuda:0")


def coo(m):
    m = m.tocoo()
    idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data.astype(np.float32)), m.shape).to(DEV)


def run(order, relabel, steps=30, fused_optimizer=False):
    acm_gnn_amd.tuning.apply(relabel={"auto": -1}.get(relabel, None) if relabel == "auto" else int(relabel))
    graph.clear_cache()
    wl = D.bench_workload("twitch-gamer", node_order=order)
    n = wl["adj"].shape[0]
    import scipy.sparse as sp
    low = coo(wl["low"])
    high = coo(sp.identity(n, dtype=np.float32, format="csr") - wl["low"])
    x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
    idx = torch.from_numpy(wl["splits"][0]).to(DEV)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    # (--fused-optimizer of the drop-in launcher: torch.optim.AdamW bound to the one-launch FusedAdamW for the script's run)
    opt = (acm_gnn_amd.FusedAdamW if fused_optimizer else torch.optim.AdamW)(model.parameters(), lr=0.05, weight_decay=1e-3)

    def step():                                      # train.py:119-137
        model.train()
        opt.zero_grad()
        out = F.log_softmax(model(x, low, high, None), dim=1)
        loss = F.nll_loss(out[idx], y[idx])
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        loss = step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    # the library's own kernels inside that loop (HIP events around every C-ABI call), and torch's share: the loop's
    # out[idx] / y[idx] indexing and its backward (sort-based index_put), F.dropout, ~80 AdamW launches
    from acm_gnn_amd import functional as AF
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(5):
        step()
    lib_us = sum(v[1] for v in timer.summary().values()) / 5 * 1e3
    AF.set_kernel_timer(None)
    ops = graph.operators_for(low, high, None)
    return {"node_order": order, "relabel": relabel, "optimizer": "FusedAdamW (--fused-optimizer)" if fused_optimizer else "torch.optim.AdamW",
            "relabelled_in_operator": ops.perm is not None,
            "eager_ms_per_step": round(ms, 3), "library_kernels_us_per_step": round(lib_us, 1), "loss": float(loss)}


if __name__ == "__main__":
    for order, relabel in (("random", "auto"), ("random", "0"), ("degree", "0")):
        print(json.dumps(run(order, relabel)), flush=True)
    print(json.dumps(run("random", "auto", fused_optimizer=True)), flush=True)
