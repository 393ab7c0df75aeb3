#!/usr/bin/env python3
"""ms/step of one full-batch train step (fwd + fused NLL + bwd + Adam) for the other BASELINE.json
configs, eager and hipGraph-replayed, with the per-kernel HIP-event breakdown.

    python scripts/bench_configs.py [names...]      # default: all
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T  # noqa: E402

DEV = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")


def real_graph(name):
    g = np.load(os.path.join(GOLD, f"graph_{name}.npz"))
    n = int(g["n"])
    a = sp.csr_matrix((np.ones(len(g["adj_un_indices"]), np.float32), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    return a, g


CONFIGS = {
    # name: (builder, model kwargs)
    "cora/acmgcn": dict(graph="cora", f_in=1433, classes=7, method="acmgcn", s=0, variant=0, dropout=0.6),
    "squirrel/acmgcnp+A": dict(graph="squirrel", f_in=2089, classes=5, method="acmgcnp", s=1, variant=0, dropout=0.6),
    "chameleon/acmgcnp+A": dict(graph="chameleon", f_in=2325, classes=5, method="acmgcnp", s=1, variant=0, dropout=0.7),
    "twitch/acmgcn": dict(graph="syn:twitch-gamer", method="acmgcn", s=0, variant=0, dropout=0.1),
    "twitch/acmgcnp+A": dict(graph="syn:twitch-gamer", method="acmgcnp", s=1, variant=0, dropout=0.0),
    "twitch/acmiigcnp": dict(graph="syn:twitch-gamer", method="acmgcnp", s=0, variant=1, dropout=0.1),
    # ACM-GCN++ (the residual Linear on the raw features; ACM-Geometric/sh/run_all_settings.sh:13-14, models.py:27,55-56,73)
    "twitch/acmgcnpp": dict(graph="syn:twitch-gamer", method="acmgcnpp", s=0, variant=0, dropout=0.1),
    "twitch/acmgcnpp+A": dict(graph="syn:twitch-gamer", method="acmgcnpp", s=1, variant=0, dropout=0.0),
    "twitch/acmiigcnpp": dict(graph="syn:twitch-gamer", method="acmgcnpp", s=0, variant=1, dropout=0.1),
    "twitch/acmiigcnp+A": dict(graph="syn:twitch-gamer", method="acmgcnp", s=1, variant=1, dropout=0.0),
    # the fabric-bound cells with bf16 storage of the gathered operands (opt-in, BASELINE config 3; fp32 sums)
    "twitch/acmgcnp+A/bf16": dict(graph="syn:twitch-gamer", method="acmgcnp", s=1, variant=0, dropout=0.0, gather_dtype="bf16"),
    "twitch/acmiigcnp/bf16": dict(graph="syn:twitch-gamer", method="acmgcnp", s=0, variant=1, dropout=0.1, gather_dtype="bf16"),
    "twitch/acmiigcnp+A/bf16": dict(graph="syn:twitch-gamer", method="acmgcnp", s=1, variant=1, dropout=0.0, gather_dtype="bf16"),
    "arxiv-year/acmgcnp": dict(graph="syn:arxiv-year", method="acmgcnp", s=0, variant=0, dropout=0.1),
    "penn94/acmgcnp": dict(graph="syn:penn94", method="acmgcnp", s=0, variant=0, dropout=0.1),
    # the same wide-feature configs with CSR features (sparse-feature projection)
    "cora/acmgcn/csrX": dict(graph="cora", f_in=1433, classes=7, method="acmgcn", s=0, variant=0, dropout=0.6, sparse=1),
    "squirrel/acmgcnp+A/csrX": dict(graph="squirrel", f_in=2089, classes=5, method="acmgcnp", s=1, variant=0, dropout=0.6, sparse=1),
    "penn94/acmgcnp/csrX": dict(graph="syn:penn94", method="acmgcnp", s=0, variant=0, dropout=0.1, sparse=1),
    # ... and handed over DENSE, as the reference's loaders do: the model makes the CSR twin itself (tuning csr_features)
    "cora/acmgcn/auto": dict(graph="cora", f_in=1433, classes=7, method="acmgcn", s=0, variant=0, dropout=0.6, sparse="auto"),
    "squirrel/acmgcnp+A/auto": dict(graph="squirrel", f_in=2089, classes=5, method="acmgcnp", s=1, variant=0, dropout=0.6, sparse="auto"),
    "chameleon/acmgcnp+A/auto": dict(graph="chameleon", f_in=2325, classes=5, method="acmgcnp", s=1, variant=0, dropout=0.7, sparse="auto"),
    "penn94/acmgcnp/auto": dict(graph="syn:penn94", method="acmgcnp", s=0, variant=0, dropout=0.1, sparse="auto"),
    "penn94/acmsgc-3hop/auto": dict(graph="syn:penn94", method="acmsgc", s=0, variant=0, dropout=0.1, sparse="auto", hops=3),
    # BASELINE config 5: ACM-SGC 3-hop (one linear ACM layer, the low channel through A_low three times)
    "arxiv-year/acmsgc-3hop": dict(graph="syn:arxiv-year", method="acmsgc", s=0, variant=0, dropout=0.1, hops=3),
    "penn94/acmsgc-3hop/csrX": dict(graph="syn:penn94", method="acmsgc", s=0, variant=0, dropout=0.1, sparse=1, hops=3),
}


def run(name, cfg, steps=20):
    if cfg["graph"].startswith("syn:"):
        ds = cfg["graph"][4:]
        adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset(ds)
        perm = D.degree_order(adj)
        adj, x_np, y_np, (tr, _, _) = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
        f_in, classes = x_np.shape[1], int(y_np.max()) + 1
    else:
        adj, g = real_graph(cfg["graph"])
        n, f_in, classes = adj.shape[0], cfg["f_in"], cfg["classes"]
        rng = np.random.default_rng(0)
        x_np = (rng.random((n, f_in)) < 0.01).astype(np.float32)       # sparse binary bag-of-words-like features
        y_np = rng.integers(0, classes, n)
        tr = np.sort(rng.permutation(n)[: int(0.48 * n)])
    if not (cfg["method"] in ("acmgcnp", "acmgcnpp") and cfg["s"]):
        x_np = D.row_normalize_features(x_np)
    low, deg = D.build_filters(adj)
    ops = DD.make_sharded_operators(low, deg, DEV, with_structure=bool(cfg["s"]))
    ops.hops = cfg.get("hops", 1)
    x, y = torch.from_numpy(x_np).to(DEV), torch.from_numpy(y_np.astype(np.int64)).to(DEV)
    from acm_gnn_amd import tuning
    tuning.apply(csr_features=256 if cfg.get("sparse") == "auto" else 0)      # the dense rows measure the dense projection
    if cfg.get("sparse") == 1:
        x = acm_gnn_amd.SparseFeatures.from_scipy(sp.csr_matrix(x_np), DEV)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(f_in, 64, classes, 2, n, cfg["dropout"], cfg["method"], cfg["s"],
                            variant=bool(cfg["variant"]), attn_layernorm=True, gather_dtype=cfg.get("gather_dtype")).to(DEV)
    opt = acm_gnn_amd.FusedAdam(model.parameters(), lr=0.01, weight_decay=1e-4)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    step = T.TrainStep(model, opt, x, ops, y, w)
    for _ in range(5):
        step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(3):
        step()
    kern = {k: round(v[1] / v[0] * 1e3, 1) for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])}
    AF.set_kernel_timer(None)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / steps * 1e3
    gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True)
    for _ in range(3):
        gstep()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        gstep()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / steps * 1e3
    return dict(config=name, n=n, nnz_low=int(low.nnz), f_in=f_in, eager_ms=round(eager, 3), graph_ms=round(graph, 3),
                kernel_us=kern)


if __name__ == "__main__":
    names = sys.argv[1:] or list(CONFIGS)
    for nm in names:
        print(json.dumps(run(nm, CONFIGS[nm])), flush=True)
