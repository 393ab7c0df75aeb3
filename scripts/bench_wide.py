#!/usr/bin/env python3
"""A/B of the wide (F-column) gather kernels on the LINKX-shaped graphs: per-kernel HIP-event times and the
hipGraph-replayed step for the configurations that cannot use the aggregate-first rewrite (ACMII: ReLU between
projection and filter; the structure channel; hidden -> hidden layers).

    python scripts/bench_wide.py [--modes scalar,vec] [--configs twitch/acmii,...]

mode = acm_tuning_t.wide_form:  scalar: 1 (dword-per-lane kernel),
default: 0, the library's choice (vector form for single-channel products), vec: 2 (dwordx4 rows, four
neighbours per instruction, everywhere).
"""
import argparse
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
CONFIGS = {
    "twitch/acmii": dict(ds="twitch-gamer", method="acmgcnp", s=0, variant=1, dropout=0.1),
    "twitch/acm+A": dict(ds="twitch-gamer", method="acmgcnp", s=1, variant=0, dropout=0.1),
    "twitch/acmii+A": dict(ds="twitch-gamer", method="acmgcnp", s=1, variant=1, dropout=0.1),
    "arxiv/acm": dict(ds="arxiv-year", method="acmgcnp", s=0, variant=0, dropout=0.1),
    "penn94/acm/csrX": dict(ds="penn94", method="acmgcnp", s=0, variant=0, dropout=0.1, sparse=1),
}
MODES = {"scalar": 1, "default": 0, "vec": 2, "pair": 3}
_WL = {}


def workload(ds, normalize):
    key = (ds, normalize)
    if key not in _WL:
        _WL[key] = D.bench_workload(ds, seed=0, node_order="degree", normalize_features=normalize)
    return _WL[key]


def run(name, cfg, mode, steps=20):
    acm_gnn_amd.tuning.apply(wide_form=MODES[mode])
    wl = workload(cfg["ds"], not cfg["s"])
    n = wl["adj"].shape[0]
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV, with_structure=bool(cfg["s"]))
    x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
    if cfg.get("sparse"):
        x = acm_gnn_amd.SparseFeatures.from_scipy(sp.csr_matrix(wl["x"]), DEV)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(wl["x"].shape[1], 64, int(wl["y"].max()) + 1, 2, n, cfg["dropout"], cfg["method"], cfg["s"],
                            variant=bool(cfg["variant"]), attn_layernorm=True).to(DEV)
    opt = acm_gnn_amd.FusedAdam(model.parameters(), lr=0.01, weight_decay=1e-4)
    w = T.row_weights(torch.from_numpy(wl["splits"][0]).to(DEV), n)
    step = T.TrainStep(model, opt, x, ops, y, w)
    for _ in range(5):
        loss = step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(5):
        step()
    kern = {k: round(v[1] / v[0] * 1e3, 1) for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])}
    AF.set_kernel_timer(None)
    gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True)
    for _ in range(3):
        gstep()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        loss = gstep()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / steps * 1e3
    return dict(config=name, mode=mode, nnz_low=int(wl["low"].nnz), graph_ms=round(graph, 3), loss=float(loss),
                kernel_us=kern)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="scalar,vec")
    ap.add_argument("--configs", default=",".join(CONFIGS))
    a = ap.parse_args()
    for nm in a.configs.split(","):
        for mode in a.modes.split(","):
            print(json.dumps(run(nm, CONFIGS[nm], mode)), flush=True)
