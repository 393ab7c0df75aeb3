#!/usr/bin/env python3
"""ms per full-batch training step (fwd + fused NLL + bwd + fused AdamW; eager and hipGraph replay) of the 2-layer ACM-GCN+
on graphs of the LARGE shapes of the reference's grid (ACM-Geometric/sh/run_all_settings.sh:2): pokec (1.63 M nodes /
30.6 M edges, 65 features) and snap-patents (2.92 M nodes / 13.98 M directed edges, 269 features) -- ten to seventeen times
the rows of the benchmark graph, drawn on the GPU by the generator of tests/test_gpu_scale.py (which holds the parity of the
same shapes against the oracle).  One JSON line per graph: ms/step, stored edges per second, peak device memory, the
per-kernel HIP-event breakdown.

    python scripts/bench_scale.py [pokec] [snap-patents] [pokec/bf16] [snap-patents/bf16]

``<graph>/bf16``: the opt-in bf16 storage of the gathered operands (GCN(gather_dtype="bf16"): BASELINE config 3's tolerance;
fp32 sums) -- at these sizes the wide gathers are HBM-bandwidth kernels (profiles/r04_pokec_pmc.txt), so bytes per edge are
what moves them.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, functional as AF, train as T  # noqa: E402
from acm_gnn_amd.graph import CsrGraph, FilterOperators, as_implicit, relabel_by_degree  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = {"pokec": (1_632_803, 30_622_564, 14_854, 65, 2, False),
          "snap-patents": (2_923_922, 13_975_788, 800, 269, 5, True)}


def run(name, steps=20):
    from test_gpu_scale import _powerlaw_graph_on_gpu
    name, _, dt = name.partition("/")
    n, n_edges, max_deg, f_in, n_cls, directed = SHAPES[name]
    t0 = time.time()
    adj = _powerlaw_graph_on_gpu(n, n_edges, max_deg, seed=3, directed=directed)
    low, deg = D.build_filters(adj)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(D.row_normalize_features(np.abs(rng.standard_normal((n, f_in))).astype(np.float32))).to(DEV)
    y = torch.from_numpy(rng.integers(0, n_cls, n).astype(np.int64)).to(DEV)
    tr = torch.from_numpy(np.sort(rng.permutation(n)[: n // 2])).to(DEV)
    ops = relabel_by_degree(as_implicit(FilterOperators(CsrGraph.from_scipy(low, DEV))))
    prep = time.time() - t0
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(f_in, 64, n_cls, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True,
                            gather_dtype=dt or None).to(DEV)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    w = T.row_weights(tr, n)
    torch.cuda.reset_peak_memory_stats()
    step = T.TrainStep(model, opt, x, ops, y, w)
    for _ in range(3):
        step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(3):
        step()
    kern = {k: round(v[1] / v[0] * 1e3, 1) for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])}
    AF.set_kernel_timer(None)

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            loss = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps * 1e3, float(loss)

    eager, _ = timed(step)
    gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True)
    for _ in range(3):
        gstep()
    graph, loss = timed(gstep)
    ms = min(eager, graph)
    return {"graph": name, "gather_dtype": dt or "fp32", "nodes": n, "nnz_A_low": int(low.nnz), "f_in": f_in, "directed": directed,
            "operator": "pattern-only" if ops.implicit else "explicit + transposed CSR",
            "eager_ms": round(eager, 3), "graph_ms": round(graph, 3),
            "stored_edges_per_s": round(low.nnz / (ms * 1e-3), 1), "loss": loss, "prep_s": round(prep, 1),
            "peak_device_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "kernel_us": kern}


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(SHAPES)):
        print(json.dumps(run(nm)), flush=True)
