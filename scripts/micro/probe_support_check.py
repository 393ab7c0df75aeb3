#!/usr/bin/env python3
"""Pieces of layers.GraphConvolution._csr_input's per-step support check on the Penn94-shaped dense input, timed."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acm_gnn_amd import SparseFeatures  # noqa: E402

dev = torch.device("cuda:0")
n, f = 41554, 4814
x = torch.zeros(n, f, device=dev)
idx = torch.randint(0, f, (n, 5), device=dev)
x.scatter_(1, idx, 1.0)
tw = SparseFeatures.auto(x)
xd = F.dropout(x, 0.5)


def t(name, fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:42s} {(time.perf_counter() - a) / reps * 1e3:8.3f} ms", flush=True)
    return r


t("F.dropout(x)", lambda: F.dropout(x, 0.5))
t("torch.count_nonzero(x)", lambda: torch.count_nonzero(xd))
t("(x != 0).sum()", lambda: (xd != 0).sum())
t("x.view(int32).ne(0).sum(dtype=int32)", lambda: xd.view(torch.int32).ne(0).sum(dtype=torch.int32))
t("x.abs().sum()", lambda: xd.abs().sum())
t("x.sum()", lambda: xd.sum())
t("torch.linalg.vector_norm(x, 0)", lambda: torch.linalg.vector_norm(xd, 0))
t("x.amax()", lambda: xd.amax())
flat = tw.twin_of_masked(xd)._flat_index
t("index_select(flat)", lambda: xd.reshape(-1).index_select(0, flat))
t("twin_of_masked (whole, with the host copy)", lambda: tw.twin_of_masked(xd))
