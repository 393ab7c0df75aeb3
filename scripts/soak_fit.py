#!/usr/bin/env python3
"""The reference's whole training loop at the benchmark's size: train.fit(use_graph=True) -- captured training step +
captured evaluation pass + model selection -- for N epochs on the twitch-shaped graph; wall time per epoch, loss at both
ends, finiteness of every parameter.  (Synthetic labels: the accuracies are not a result, the loop's health is.)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, train as T  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    n = wl["adj"].shape[0]
    ops = DD.make_sharded_operators(wl["low"], wl["deg"], DEV)
    x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
    tr, va, te = (torch.from_numpy(s).to(DEV) for s in wl["splits"])
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(x.shape[1], 64, int(wl["y"].max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False,
                            attn_layernorm=True).to(DEV)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, hist = T.fit(model, opt, x, ops, y, tr, va, te, epochs=epochs, rule="max_val_acc", use_graph=True, fused_dropout=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    h = np.array(hist)
    print(f"{epochs} epochs in {dt:.3f} s including the two captures = {dt / epochs * 1e3:.3f} ms per epoch "
          f"(train step + evaluation pass + selection on the host)")
    print(f"train loss {h[0, 0]:.4f} -> {h[-1, 0]:.4f}; validation loss {h[0, 4]:.4f} -> {h[-1, 4]:.4f}; "
          f"selected test accuracy {acc:.4f}; every parameter finite: {finite}")
    assert finite and h[-1, 0] < h[0, 0]


if __name__ == "__main__":
    main()
