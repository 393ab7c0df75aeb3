#!/usr/bin/env python3
"""The zero-edit route on a wide one-hot input, timed: the reference's own model wiring (ACM-Geometric/models.py:52-76 --
F.dropout on the DENSE features, first layer, relu, dropout, output layer) around the drop-in GraphConvolution, in the loop
of ACM-Geometric/train.py:119-140 (one training step + one evaluation pass per epoch) on the Penn94-shaped workload
(41 554 nodes, 4 814 one-hot feature columns handed over dense as dataset.graph["node_feat"] is).
Two arms: tuning csr_features = 0 (the dense projection) and the default 256 (layers.GraphConvolution._csr_input: the
evaluation pass makes the CSR twin of the loader's tensor, every training pass takes its structure with the dropped copy's
values after the support check).  Prints ms per epoch, per training step and per evaluation pass (eager, wall clock with a
device synchronisation around the timed loop) and the library's kernel time inside them.

    python scripts/bench_dropin_features.py [dataset ...]       # default: penn94
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, functional as AF, graph, tuning  # noqa: E402

DEV = torch.device("cuda:0")


class ReferenceWiring(nn.Module):
    """What ACM-Geometric/models.py's GCN does around its two layers for 'acmgcnp' (restated, not imported: the reference
    checkout does not exist on the GPU box)."""

    def __init__(self, f_in, hidden, classes, n, p):
        super().__init__()
        self.gcns = nn.ModuleList([acm_gnn_amd.GraphConvolution(f_in, hidden, n, "acmgcnp"),
                                   acm_gnn_amd.GraphConvolution(hidden, classes, n, "acmgcnp", output_layer=1)])
        self.p = p

    def forward(self, x, low, high):
        x = F.dropout(x, self.p, training=self.training)
        h = F.dropout(F.relu(self.gcns[0](x, low, high, None)), self.p, training=self.training)
        return self.gcns[1](h, low, high, None)


def coo(m):
    m = m.tocoo()
    idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data.astype(np.float32)), m.shape).to(DEV)


def run(dataset, key, epochs=20):
    tuning.apply(csr_features=key)
    graph.clear_cache()
    adj, x_np, y_np, (tr, va, _), n = D.synthetic_dataset(dataset)
    x_np = D.row_normalize_features(x_np)
    low_sp, _ = D.build_filters(adj)
    low = coo(low_sp)
    high = coo(sp.identity(n, dtype=np.float32, format="csr") - low_sp)
    x, y = torch.from_numpy(x_np).to(DEV), torch.from_numpy(y_np.astype(np.int64)).to(DEV)
    tr_i, va_i = torch.from_numpy(tr).to(DEV), torch.from_numpy(va).to(DEV)
    torch.manual_seed(0)
    model = ReferenceWiring(x.shape[1], 64, int(y_np.max()) + 1, n, 0.5).to(DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    seen = {"train": set(), "eval": set()}
    conv = AF.acm_conv

    def spy(inp, *a, **k):
        if inp.shape[1] == x.shape[1]:
            seen["train" if model.training else "eval"].add(type(inp).__name__)
        return conv(inp, *a, **k)
    acm_gnn_amd.layers.AF.acm_conv = spy

    def train_step():                                   # train.py:119-137
        model.train()
        opt.zero_grad()
        out = F.log_softmax(model(x, low, high), dim=1)
        loss = F.nll_loss(out[tr_i], y[tr_i])
        loss.backward()
        opt.step()
        return loss

    @torch.no_grad()
    def eval_pass():                                    # train.py:138-140, data_utils.py:153-168
        model.eval()
        out = model(x, low, high)
        return (out.argmax(1)[va_i] == y[va_i]).float().mean()

    def timed(fn, reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3, r

    try:
        for _ in range(3):
            train_step(), eval_pass()
        for k in seen:
            seen[k].clear()
        ep_ms, _ = timed(lambda: (train_step(), eval_pass()), epochs)
        tr_ms, loss = timed(train_step, epochs)
        ev_ms, acc = timed(eval_pass, epochs)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        for _ in range(3):
            train_step()
        lib_tr = sum(v[1] for v in timer.summary().values()) / 3 * 1e3
        AF.set_kernel_timer(None)
    finally:
        acm_gnn_amd.layers.AF.acm_conv = conv
    return {"dataset": dataset, "n": n, "f_in": int(x.shape[1]), "density": round(float((x_np != 0).mean()), 5),
            "csr_features": key, "first_layer_input": {k: sorted(v) for k, v in seen.items()},
            "epoch_ms": round(ep_ms, 3), "train_step_ms": round(tr_ms, 3), "eval_pass_ms": round(ev_ms, 3),
            "library_kernels_us_per_train_step": round(lib_tr, 1), "loss": float(loss), "val_acc": float(acc)}


if __name__ == "__main__":
    for ds in (sys.argv[1:] or ["penn94"]):
        for key in (0, 256):
            print(json.dumps(run(ds, key)), flush=True)
