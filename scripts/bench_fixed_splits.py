#!/usr/bin/env python3
"""The reference's fixed-split protocol on the small graphs (ACM-Pytorch/train.py:49-139: ten splits, a fresh model each, an epoch =
training step + evaluation pass + model selection) on this package's own loop: train.fit per split one after the other against
train.fit_concurrent (two splits side by side on their own streams).  Prints ms per split-epoch for both."""
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
from acm_gnn_amd import data as D, train as T  # noqa: E402
from acm_gnn_amd.distributed import make_sharded_operators  # noqa: E402

DEV = torch.device("cuda:0")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main(name, epochs=200, splits=10):
    g = np.load(os.path.join(GOLDEN, f"graph_{name}.npz"))
    n = int(g["n"])
    a = sp.csr_matrix((np.ones(len(g["adj_un_indices"]), np.float32), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    vals = g["feat_vals"] if "feat_vals" in g.files else np.ones(len(g["feat_indices"]), np.float32)
    xm = sp.csr_matrix((vals, g["feat_indices"], g["feat_indptr"]), shape=(n, int(g["feat_dim"])))
    low, deg = D.build_filters(a)
    four = name != "cora"
    ops = make_sharded_operators(low, deg, DEV, with_structure=four)
    xs = acm_gnn_amd.SparseFeatures.from_scipy(xm, DEV)
    y = torch.from_numpy(np.asarray(g["labels"], np.int64)).to(DEV)
    classes = int(y.max()) + 1

    def make(k):
        torch.manual_seed(k)
        m = acm_gnn_amd.GCN(xm.shape[1], 64, classes, 1, n, 0.6, "acmgcnp" if four else "acmgcn", int(four), attn_layernorm=False).to(DEV)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(k)).to(DEV)
        return m, acm_gnn_amd.FusedAdam(m.parameters(), lr=0.01, weight_decay=5e-5), idx[: int(.6 * n)], idx[int(.6 * n): int(.8 * n)], idx[int(.8 * n):]

    out = {"graph": name, "splits": splits, "epochs": epochs}
    for label, streams in (("one_after_the_other", 1), ("two_side_by_side", 2), ("three_side_by_side", 3)):
        runs = [make(k) for k in range(splits)]
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = T.fit_concurrent(runs, xs, ops, y, epochs, rule="min_val_loss", early_stopping=0, streams=streams)
        torch.cuda.synchronize()
        total = time.perf_counter() - t
        out[f"{label}_ms_per_split_epoch"] = round(total / (splits * epochs) * 1e3, 4)
        out[f"{label}_total_s"] = round(total, 3)
        out[f"{label}_mean_selected"] = round(float(np.mean([r[0] for r in res])), 5)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    for name in sys.argv[1:] or ("cora", "squirrel"):
        main(name)
