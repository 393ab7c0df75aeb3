#!/usr/bin/env python3
"""Do S independent small-graph training runs (the reference's ten fixed splits: ACM-Pytorch/train.py:49-139 trains them one after
the other) overlap on the device when each has its own stream and its own captured step?  S models / optimizers / TrainSteps on
the Cora / Squirrel structure, every step() replayed on its own stream, one synchronisation per round; ms per ROUND (S steps)."""
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


def main(name, s_list):
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
    for S in s_list:
        steps, streams = [], []
        for k in range(S):
            torch.manual_seed(k)
            m = acm_gnn_amd.GCN(xm.shape[1], 64, classes, 1, n, 0.6, "acmgcnp" if four else "acmgcn", int(four), attn_layernorm=False).to(DEV)
            o = acm_gnn_amd.FusedAdam(m.parameters(), lr=0.01, weight_decay=5e-5)
            tr = torch.randperm(n, generator=torch.Generator().manual_seed(k))[: n // 2].to(DEV)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                step = T.TrainStep(m, o, xs, ops, y, T.row_weights(tr, n), use_graph=True)
            assert step.small is not None
            steps.append(step), streams.append(st)
        torch.cuda.synchronize()

        def round_():
            for step, st in zip(steps, streams):
                with torch.cuda.stream(st):
                    step()
        for _ in range(10):
            round_()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(50):
                round_()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 50 * 1e3)
        print(json.dumps({"graph": name, "concurrent_runs": S, "ms_per_round": round(best, 4), "ms_per_step": round(best / S, 4)}), flush=True)
        del steps, streams


if __name__ == "__main__":
    for name in ("cora", "squirrel"):
        main(name, [1, 2, 5, 10])
