#!/usr/bin/env python3
"""The kernels of one evaluation pass (train.EvalStep) on the twitch-shaped graph and on the small graphs: per-call HIP-event times
(functional.KernelTimer) and the captured pass's wall time."""
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


def main():
    wl = D.bench_workload("twitch-gamer", node_order="degree")
    low, deg = wl["low"], wl["deg"]
    ops = DD.make_sharded_operators(low, deg, DEV)
    x, y = torch.from_numpy(wl["x"]).to(DEV), torch.from_numpy(wl["y"]).to(DEV)
    sets = tuple(torch.from_numpy(s).to(DEV) for s in wl["splits"])
    n = x.shape[0]
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, attn_layernorm=True).to(DEV)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
    step = T.TrainStep(model, opt, x, ops, y, T.row_weights(sets[0], n), use_graph=True)
    for use_graph in (False, True):
        ev = T.EvalStep(model, x, ops, y, sets, use_graph=use_graph)
        for _ in range(5):
            step(), ev()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            ev()
        torch.cuda.synchronize()
        print(json.dumps({"eval_pass_ms": round((time.perf_counter() - t) / 50 * 1e3, 4), "captured": use_graph}))
        if not use_graph:
            timer = AF.KernelTimer()
            AF.set_kernel_timer(timer)
            for _ in range(5):
                ev()
            AF.set_kernel_timer(None)
            print(json.dumps({k: round(v[1] / 5 * 1e3, 1) for k, v in timer.summary().items()}))
    t = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    print(json.dumps({"train_step_ms": round((time.perf_counter() - t) / 50 * 1e3, 4)}))


if __name__ == "__main__":
    main()
