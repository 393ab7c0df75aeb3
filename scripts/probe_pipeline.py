#!/usr/bin/env python3
"""The captured training step of the headline configuration with and without the input pipeline (functional.InputPipeline:
the next step's P = A_low dropout(x) gathered inside the first layer's backward).  Prints the loss trajectories of both
(same seeds: they differ only by the summation order of the gather) and the replayed step time."""
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


def build(pipeline, wl, use_graph=True):
    low, deg = wl["low"], wl["deg"]
    n = low.shape[0]
    ops = DD.make_sharded_operators(low, deg, DEV)
    x = torch.from_numpy(np.ascontiguousarray(wl["x"])).to(DEV)
    y = torch.from_numpy(np.ascontiguousarray(wl["y"])).to(DEV)
    tr = torch.from_numpy(np.asarray(wl["splits"][0])).to(DEV)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(x.shape[1], 64, int(wl["y"].max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    model.dropout_state = AF.DropoutState(DEV, seed=1234)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    w = T.row_weights(tr, n, device=DEV)
    step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph, pipeline_input=pipeline)
    return step, model


def main():
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    res = {}
    for pipeline in (False, None):
        step, model = build(pipeline, wl)
        name = "pipelined" if step.pipe is not None else "plain"
        losses = [float(step().item()) for _ in range(12)]
        torch.cuda.synchronize()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        res[name] = losses
        print(f"{name:9s}: {ms:.4f} ms/step  losses " + " ".join(f"{v:.6f}" for v in losses), flush=True)
        del step, model
    if len(res) == 2:
        a, b = np.array(res["plain"]), np.array(res["pipelined"])
        print(f"max |loss difference| over 12 steps: {np.abs(a - b).max():.3e}")
    # per-kernel times of the pipelined eager step
    step, model = build(None, wl, use_graph=False)
    for _ in range(5):
        step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(5):
        step()
    AF.set_kernel_timer(None)
    for k, (cnt, tot) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:40s} {tot / cnt * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
