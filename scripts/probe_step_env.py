#!/usr/bin/env python3
"""Run ON THE GPU BOX: the benchmark's training step under a list of tuning settings (acm_gnn_amd.tuning: the items of
ACM_TUNING), per-kernel HIP-event times of the eager step and the replayed-graph step time.

    python scripts/probe_step_env.py "" "rows16=5" "rows16=6,pipeline=0" "lever:no_refill" "lever:spg5" ...
An empty string is the default configuration.  ``lever:NAME`` items are same-box A/B switches of this script alone (not
tuning switches): no_refill = the input pipeline's table drawn by its own acm_dropout launch instead of inside the forward's
row-local kernel; spgK = K optimizer steps per captured graph; host_flush = the deferred reductions flushed by their own
launch instead of inside the optimizer's (TrainStep(flush_in_optimizer=False))."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, functional as AF, optim as O, train as T  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    low, deg, x_np, y_np, (tr, va, te) = wl["low"], wl["deg"], wl["x"], wl["y"], wl["splits"]
    n = low.shape[0]
    x = torch.from_numpy(x_np).to(dev)
    y = torch.from_numpy(y_np).to(dev)
    w = T.row_weights(torch.from_numpy(tr).to(dev), n, device=dev)
    configs = sys.argv[1:] or [""]
    for cfg in configs:
        levers = [it[6:] for it in cfg.split(",") if it.startswith("lever:")]
        kern, host = acm_gnn_amd.tuning.parse(",".join(it for it in cfg.split(",") if not it.startswith("lever:")))
        acm_gnn_amd.tuning.reset()
        acm_gnn_amd.tuning.apply(**kern, **host)
        real_refill = AF.InputPipeline.refill_spec
        if "no_refill" in levers:
            AF.InputPipeline.refill_spec = lambda self, p: False
        spg = next((int(lv[3:]) for lv in levers if lv.startswith("spg")), 1)
        fio = "host_flush" not in levers
        ops = DD.make_sharded_operators(low, deg, dev)
        torch.manual_seed(0)
        model = acm_gnn_amd.GCN(x.shape[1], 64, int(y_np.max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
        opt = O.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
        step = T.TrainStep(model, opt, x, ops, y, w, flush_in_optimizer=fio)
        for _ in range(5):
            step()
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        for _ in range(20):
            step()
        AF.set_kernel_timer(None)
        ks = {k: round(v[1] / v[0] * 1e3, 1) for k, v in timer.summary().items()}
        # the captured step
        model2 = acm_gnn_amd.GCN(x.shape[1], 64, int(y_np.max()) + 1, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
        opt2 = O.FusedAdamW(model2.parameters(), lr=0.05, weight_decay=1e-3)
        gstep = T.TrainStep(model2, opt2, x, ops, y, w, use_graph=True, steps_per_graph=spg, flush_in_optimizer=fio)
        for _ in range(10):
            gstep()
        best = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50 // spg):
                gstep()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / (50 // spg * spg) * 1e3)
        print(json.dumps({"env": cfg, "graph_ms": [round(b, 4) for b in sorted(best)], "pipe": step.pipe is not None,
                          "kernel_us": ks}), flush=True)
        acm_gnn_amd.tuning.reset()
        AF.InputPipeline.refill_spec = real_refill
        del step, gstep, model, model2, opt, opt2, ops
        from acm_gnn_amd.graph import clear_cache
        clear_cache()


if __name__ == "__main__":
    main()
