#!/usr/bin/env python3
"""Run ON THE GPU BOX: the benchmark's plain training step with the output layer's projection backward inside the hidden
layer's backward kernel (acm_conv_agg_bwd_t.proj_*) against ACM_LAZY_DX=0: loss, every gradient, step time."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd
from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T

dev = torch.device("cuda", 0)
wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
low, deg, x_np, y_np, splits = wl["low"], wl["deg"], wl["x"], wl["y"], wl["splits"]
n = low.shape[0]
x, y = torch.from_numpy(x_np).to(dev), torch.from_numpy(y_np).to(dev)
w = T.row_weights(torch.from_numpy(splits[0]).to(dev), n, device=dev)
res = {}
for lazy in ("0", "1"):
    os.environ["ACM_LAZY_DX"] = lazy
    ops = DD.make_sharded_operators(low, deg, dev)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, "acmgcnp", 0, variant=False).to(dev)
    model.dropout_state = AF.DropoutState(dev, seed=3)
    opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    step = T.TrainStep(model, opt, x, ops, y, w, use_graph=False, pipeline_input=False)
    opt.zero_grad(set_to_none=True)
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    loss = step._forward_backward()
    torch.cuda.synchronize()
    AF.set_kernel_timer(None)
    res[lazy] = (float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}, sorted(timer.events))
    opt2 = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
    g = T.TrainStep(model, opt2, x, ops, y, w, use_graph=True, pipeline_input=False)
    for _ in range(10):
        g()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            g()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print(json.dumps({"lazy": lazy, "kernels": res[lazy][2], "loss": res[lazy][0], "graph_ms": [round(t, 4) for t in sorted(ts)]}), flush=True)
worst = 0.0
for k in res["0"][1]:
    a, b = res["1"][1][k], res["0"][1][k]
    rel = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    worst = max(worst, rel)
    if rel > 1e-5:
        print("grad", k, rel)
print(json.dumps({"loss_diff": abs(res["0"][0] - res["1"][0]), "worst_grad_rel": worst}))
