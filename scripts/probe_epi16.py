#!/usr/bin/env python3
"""Run ON THE GPU BOX: the sixteen-rows-per-wave row-local stages (acm_conv_agg16.hip) against the older kernels
(acm_tuning_t.rows16 bits 1 / 2 cleared) on the benchmark's first layer: results, head statistics, gradients, and HIP-event times
of both.  Prints one line per comparison."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from acm_gnn_amd import data as D, distributed as DD, functional as AF  # noqa: E402
from acm_gnn_amd.layers import GraphConvolution  # noqa: E402


def run(layer, x, ops, go, drop, unfused, off):
    from acm_gnn_amd import tuning
    tuning.apply(agg_fused=0 if unfused else 1, rows16=4 if off else 7)
    layer.zero_grad(set_to_none=True)
    out = layer(x, ops, post_relu=True, post_drop=drop)
    out.backward(go)
    grads = {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}
    return out.detach().clone(), torch.cat([layer.att_low, layer.att_high, layer.att_mlp], 1).clone(), grads


def main():
    dev = torch.device("cuda", 0)
    size = sys.argv[1] if len(sys.argv) > 1 else "twitch-gamer"
    wl = D.bench_workload(size, seed=0, node_order="degree", uniform=False, pad_to=1)
    low, deg, x_np = wl["low"], wl["deg"], wl["x"]
    n = low.shape[0]
    ops = DD.make_sharded_operators(low, deg, dev)
    x8 = torch.zeros(n, 8, device=dev)
    x8[:, : x_np.shape[1]] = torch.from_numpy(x_np).to(dev)
    res = {}
    for mt in ("acmgcnp", "acmgcn"):
        torch.manual_seed(1)
        layer = GraphConvolution(x_np.shape[1], 64, n, mt).to(dev)
        layer.train()
        go = torch.randn(n, 64, device=dev) * 1e-3
        state = AF.DropoutState(dev, seed=5)
        drop = (0.1, 1, state)
        ref = run(layer, x8, ops, go, drop, unfused=True, off=True)
        new = run(layer, x8, ops, go, drop, unfused=True, off=False)
        fused = run(layer, x8, ops, go, drop, unfused=False, off=True)
        for name, a, b in (("new_vs_old_unfused", new, ref), ("old_unfused_vs_fused", ref, fused)):
            scale = float(b[0].abs().max())
            line = {"model": mt, "cmp": name, "out_max_abs": float((a[0] - b[0]).abs().max()), "out_range": scale,
                    "att_max_abs": float((a[1] - b[1]).abs().max()),
                    "mask_mismatch": int(((a[0] == 0) != (b[0] == 0)).sum())}
            for k in b[2]:
                gr = float(b[2][k].abs().max())
                line["d_" + k] = float((a[2][k] - b[2][k]).abs().max()) / max(gr, 1e-30)
            print(json.dumps(line))
        # timing: the row-local stages alone (unfused path: gather + epilogue; backward kernel)
        for off in (True, False):
            timer = AF.KernelTimer()
            AF.set_kernel_timer(timer)
            for _ in range(20):
                run(layer, x8, ops, go, drop, unfused=True, off=off)
            AF.set_kernel_timer(None)
            s = timer.summary()
            res[(mt, off)] = {k: round(v[1] / v[0] * 1e3, 1) for k, v in s.items()}
            print(json.dumps({"model": mt, "old_kernels": off, "us_per_launch": res[(mt, off)]}))


if __name__ == "__main__":
    main()
