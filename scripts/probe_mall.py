#!/usr/bin/env python3
"""Is the fused aggregate-first forward slower inside the training step because its id stream comes from HBM instead of
the Infinity Cache?  Times the layer-1 forward kernel of the benchmark model (twitch-shaped graph) warm (back to back),
cold (640 MB written before every call) and cold + the id stream pulled back into the Infinity Cache by an unrelated
narrow product over the same operator (acm_spmm, width 2) right before the call."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import GraphConvolution, data as D, functional as AF  # noqa: E402
from acm_gnn_amd.distributed import make_sharded_operators  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    low, deg = wl["low"], wl["deg"]
    n = low.shape[0]
    ops = make_sharded_operators(low, deg, DEV)
    torch.manual_seed(0)
    layer = GraphConvolution(7, 64, n, "acmgcnp", variant=0, structure_info=0, attn_layernorm=True).to(DEV)
    x = torch.randn(n, 7, device=DEV)
    x2 = torch.randn(n, 2, device=DEV)
    y2 = torch.empty(n, 2, device=DEV)
    flush = torch.empty(160 * 1024 * 1024, device=DEV)

    def run(mode, reps=15):
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        with torch.no_grad():
            for _ in range(reps):
                if mode != "warm":
                    flush.fill_(1.0)
                if mode == "cold+ids":
                    AF.spmm(ops.low, x2, out=y2)
                layer(x, ops, None)
        AF.set_kernel_timer(None)
        torch.cuda.synchronize()
        ev = {k: v for k, v in timer.summary().items() if k.startswith("conv_agg_fwd")}
        return ev

    for mode in ("warm", "cold", "cold+ids", "warm"):
        print(mode, run(mode), flush=True)


if __name__ == "__main__":
    main()
