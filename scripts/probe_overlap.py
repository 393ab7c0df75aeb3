#!/usr/bin/env python3
"""Can the VALU-bound row-local backward of layer 1 (agg_bwd, 65 us) and a memory-bound narrow gather (P = A X, 80 us) share
the chip?  (The gather of step t+1's layer-1 forward does not depend on step t's parameter update.)  Times both alone and
on two streams at once, eager launches, twitch-shaped graph."""
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
    x8 = torch.randn(n, 8, device=DEV)
    y8 = torch.empty(n, 8, device=DEV)
    gout = torch.randn(n, 64, device=DEV)
    params = [p for p in layer.parameters() if p.requires_grad]

    def fwd_bwd():
        for p in params:
            p.grad = None
        out = layer(x, ops, None, post_relu=True)
        out.backward(gout)

    def gath():
        AF.spmm(ops.low, x8, out=y8)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                         # warm-up outside any capture, on a side stream
        for _ in range(3):
            fwd_bwd()
            gath()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    s2 = torch.cuda.Stream()

    def capture(body):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        return g

    def both_forked():
        cur = torch.cuda.current_stream()
        out = layer(x, ops, None, post_relu=True)
        s2.wait_stream(cur)                               # fork after the forward: the gather runs beside the backward
        with torch.cuda.stream(s2):
            gath()
        out.backward(gout)
        cur.wait_stream(s2)

    def both_serial():
        fwd_bwd()
        gath()

    def fwd_only():
        with torch.no_grad():
            layer(x, ops, None, post_relu=True)

    def timeit(g, reps=30):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for blocks in ("768", "512", "256"):
        os.environ["ACM_AGG_BWD_BLOCKS"] = blocks
        for p in params:
            p.grad = None
        t_f = timeit(capture(fwd_only))
        t_fb = timeit(capture(fwd_bwd))
        t_g = timeit(capture(gath))
        t_ser = timeit(capture(both_serial))
        t_fork = timeit(capture(both_forked))
        print(f"agg_bwd blocks {blocks}: forward {t_f:6.1f}  forward+backward {t_fb:6.1f}  gather {t_g:6.1f}  "
              f"fwd+bwd then gather {t_ser:6.1f}  fwd, then backward beside the gather {t_fork:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
