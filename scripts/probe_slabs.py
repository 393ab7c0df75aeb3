#!/usr/bin/env python3
"""Would column slabs help the wide gathers?  The twitch-shaped operator (degree-ordered labels, pattern-only) cut into
column slabs A = sum_s A[:, slab s]; every slab product is an acm_spmm launch of its own over the SAME 64-wide table, so the
rows one launch gathers span slab_rows x 256 bytes.  Prints the whole operator's time and, per slab size, the sum of the
slab launches' times (HIP events around each; 20 rounds) -- an upper bound of what a slab-phased kernel could reach: the
sum pays every launch's ramp and walks the row pointers once per slab."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import data as D, functional as AF  # noqa: E402
from acm_gnn_amd.graph import CsrGraph  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, rounds=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "twitch-gamer"
    widths = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64]
    slabs = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4096, 8192, 16384, 32768, 65536]
    if name == "pokec":                                    # the pokec-shaped graph of scripts/bench_scale.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_scale import _powerlaw_graph_on_gpu
        n = 1_632_803
        adj = _powerlaw_graph_on_gpu(n, 30_622_564, 14_854, seed=3, directed=False)
        perm = D.degree_order(adj)
        adj = adj[perm][:, perm].tocsr()
    else:
        adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset(name)
        perm = D.degree_order(adj)
        adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
    low, deg = D.build_filters(adj)
    low = low.tocsr()
    low.sort_indices()
    print(f"{name}: n {n}, nnz {low.nnz}", flush=True)

    def handle(m):
        m = m.tocsr()
        return CsrGraph.from_csr(torch.from_numpy(m.indptr.astype(np.int32)).to(DEV),
                                 torch.from_numpy(m.indices.astype(np.int32)).to(DEV), None, n)
    for width in widths:
        table = torch.randn(n, width, device=DEV)
        y = torch.empty(n, width, device=DEV)
        whole = handle(low)
        print(f"W{width} whole operator: {timed(lambda: AF.spmm(whole, table, out=y)):7.1f} us", flush=True)
        csc = low.tocsc()
        for slab in slabs:
            parts = []
            for b in range(0, n, slab):
                e = min(n, b + slab)
                m = sp.csc_matrix((csc.data[csc.indptr[b]:csc.indptr[e]], csc.indices[csc.indptr[b]:csc.indptr[e]],
                                   np.concatenate([np.zeros(b, np.int64), csc.indptr[b:e + 1] - csc.indptr[b],
                                                   np.full(n - e, csc.indptr[e] - csc.indptr[b], np.int64)])), shape=(n, n))
                parts.append((handle(m), m.nnz))

            def run():
                for g, _ in parts:
                    AF.spmm(g, table, out=y)
            total = timed(run, rounds=5)
            first = timed(lambda: AF.spmm(parts[0][0], table, out=y))
            last = timed(lambda: AF.spmm(parts[-1][0], table, out=y))
            print(f"W{width} slabs of {slab:6d} rows ({len(parts):3d} launches): {total:7.1f} us in all; first slab "
                  f"({parts[0][1]} nnz) {first:6.1f} us, last ({parts[-1][1]} nnz) {last:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
