#!/usr/bin/env python3
"""Column-blocked wide gather: Y = sum_b A[:, block_b] X[block_b, :] with the column blocks chosen so that one block's rows
of X fit the L2 (the 64-wide gathers of the twitch graph are fabric-bound: half their lines come from beyond the XCD's L2).
Times one 64-wide acm_spmm over the whole operator against B passes over column-block sub-operators (same kernels, each pass
writes its own partial output; the sum over passes is not included -- it streams B x n x 256 B)."""
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


def main():
    adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset("twitch-gamer")
    perm = D.degree_order(adj)
    adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
    low, deg = D.build_filters(adj)
    low = low.tocsr()
    low.sort_indices()
    pat = low.copy()
    pat.data[:] = 1.0

    def handle(m):
        m = m.tocsr()
        m.sort_indices()
        return CsrGraph.from_csr(torch.from_numpy(m.indptr.astype(np.int32)).to(DEV),
                                 torch.from_numpy(m.indices.astype(np.int32)).to(DEV), None, n)

    def timeit(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    x = torch.randn(n, 64, device=DEV)
    y = torch.empty(n, 64, device=DEV)
    g = handle(pat)
    print(f"one pass, 64 columns: {timeit(lambda: AF.spmm(g, x, out=y)):7.1f} us", flush=True)
    ref = y.clone()
    cnt = np.bincount(pat.indices, minlength=n)
    cum = np.cumsum(cnt)
    for nb in (2, 4, 8, 16):
        cuts = [0] + [int(np.searchsorted(cum, cum[-1] * b / nb)) for b in range(1, nb)] + [n]
        subs, ys = [], []
        csc = pat.tocsc()
        for b in range(nb):
            keep = sp.csc_matrix((n, n), dtype=pat.dtype).tolil()
            m = pat[:, cuts[b]:cuts[b + 1]]
            full = sp.hstack([sp.csr_matrix((n, cuts[b])), m, sp.csr_matrix((n, n - cuts[b + 1]))]).tocsr()
            subs.append(handle(full))
            ys.append(torch.empty(n, 64, device=DEV))

        def run():
            for h, yy in zip(subs, ys):
                AF.spmm(h, x, out=yy)
        t = timeit(run)
        err = float((sum(ys) - ref).abs().max())
        rows = [cuts[b + 1] - cuts[b] for b in range(nb)]
        print(f"{nb:2d} column blocks (rows per block {min(rows)}..{max(rows)}, table slice up to {max(rows) * 256 / 1e6:.1f} MB): "
              f"{t:7.1f} us for the passes (+ {nb * n * 256 / 1e6:.0f} MB of partial outputs to add), max err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
