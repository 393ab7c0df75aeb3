#!/usr/bin/env python3
"""Is the output layer's narrow gather bound by the chain of its longest row?  

The twitch-shaped operator's top row has ~21 k neighbours: sixteen pieces of ~1.3 k, all in one window = one workgroup, each
16-lane group walking its piece in ~41 dependent steps.  This probe times the same gather kernel
(acm_spmm on a [n, 4] table: spmm_narrow_kernel<.., EpiPlain>) on the benchmark's operator as it is, and on the same operator
with every row CAPPED at `cap` neighbours (the hubs' extra edges dropped: < 1 % of the nonzeros) -- if the longest row's chain
bounds the launch, capping the hubs must make it faster by about the chain's length.

    python scripts/probe_hub_chain.py > gpurun_out/r05_probe_hub_chain.txt
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import data as D, functional as AF  # noqa: E402
from acm_gnn_amd.graph import CsrGraph  # noqa: E402

DEV = torch.device("cuda:0")


def timed(csr, table, reps=20, replays=20):
    """us per launch: `reps` launches captured in one hipGraph (an eager loop is bound by the host's ~40 us per call)."""
    out = torch.empty(csr.n_rows, table.shape[1], device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            AF.spmm(csr, table, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            AF.spmm(csr, table, out=out)
    g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (reps * replays) * 1e6


def main():
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    low = wl["low"].tocsr()
    low.sort_indices()
    n = low.shape[0]
    deg = np.diff(low.indptr)
    table = torch.randn(n, 4, device=DEV)
    print(f"# twitch-shaped operator: {n} rows, nnz {low.nnz}, five longest rows {np.sort(deg)[-5:][::-1].tolist()}")
    print("# cap (neighbours kept per row) | nnz | nonzeros dropped % | rows above the cap | us per launch (acm_spmm, width 4, pattern-only)")
    for cap in (0, 16384, 8192, 4096, 2048, 1024):
        if cap:
            keep = np.ones(low.nnz, bool)
            for r in np.nonzero(deg > cap)[0]:
                keep[low.indptr[r] + cap: low.indptr[r + 1]] = False
            rows = np.repeat(np.arange(n), deg)[keep]
            m = sp.csr_matrix((np.ones(keep.sum(), np.float32), (rows, low.indices[keep])), shape=(n, n))
        else:
            m = sp.csr_matrix((np.ones(low.nnz, np.float32), low.indices, low.indptr), shape=(n, n))
        m.sort_indices()
        csr = CsrGraph.from_csr(torch.from_numpy(m.indptr.astype(np.int32)).to(DEV), torch.from_numpy(m.indices.astype(np.int32)).to(DEV),
                                None, n)
        us = timed(csr, table)
        print(f"{cap or 'none':>6} | {m.nnz} | {100 * (1 - m.nnz / low.nnz):.2f} | {int((deg > cap).sum()) if cap else 0} | {us:.1f}", flush=True)


if __name__ == "__main__":
    main()
