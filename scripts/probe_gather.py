#!/usr/bin/env python3
"""Where does the narrow gather's time go?  Same row structure as the twitch-shaped graph, column ids
rewritten to control which cache level serves the gathered rows:
    real      the graph's own columns (degree-ordered labels)
    seq       column = edge position mod N            (coalesced: floor without the random gather)
    l1        column = random in [0, 256)             (every row in the CU's L1)
    l2small   column = random in [0, 16384)           (512 KB at 32 B rows: L2 hits)
    random    column = uniform random in [0, N)       (no locality at all)
    */unit    the same columns in a pattern-only handle (no value stream)
Prints us per acm_spmm call, warm (20 back-to-back calls: the 110 MB of column ids + values stay in the 256 MB
MALL) / cold (640 MB written before every call), for widths 2, 4, 8 (row = 8, 16, 32 bytes) and 64.
"""
import os
import sys

import numpy as np
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
    nnz = low.nnz
    rng = np.random.default_rng(0)
    variants = {
        "real": low.indices,
        "seq": (np.arange(nnz, dtype=np.int64) % n).astype(np.int32),
        "l1": rng.integers(0, 256, nnz).astype(np.int32),
        "l2small": rng.integers(0, 16384, nnz).astype(np.int32),
        "random": rng.integers(0, n, nnz).astype(np.int32),
    }
    indptr = torch.from_numpy(low.indptr.astype(np.int32)).to(DEV)
    vals = torch.from_numpy(low.data.astype(np.float32)).to(DEV)
    flush = torch.empty(160 * 1024 * 1024, device=DEV)                   # 640 MB: evicts L2 and the 256 MB MALL
    for name, idx in list(variants.items()) + [("real/unit", variants["real"]), ("random/unit", variants["random"])]:
        g = CsrGraph.from_csr(indptr, torch.from_numpy(np.ascontiguousarray(idx)).to(DEV),
                              None if name.endswith("/unit") else vals, n)
        line = [f"{name:11s}"]
        for width in (2, 4, 8, 64):
            x = torch.randn(n, width, device=DEV)
            y = torch.empty(n, width, device=DEV)
            for _ in range(3):
                AF.spmm(g, x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                AF.spmm(g, x, out=y)
            e1.record()
            torch.cuda.synchronize()
            warm = e0.elapsed_time(e1) / 20 * 1e3
            cold = 0.0
            for _ in range(5):                                             # cold: caches flushed before every call
                flush.fill_(1.0)
                e0.record()
                AF.spmm(g, x, out=y)
                e1.record()
                torch.cuda.synchronize()
                cold += e0.elapsed_time(e1) / 5 * 1e3
            line.append(f"w{width}: {warm:6.1f} / {cold:6.1f} us")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
