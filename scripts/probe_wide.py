#!/usr/bin/env python3
"""Where does the WIDE gather's time go?  acm_spmm on the twitch-shaped structure (pattern-only handle, degree-ordered
ids) with 64- and 128-column tables, column ids rewritten to control which cache level serves the gathered rows:
    real      the graph's own columns
    l2hot     column = random in [0, 8192)      (2 MB at 256 B rows: L2 hits)
    top64k    column = random in [0, 65536)     (16 MB: beyond one XCD's L2, inside the Infinity Cache)
    random    column = uniform random in [0, N) (43 / 86 MB table through the Infinity Cache)
    seq       column = edge position mod N      (streaming floor)
for the dword-per-lane kernel (acm_tuning_t.wide_form = 1) and the dwordx4 kernel.  Prints us per call (20 back-to-back).
    python scripts/probe_wide.py [variant ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import data as D, functional as AF, tuning  # noqa: E402
from acm_gnn_amd.graph import CsrGraph  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    wl = D.bench_workload("twitch-gamer", node_order="degree")
    low = wl["low"].tocsr()
    n, nnz = low.shape[0], low.nnz
    rng = np.random.default_rng(0)
    variants = {
        "real": low.indices,
        "l2hot": rng.integers(0, 8192, nnz).astype(np.int32),
        "top64k": rng.integers(0, 65536, nnz).astype(np.int32),
        "random": rng.integers(0, n, nnz).astype(np.int32),
        "seq": (np.arange(nnz, dtype=np.int64) % n).astype(np.int32),
    }
    want = sys.argv[1:] or list(variants)
    ip = torch.from_numpy(low.indptr.astype(np.int32)).to(DEV)
    for name in want:
        ix = torch.from_numpy(np.ascontiguousarray(variants[name])).to(DEV)
        g = CsrGraph.from_csr(ip, ix, None, n)
        for width in (64, 128):
            dense = torch.randn(n, width, device=DEV)
            out = torch.empty(n, width, device=DEV)
            res = {}
            for mode in ("scalar", "vec"):
                tuning.apply(wide_form=1 if mode == "scalar" else 0)
                for _ in range(3):
                    AF.spmm(g, dense, out=out)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(20):
                    AF.spmm(g, dense, out=out)
                torch.cuda.synchronize()
                res[mode] = (time.perf_counter() - t) / 20 * 1e6
            gb = nnz * width * 4 / 1e9
            print(f"{name:8s} W={width:3d}  scalar {res['scalar']:7.1f} us ({gb / res['scalar'] * 1e3:5.1f} TB/s gathered)   "
                  f"vec {res['vec']:7.1f} us ({gb / res['vec'] * 1e3:5.1f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
