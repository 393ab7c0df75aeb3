#!/usr/bin/env python3
"""Two roles in one kernel (scripts/micro/sell_gather.hip (f)): does the next step's input gather P = A dropout(X) fit
UNDER the VALU-bound layer-1 backward when both live in the same workgroups (extra gather waves beside the backward's
four)?  Two kernels on two streams do not overlap (scripts/probe_overlap.py -> profiles/r02_probe_overlap.txt).  The VALU
role is a dependent-FMA loop calibrated to the backward's 67 us at two workgroups per CU; the gather role is the streamed
pair kernel with R steps of rows in flight per wave.  Prints us for VALU only / gather only / both."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from acm_gnn_amd import data as D  # noqa: E402
from probe_sell import build_streams  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    so = "/tmp/sell_gather.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", "-o", so,
                           os.path.join(ROOT, "scripts", "micro", "sell_gather.hip")])
    lib = C.CDLL(so)
    lib.roles.argtypes = [C.c_int] * 6 + [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset("twitch-gamer")
    perm = D.degree_order(adj)
    adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
    low, deg = D.build_filters(adj)
    low = low.tocsr()
    low.sort_indices()
    pat = low.copy()
    pat.data[:] = 1.0
    indptr, indices = low.indptr.astype(np.int64), low.indices.astype(np.int32)
    sh = torch.cuda.current_stream().cuda_stream
    x = torch.randn(n, 8, device=DEV)
    ref = pat @ x.cpu().numpy().astype(np.float64)
    sink = torch.zeros(1024, device=DEV)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"n {n} nnz {low.nnz}; VALU role calibrated to {TARGET:.0f} us alone", flush=True)
    for gw, n_blocks, lds in ((2, 512, 80 * 1024), (4, 512, 80 * 1024), (2, 768, 53 * 1024), (8, 512, 80 * 1024)):
        n_waves = gw * n_blocks
        stream, wptr, wstep, desc, item_row, total = build_streams(indptr, indices, n_waves, max_steps=8)
        d = [torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc)]
        out = torch.zeros(item_row.size, 8, device=DEV)

        def run(mode, rows, iters):
            st = lib.roles(mode, rows, gw, iters, n_blocks, lds, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                           d[3].data_ptr(), x.data_ptr(), n * 32, out.data_ptr(), sink.data_ptr(), sh)
            assert st == 0, st

        t1k = timeit(lambda: run(1, 2, 1000))
        iters = int(round(1000 * TARGET / t1k))
        line = [f"gather waves/block {gw} blocks {n_blocks} ({(4 + gw) * n_blocks // 256 / 4:.1f} waves/SIMD, {n_waves} gather waves): "
                f"VALU only ({iters} iters) {timeit(lambda: run(1, 2, iters)):6.1f} us |"]
        for rows in (2, 4, 6, 8):
            if (rows == 8 and gw != 2) or (rows == 6 and gw == 8):
                continue
            out.zero_()
            run(2, rows, iters)
            torch.cuda.synchronize()
            got = np.zeros((n, 8))
            np.add.at(got, item_row, out.cpu().double().numpy())
            err = float(np.abs(got - ref).max())
            assert err < 1e-3, err
            line.append(f" R{rows}: gather only {timeit(lambda: run(2, rows, iters)):6.1f}  both {timeit(lambda: run(3, rows, iters)):6.1f} |")
        print("".join(line), flush=True)


TARGET = float(os.environ.get("VALU_US", "67"))
if __name__ == "__main__":
    main()
