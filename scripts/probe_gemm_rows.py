#!/usr/bin/env python3
"""Run ON THE GPU BOX: the row-panel GEMMs (acm_gemm_rows.hip) against the tile kernel on the arXiv-year projection shapes."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import functional as AF, tuning

dev = torch.device("cuda", 0)
def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for (n, k, nn) in ((169343, 128, 192), (169343, 128, 21), (41554, 128, 192)):
    x = torch.randn(n, k, device=dev); w = torch.randn(k, nn, device=dev); dz = torch.randn(n, nn, device=dev)
    st = AF.DropoutState(dev, seed=1)
    spec = st.spec(0.1, 0, 0)
    z = torch.empty(n, nn, device=dev)
    dw = torch.empty(k, nn, device=dev)
    res = {"shape": [n, k, nn]}
    for off in ("", "1"):
        tuning.apply(gemm_forms=0 if off else 9)
        tag = "tile" if off else "rows"
        res[f"nn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, w, out=z)), 1)
        res[f"tn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw)), 1)
    tuning.apply(gemm_forms=9)
    res["nn_rows_drop_us"] = round(timeit(lambda: AF.gemm(x, w, out=z, a_drop=spec)), 1)
    res["tn_rows_drop_us"] = round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw, a_drop=spec)), 1)
    res["dropout_pass_us"] = round(timeit(lambda: AF.dropout(x, 0.1, st)), 1)
    print(json.dumps(res), flush=True)
