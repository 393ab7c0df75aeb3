#!/usr/bin/env python3
"""Run ON THE GPU BOX: the split-bf16 projections (acm_gemm_bx3.hip) against the fp32 MFMA kernels -- time and error
against a float64 product."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import functional as AF, tuning

dev = torch.device("cuda", 0)
def timeit(fn, reps=10, inner=10):
    """us per call: `inner` calls captured in one graph (no host time between the launches), `reps` replays"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
    g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * inner) * 1e6
def err(got, ref64, scale):
    return float(((got.double() - ref64).abs() / scale).max())
for (n, k, nn) in ((169343, 128, 192), (169343, 128, 21), (169343, 128, 15), (41554, 128, 192), (100000, 100, 70)):
    torch.manual_seed(0)
    x = torch.randn(n, k, device=dev); w = torch.randn(k, nn, device=dev) * 0.1; dz = torch.randn(n, nn, device=dev)
    x[::7] *= 1e3; x[::5] *= 1e-4
    st = AF.DropoutState(dev, seed=1)
    spec = st.spec(0.1, 0, 0)
    res = {"shape": [n, k, nn]}
    ref_z = x.double() @ w.double()
    sc_z = x.double().abs() @ w.double().abs() + 1e-30
    ref_dw = x.double().t() @ dz.double()
    sc_dw = x.double().abs().t() @ dz.double().abs() + 1e-30
    xd = AF.dropout(x, 0.1, st)
    ref_zd = xd.double() @ w.double()
    ref_dwd = xd.double().t() @ dz.double()
    for off in ("1", ""):
        tuning.apply(gemm_forms=1 if off else 7)
        tag = "f32" if off else "bx3"
        z = torch.empty(n, nn, device=dev); dw = torch.empty(k, nn, device=dev)
        res[f"nn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, w, out=z)), 1)
        res[f"nn_{tag}_err"] = err(z, ref_z, sc_z)
        res[f"tn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw)), 1)
        res[f"tn_{tag}_err"] = err(dw, ref_dw, sc_dw)
        if True:
            try:
                res[f"nn_{tag}_drop_us"] = round(timeit(lambda: AF.gemm(x, w, out=z, a_drop=spec)), 1)
                res[f"nn_{tag}_drop_err"] = err(z, ref_zd, sc_z)
                res[f"tn_{tag}_drop_us"] = round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw, a_drop=spec)), 1)
                res[f"tn_{tag}_drop_err"] = err(dw, ref_dwd, sc_dw)
            except Exception as e:  # noqa: BLE001
                res[f"{tag}_drop"] = str(e)[:80]
    tuning.reset()
    print(json.dumps(res), flush=True)
