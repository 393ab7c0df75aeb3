#!/usr/bin/env python3
"""Run ON THE GPU BOX: the split-bf16 kernels on wide inputs (K > 128: Squirrel / Chameleon / Cora / Penn94 shapes) against the
fp32 tile kernel -- time (graph replay) and error against float64."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from acm_gnn_amd import functional as AF, tuning
from probe_bx3_parts import timeit
dev = torch.device("cuda", 0)
def err(got, a, b):
    return float(((got.double() - a @ b).abs() / (a.abs() @ b.abs() + 1e-30)).max())
for (n, k, nn) in ((5201, 2089, 192), (2708, 1433, 192), (41554, 4814, 192), (41554, 4816, 192), (20000, 1024, 70), (169343, 256, 192)):
    torch.manual_seed(0)
    x = torch.randn(n, k, device=dev); w = torch.randn(k, nn, device=dev) * 0.1; dz = torch.randn(n, nn, device=dev)
    res = {"shape": [n, k, nn]}
    x64, w64, dz64 = x.double(), w.double(), dz.double()
    for off in ("1", ""):
        tuning.apply(gemm_forms=3 if off else 7)
        tag = "f32" if off else "bx3"
        z = torch.empty(n, nn, device=dev); dw = torch.empty(k, nn, device=dev)
        res[f"nn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, w, out=z)), 1)
        res[f"nn_{tag}_err"] = err(z, x64, w64)
        res[f"tn_{tag}_us"] = round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw)), 1)
        res[f"tn_{tag}_err"] = err(dw, x64.t(), dz64)
    tuning.reset()
    print(json.dumps(res), flush=True)
