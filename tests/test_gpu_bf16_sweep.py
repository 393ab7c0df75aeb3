"""BASELINE config 3: fp32 vs bf16 storage of the gathered operand (fp32 accumulation) on the real
Chameleon / Squirrel structures -- a sweep over the layer width (F = 16, 64), the ReLU placement (ACM / ACMII) and the
structure channel; the error of the layer output and of every gradient is written to
gpurun_out/bf16_sweep.json (-> profiles/r02_bf16_sweep.json) and bounded by what that table shows (thresholds first
seeded by the survey's CPU probe, SURVEY.md section 6: bf16 max abs 7.8e-3 / 3.9e-3, rms 3.7e-4 on rms(out) ~ 0.38).
bf16 operands are implemented for even 8 < F <= 64 (acm_conv_fwd_t.gather_bf16); wider layers stay fp32."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, load_npz
from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# max |g_bf16 - g_fp32| over the gradient's range (floor 1).  Recorded sweep (profiles/r02_bf16_sweep.json): <= 0.092 for
# every weight / attention / LayerNorm gradient; struc_low -- itself the rounded operand, behind a ReLU -- has single
# entries whose pre-activation changes sign under the rounding (kink flips): up to 0.27 of the range at F = 16, while its
# Frobenius error stays small (bounded separately).
GRAD_BOUND, STRUC_BOUND, FRO_BOUND = 0.12, 0.35, 0.25


TABLE = {}
SWEEP = [(name, mt, v, s, f) for name in ("chameleon", "squirrel") for f in (16, 64)
         for (mt, v, s) in (("acmgcnp", 0, 1), ("acmgcnp", 1, 0), ("acmgcnp", 1, 1), ("acmgcn", 0, 0))]


@pytest.mark.parametrize("name,model_type,variant,s,f_out", SWEEP)
def test_bf16_gather_tolerance(name, model_type, variant, s, f_out):
    from acm_gnn_amd import GraphConvolution, functional as AF
    from acm_gnn_amd.graph import clear_cache
    g = load_npz(os.path.join(GOLDEN, f"graph_{name}.npz"))
    n = int(g["n"])
    adj = sp.csr_matrix((np.ones(len(g["adj_un_indices"])), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    low, high, un = O.filters_linkx(adj)
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(n, 300, generator=gen) < 0.05).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
    gout = torch.randn(n, f_out, generator=gen)
    res = {}
    for dt in ("fp32", "bf16"):
        clear_cache()
        torch.manual_seed(3)
        layer = GraphConvolution(300, f_out, n, model_type, variant=variant, structure_info=s, attn_layernorm=True,
                                 gather_dtype=dt)
        params = {k: v.detach().cpu().clone() for k, v in layer.named_parameters()}
        layer = layer.to(DEV)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        xd = x.to(DEV).requires_grad_(True)
        out = layer(xd, low.to(DEV), high.to(DEV), un.to(DEV) if s else None)
        out.backward(gout.to(DEV))
        AF.set_kernel_timer(None)
        used = set(k.split("/")[0] for k in timer.events)
        assert ("cast_bf16" in used) == (dt == "bf16")
        res[dt] = (out.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in layer.named_parameters() if p.grad is not None})
    # fp32 path against the oracle (sanity), then bf16 against fp32
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.layer_forward(pr, x, low, high, un if s else None, model_type=model_type, variant=variant,
                          structure_info=s, attn_layernorm=True)
    assert float((res["fp32"][0] - ref.detach()).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    o32, o16 = res["fp32"][0], res["bf16"][0]
    rms = float(o32.pow(2).mean().sqrt())
    max_abs = float((o16 - o32).abs().max())
    rms_err = float((o16 - o32).pow(2).mean().sqrt())
    print(f"\n{name} {model_type} v{variant} s{s}: rms(out)={rms:.3f} max|out|={float(o32.abs().max()):.2f}  "
          f"bf16 max abs err {max_abs:.2e}  rms err {rms_err:.2e}")
    assert max_abs < 2e-2 * max(1.0, float(o32.abs().max())) and rms_err < 3e-3 * max(rms, 1e-3)
    assert max_abs > 0                                          # the option really changes the numerics
    row = {"rms_out": rms, "max_out": float(o32.abs().max()), "out_max_abs_err": max_abs, "out_rms_err": rms_err, "grads": {}}
    worst = 0.0
    for k, g32 in res["fp32"][2].items():
        g16 = res["bf16"][2][k]
        scale = max(1.0, float(g32.abs().max()))
        err = float((g16 - g32).abs().max())
        fro = float((g16 - g32).norm() / g32.norm().clamp_min(1e-30))
        row["grads"][k] = [err, float(g32.abs().max()), fro]
        if k != "struc_low":
            worst = max(worst, err / scale)
        # attention gradients amplify the operand's rounding (8 mantissa bits)
        assert err < (STRUC_BOUND if k == "struc_low" else GRAD_BOUND) * scale, (k, err, scale)
        assert fro < FRO_BOUND or float(g32.norm()) < 1e-6, (k, fro)
    row["worst_grad_err_over_range"] = worst
    TABLE[f"{name}/{model_type}/v{variant}s{s}/F{f_out}"] = row
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "bf16_sweep.json"), "w") as fh:
        json.dump(TABLE, fh, indent=1, sort_keys=True)


@pytest.mark.parametrize("model_type,variant,s", [("acmgcnp", 0, 0), ("acmgcnp", 1, 1)])
def test_bf16_forward_forms_agree(model_type, variant, s):
    """The two kernels a bf16 wide gather under the fused head can take (acm_conv.hip: the four-neighbour vector form on
    cache-resident tables, the pair kernel once the tables outgrow the Infinity Cache -- pokec / snap-patents sizes) read the
    same bf16 tables and differ in fp32 summation order only: same output and gradients on the Squirrel structure."""
    from conftest import tune_now as tune
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    g = load_npz(os.path.join(GOLDEN, "graph_squirrel.npz"))
    n = int(g["n"])
    adj = sp.csr_matrix((np.ones(len(g["adj_un_indices"])), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    low, high, un = O.filters_linkx(adj)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, 40, generator=gen)
    gout = torch.randn(n, 64, generator=gen)
    res = {}
    for form in (2, 3):                                  # 2: vector form everywhere, 3: the pair kernel
        tune(wide_form=form)
        clear_cache()
        torch.manual_seed(3)
        layer = GraphConvolution(40, 64, n, model_type, variant=variant, structure_info=s, attn_layernorm=True,
                                 gather_dtype="bf16").to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        out = layer(xd, low.to(DEV), high.to(DEV), un.to(DEV) if s else None)
        out.backward(gout.to(DEV))
        res[form] = (out.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in layer.named_parameters() if p.grad is not None})
    a, b = res[2], res[3]
    assert float((a[0] - b[0]).abs().max()) < 2e-5 * max(1.0, float(a[0].abs().max()))
    assert float((a[1] - b[1]).abs().max()) < 1e-4 * max(1.0, float(a[1].abs().max()))
    for k, v in a[2].items():
        assert float((b[2][k] - v).abs().max()) < 2e-4 * max(1.0, float(v.abs().max())), k
