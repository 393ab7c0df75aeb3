"""The HIP layer / model against the golden vectors recorded from the imported
reference (tests/golden) -- the parity gate.  Runs on the MI355X box."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files, graph_tensors, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32 parity thresholds (fused formulation vs reference <= 3e-7 relative)
FWD = dict(rtol=1e-5, atol=1e-5)
GRAD = dict(rtol=1e-4, atol=5e-5)


def _set_params(module, rec, prefix="param:"):
    sd = module.state_dict()
    for k, v in rec.items():
        if k.startswith(prefix):
            name = k[len(prefix):]
            assert name in sd, name
            assert tuple(sd[name].shape) == tuple(v.shape), (name, sd[name].shape, v.shape)
            sd[name].copy_(torch.from_numpy(v))


def _adj(dialect, structure_info):
    low, high, un, _ = graph_tensors(dialect)
    return low.to(DEV), high.to(DEV), (un.to(DEV) if structure_info else None)


def _close(actual, desired, what, rtol, atol):
    # atol scales with the tensor's magnitude: LayerNorm over F = 2 columns (the 2-class output
    # layer) amplifies fp32 round-off of either implementation by rstd ~ 1/sqrt(eps).
    atol = atol * max(1.0, float(np.abs(desired).max()))
    np.testing.assert_allclose(actual.detach().cpu().numpy(), desired, err_msg=what, rtol=rtol, atol=atol)


@pytest.mark.parametrize("path", golden_files("layer_*.npz"), ids=os.path.basename)
def test_layer_matches_reference_golden(path):
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    rec = load_npz(path)
    cfg = rec["cfg"]
    clear_cache()
    low, high, un = _adj(cfg["dialect"], cfg["structure_info"])
    n = rec["x"].shape[0]
    layer = GraphConvolution(cfg["f_in"], cfg["f_out"], n, cfg["model_type"], variant=cfg["variant"],
                             structure_info=cfg["structure_info"],
                             attn_layernorm=bool(cfg["attn_layernorm"])).to(DEV)
    _set_params(layer, rec)
    x = torch.from_numpy(rec["x"]).to(DEV).requires_grad_(True)
    out = layer(x, low, high, un)
    out.backward(torch.from_numpy(rec["grad_out"]).to(DEV))
    _close(out, rec["out"], "out", **FWD)
    atts = [layer.att_low, layer.att_high, layer.att_mlp]
    if rec["att"].shape[1] == 4:
        atts.append(layer.att_struc_vec_low)
    _close(torch.cat(atts, 1), rec["att"], "att", **FWD)
    _close(x.grad, rec["grad_x"], "grad_x", **GRAD)
    named = dict(layer.named_parameters())
    for k, v in rec.items():
        if k.startswith("grad:"):
            g = named[k[5:]].grad
            assert g is not None, k
            _close(g, v, k, **GRAD)
    for name, p in named.items():                       # unused parameters stay grad-less, as in the reference
        if "grad:" + name not in rec:
            assert p.grad is None, name


class _MaskReplay:
    def __init__(self, masks):
        self.masks = list(masks)

    def __call__(self, inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        return inp * self.masks.pop(0) / (1.0 - p)


@pytest.mark.parametrize("path", golden_files("model_*.npz"), ids=os.path.basename)
def test_model_matches_reference_golden(path, monkeypatch):
    import torch.nn.functional as F
    from acm_gnn_amd import GCN
    from acm_gnn_amd.graph import clear_cache
    rec = load_npz(path)
    cfg = rec["cfg"]
    clear_cache()
    low, high, un = _adj(cfg["dialect"], cfg["structure_info"])
    n = rec["x"].shape[0]
    model = GCN(cfg["f_in"], cfg["hidden"], cfg["classes"], 2, n, cfg["dropout"], cfg["model_type"],
                cfg["structure_info"], variant=cfg["variant"],
                attn_layernorm=bool(cfg["attn_layernorm"])).to(DEV)
    _set_params(model, rec)
    order = ["x"] + (["xX"] if cfg["model_type"] == "acmgcnpp" else []) + ["hidden"]
    masks = [torch.from_numpy(rec["mask:" + nm].astype(np.float32)).to(DEV) for nm in order if "mask:" + nm in rec]
    monkeypatch.setattr(F, "dropout", _MaskReplay(masks))
    model.train()
    logits = model(torch.from_numpy(rec["x"]).to(DEV), low, high, un)
    idx = torch.from_numpy(rec["train_idx"]).to(DEV)
    labels = torch.from_numpy(rec["labels"]).to(DEV)
    loss = F.nll_loss(F.log_softmax(logits, dim=1)[idx], labels[idx])
    loss.backward()
    _close(logits, rec["logits"], "logits", **FWD)
    assert abs(loss.item() - float(rec["loss"])) < 1e-5 * max(1.0, abs(float(rec["loss"])))
    named = dict(model.named_parameters())
    for k, v in rec.items():
        if k.startswith("grad:"):
            g = named[k[5:]].grad
            assert g is not None, k
            _close(g, v, k, **GRAD)


def test_state_dict_keys_match_reference_layer():
    """Parameter names recorded from the reference module == ours (drop-in state_dict)."""
    from acm_gnn_amd import GraphConvolution
    rec = load_npz(os.path.join(GOLDEN, "layer_geometric_acmgcnp_v0_s1.npz"))
    ref_names = sorted(k[6:] for k in rec if k.startswith("param:"))
    layer = GraphConvolution(12, 16, 96, "acmgcnp", structure_info=1)
    assert sorted(dict(layer.named_parameters())) == ref_names
    assert repr(layer) == "GraphConvolution (12 -> 16)"
