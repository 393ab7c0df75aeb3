#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (it needs /root/reference); the GPU box and
the test-suite only ever see the .npz files this writes.  Nothing from the
reference is copied: the script drives the reference's own classes/functions
and stores inputs + outputs as data.

    python tests/golden/make_golden.py            # regenerate everything

The two reference packages define clashing top-level modules (utils, layers,
models), so each dialect is generated in its own subprocess:

    --part pytorch     ACM-Pytorch   (attention LayerNorm never fires, quirk Q1)
    --part geometric   ACM-Geometric (LayerNorm fires for acmgcnp/acmgcnpp)
    --part graphs      real graph structures / features as data fixtures
"""
import argparse
import json
import os
import subprocess
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")


# ---------------------------------------------------------------- helpers
def small_graph(n, seed, n_self_loops=3):
    """Random undirected graph with: a few raw self-loops (quirk Q5), one
    isolated node, one hub.  Returns scipy CSR of the raw 0/1 adjacency."""
    rng = np.random.default_rng(seed)
    a = (rng.random((n, n)) < 0.06).astype(np.float64)
    a = np.triu(a, 1)
    a[0, 1:] = rng.random(n - 1) < 0.6          # hub
    a = a + a.T
    a[n - 1, :] = 0
    a[:, n - 1] = 0                              # isolated node
    for i in rng.choice(n - 1, n_self_loops, replace=False):
        a[i, i] = 1.0
    return sp.csr_matrix(a)


def csr_pack(prefix, t):
    """torch sparse/dense -> dict of CSR arrays (coalesced, sorted)."""
    if t is None:
        return {}
    if t.layout == torch.strided:
        return {prefix + "_dense": t.detach().numpy().copy()}
    t = t.coalesce()
    i = t.indices().numpy()
    m = sp.csr_matrix((t.values().numpy(), (i[0], i[1])), shape=tuple(t.shape))
    m.sort_indices()
    return {prefix + "_indptr": m.indptr.astype(np.int32),
            prefix + "_indices": m.indices.astype(np.int32),
            prefix + "_vals": m.data.astype(np.float32)}


def randomize_layernorms(layer, gen):
    with torch.no_grad():
        for name in ("low", "high", "mlp", "struc_low", "struc_high"):
            ln = getattr(layer, "layer_norm_" + name)
            ln.weight.copy_(torch.rand(ln.weight.shape, generator=gen) + 0.5)
            ln.bias.copy_(torch.rand(ln.bias.shape, generator=gen) - 0.5)


def dump_layer_case(path, GC, cfg, graph, seed):
    """Run one reference layer forward+backward and store everything."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    adj_low, adj_high, adj_un = graph
    n = adj_high.shape[0]
    layer = GC(cfg["f_in"], cfg["f_out"], n, cfg["model_type"],
               variant=cfg["variant"], structure_info=cfg["structure_info"])
    randomize_layernorms(layer, gen)
    x = torch.randn(n, cfg["f_in"], generator=gen, requires_grad=True)
    if cfg.get("sparse_x"):
        with torch.no_grad():
            x.mul_((torch.rand(x.shape, generator=gen) < 0.3).float())
    out = layer(x, adj_low, adj_high, adj_un if cfg["structure_info"] else None)
    gout = torch.randn(out.shape, generator=gen)
    out.backward(gout)
    rec = {"cfg": json.dumps(cfg), "x": x.detach().numpy(), "grad_out": gout.numpy(),
           "out": out.detach().numpy(), "grad_x": x.grad.numpy()}
    for name, p in layer.named_parameters():
        rec["param:" + name] = p.detach().numpy()
        if p.grad is not None:
            rec["grad:" + name] = p.grad.numpy()
    atts = [layer.att_low, layer.att_high, layer.att_mlp]
    if hasattr(layer, "att_struc_vec_low"):
        atts.append(layer.att_struc_vec_low)
    rec["att"] = torch.cat([a.detach() for a in atts], 1).numpy()
    np.savez_compressed(path, **rec)


class MaskRecorder:
    """Replace torch.nn.functional.dropout by explicit Bernoulli keep-masks so
    the masks the reference trained with can be replayed elsewhere."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.masks = []

    def __call__(self, inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        keep = torch.bernoulli(torch.full(inp.shape, 1.0 - p), generator=self.gen)
        self.masks.append(keep)
        return inp * keep / (1.0 - p)


def dump_model_case(path, GCN, cfg, graph, seed):
    import torch.nn.functional as F
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    adj_low, adj_high, adj_un = graph
    n = adj_high.shape[0]
    model = GCN(nfeat=cfg["f_in"], nhid=cfg["hidden"], nclass=cfg["classes"], nlayers=2,
                nnodes=n, dropout=cfg["dropout"], model_type=cfg["model_type"],
                structure_info=cfg["structure_info"], variant=cfg["variant"])
    for layer in model.gcns:
        randomize_layernorms(layer, gen)
    x = torch.randn(n, cfg["f_in"], generator=gen)
    labels = torch.randint(0, cfg["classes"], (n,), generator=gen)
    train_idx = torch.randperm(n, generator=gen)[: n // 2]
    rec_drop = MaskRecorder(seed + 2)
    real_dropout = F.dropout
    F.dropout = rec_drop
    try:
        model.train()
        logits = model(x, adj_low, adj_high, adj_un if cfg["structure_info"] else None)
    finally:
        F.dropout = real_dropout
    loss = F.nll_loss(F.log_softmax(logits, dim=1)[train_idx], labels[train_idx])
    loss.backward()
    rec = {"cfg": json.dumps(cfg), "x": x.numpy(), "labels": labels.numpy(),
           "train_idx": train_idx.numpy(), "logits": logits.detach().numpy(),
           "loss": np.float32(loss.item())}
    names = ["x"] + (["xX"] if cfg["model_type"] == "acmgcnpp" else []) + ["hidden"]
    for nm, m in zip(names, rec_drop.masks):
        rec["mask:" + nm] = m.numpy().astype(np.uint8)
    for name, p in model.named_parameters():
        if name in ("fea_param", "xX_param"):      # uninitialised, unused (models.py:41)
            continue
        rec["param:" + name] = p.detach().numpy()
        if p.grad is not None:
            rec["grad:" + name] = p.grad.numpy()
    np.savez_compressed(path, **rec)


LAYER_CASES = [
    # (tag, model_type, variant, structure_info, f_in, f_out, extra)
    ("acmgcn_v0", "acmgcn", 0, 0, 12, 16, {}),
    ("acmgcn_v1", "acmgcn", 1, 0, 12, 16, {}),
    ("acmgcn_v0_wide", "acmgcn", 0, 0, 24, 64, {"sparse_x": 1}),
    ("acmgcn_v0_out5", "acmgcn", 0, 0, 16, 5, {}),
    ("acmgcnp_v0_s0", "acmgcnp", 0, 0, 12, 16, {}),
    ("acmgcnp_v1_s0", "acmgcnp", 1, 0, 12, 16, {}),
    ("acmgcnp_v0_s1", "acmgcnp", 0, 1, 12, 16, {}),
    ("acmgcnp_v1_s1", "acmgcnp", 1, 1, 12, 16, {}),
    ("acmgcnp_v0_s1_wide", "acmgcnp", 0, 1, 7, 64, {}),
    ("acmgcnp_v1_s1_out2", "acmgcnp", 1, 1, 64, 2, {}),
    ("acmgcnpp_v0_s0", "acmgcnpp", 0, 0, 12, 16, {}),
    ("acmsgc", "acmsgc", 0, 0, 12, 7, {}),
]

MODEL_CASES = [
    ("acmgcn_v0_do0", "acmgcn", 0, 0, 0.0),
    ("acmgcn_v1_do5", "acmgcn", 1, 0, 0.5),
    ("acmgcnp_v1_s0_do5", "acmgcnp", 1, 0, 0.5),
    ("acmgcnp_v0_s1_do0", "acmgcnp", 0, 1, 0.0),
    ("acmgcnpp_v0_s0_do5", "acmgcnpp", 0, 0, 0.5),
    ("acmgcnpp_v1_s1_do5", "acmgcnpp", 1, 1, 0.5),
]


# ---------------------------------------------------------------- ACM-Pytorch
def part_pytorch():
    sys.path.insert(0, os.path.join(REF, "ACM-Pytorch"))
    sys.modules["google_drive_downloader"] = types.SimpleNamespace(GoogleDriveDownloader=object)
    os.chdir(os.path.join(REF, "ACM-Pytorch"))
    from models.layers import GraphConvolution
    from models.models import GCN
    import utils as ref_utils

    n = 96
    a_sp = small_graph(n, seed=11)
    a_un = ref_utils.sparse_mx_to_torch_sparse_tensor(a_sp)
    # filter construction exactly as train_prep does it (utils.py:619-629)
    adj_low = ref_utils.normalize_tensor(torch.eye(n) + a_un.to_dense())
    adj_high = (torch.eye(n) - adj_low).to_sparse()
    graph = (adj_low, adj_high, a_un)
    g = {}
    g.update(csr_pack("adj_low", adj_low))
    g.update(csr_pack("adj_high", adj_high))
    g.update(csr_pack("adj_un", a_un))
    # k-hop operator for ACM-SGC (utils.py:631-637), hops = 3
    a_exp = adj_low.clone()
    for _ in range(2):
        a_exp = torch.mm(a_exp, adj_low)
    g["adj_low_pow3_dense"] = a_exp.numpy()
    # feature row-normalisation (utils.py:612-617)
    feats = torch.rand(n, 9, generator=torch.Generator().manual_seed(5))
    feats[7] = 0.0                                           # zero row -> inf -> 0
    g["feat_raw"] = feats.numpy()
    g["feat_rownorm"] = ref_utils.normalize_tensor(feats.clone()).numpy()
    np.savez_compressed(os.path.join(OUT, "graph_pytorch.npz"), **g)

    seed = 100
    cases = list(LAYER_CASES) + [("acmgcn+_v0_s0_LNlive", "acmgcn+", 0, 0, 12, 16, {})]
    for tag, mt, v, s, fi, fo, extra in cases:
        cfg = dict(dialect="pytorch", model_type=mt, variant=v, structure_info=s,
                   f_in=fi, f_out=fo, attn_layernorm=int(mt in ("acmgcn+", "acmgcn++")), **extra)
        dump_layer_case(os.path.join(OUT, f"layer_pytorch_{tag}.npz"), GraphConvolution,
                        cfg, graph, seed)
        seed += 1
    for tag, mt, v, s, do in MODEL_CASES:
        cfg = dict(dialect="pytorch", model_type=mt, variant=v, structure_info=s, f_in=20,
                   hidden=16, classes=4, dropout=do, attn_layernorm=0)
        dump_model_case(os.path.join(OUT, f"model_pytorch_{tag}.npz"), GCN, cfg, graph, seed)
        seed += 1

    # --- Cora: 10-step Adam trajectory through the reference's own train_model
    adj_un, features, labels = ref_utils.load_full_data("cora")
    features = ref_utils.normalize_tensor(features)
    nn_ = labels.shape[0]
    adj_low = ref_utils.normalize_tensor(torch.eye(nn_) + adj_un.to_dense())
    adj_high = (torch.eye(nn_) - adj_low).to_sparse()
    train_mask, val_mask, test_mask = ref_utils.data_split(0, "cora")
    fx = sp.csr_matrix(features.numpy())
    cora = {"n": nn_, "labels": labels.numpy().astype(np.int64),
            "feat_indptr": fx.indptr.astype(np.int32), "feat_indices": fx.indices.astype(np.int32),
            "feat_vals": fx.data.astype(np.float32), "feat_dim": fx.shape[1],
            "train_mask": train_mask.numpy(), "val_mask": val_mask.numpy(),
            "test_mask": test_mask.numpy()}
    cora.update(csr_pack("adj_low", adj_low.to_sparse()))
    cora.update(csr_pack("adj_high", adj_high))
    cora.update(csr_pack("adj_un", adj_un))
    np.savez_compressed(os.path.join(OUT, "graph_cora.npz"), **cora)

    for tag, mt, s, opt_name in (("acmgcn_adam", "acmgcn", 0, "adam"),
                                 ("acmgcnp_s1_adam", "acmgcnp", 1, "adam")):
        torch.manual_seed(7)
        model = GCN(nfeat=features.shape[1], nhid=16, nclass=int(labels.max()) + 1, nlayers=2,
                    nnodes=nn_, dropout=0.0, model_type=mt, structure_info=s, variant=0)
        rec = {"cfg": json.dumps(dict(dialect="pytorch", model_type=mt, structure_info=s,
                                      variant=0, hidden=16, lr=0.01, weight_decay=5e-5,
                                      optimizer=opt_name, steps=10, attn_layernorm=0))}
        for name, p in model.named_parameters():
            if name not in ("fea_param", "xX_param"):
                rec["param:" + name] = p.detach().numpy().copy()
        # the two never-initialised 1x1 parameters take part in weight decay but not in the output
        with torch.no_grad():
            model.fea_param.zero_()
            model.xX_param.zero_()
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-5)
        losses = []
        for _ in range(10):
            _, loss = ref_utils.train_model(model, opt, adj_low, adj_high,
                                            adj_un if s else None, features, labels,
                                            train_mask, torch.nn.NLLLoss(), "cora")
            losses.append(loss)
        model.eval()
        with torch.no_grad():
            logits = model(features, adj_low, adj_high, adj_un if s else None)
        rec["losses"] = np.asarray(losses, dtype=np.float64)
        rec["final_logits"] = logits.numpy()
        np.savez_compressed(os.path.join(OUT, f"traj_cora_{tag}.npz"), **rec)


# ---------------------------------------------------------------- ACM-Geometric
def part_geometric():
    for m in ("dgl", "dgl.function", "dgl.utils", "dgl.nn", "dgl.nn.pytorch"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["dgl"].function = sys.modules["dgl.function"]
    sys.modules["dgl"].utils = sys.modules["dgl.utils"]
    sys.modules["dgl"].nn = sys.modules["dgl.nn"]
    sys.modules["dgl.nn"].pytorch = sys.modules["dgl.nn.pytorch"]
    sys.modules["torch_sparse"] = types.SimpleNamespace(SparseTensor=object, matmul=None)
    sys.path.insert(0, os.path.join(REF, "ACM-Geometric"))
    from layers import GraphConvolution
    from models import GCN
    import utils as ref_utils

    n = 96
    a_sp = small_graph(n, seed=23)
    # filter construction as ACM-Geometric/train.py:75-81
    low = ref_utils.normalize_tensor(sp.identity(n) + a_sp)
    high = sp.identity(n) - low
    adj_low = ref_utils.sparse_mx_to_torch_sparse_tensor(low)
    adj_high = ref_utils.sparse_mx_to_torch_sparse_tensor(high)
    a_un = ref_utils.sparse_mx_to_torch_sparse_tensor(a_sp)
    graph = (adj_low, adj_high, a_un)
    g = {}
    g.update(csr_pack("adj_low", adj_low))
    g.update(csr_pack("adj_high", adj_high))
    g.update(csr_pack("adj_un", a_un))
    feats = sp.csr_matrix(np.random.default_rng(3).standard_normal((n, 7)))
    g["feat_raw"] = feats.toarray()
    g["feat_rownorm"] = np.asarray(ref_utils.normalize_tensor(feats).todense())
    np.savez_compressed(os.path.join(OUT, "graph_geometric.npz"), **g)

    seed = 300
    for tag, mt, v, s, fi, fo, extra in LAYER_CASES:
        cfg = dict(dialect="geometric", model_type=mt, variant=v, structure_info=s, f_in=fi,
                   f_out=fo, attn_layernorm=int(mt in ("acmgcnp", "acmgcnpp")), **extra)
        dump_layer_case(os.path.join(OUT, f"layer_geometric_{tag}.npz"), GraphConvolution,
                        cfg, graph, seed)
        seed += 1
    for tag, mt, v, s, do in MODEL_CASES:
        cfg = dict(dialect="geometric", model_type=mt, variant=v, structure_info=s, f_in=20,
                   hidden=16, classes=4, dropout=do,
                   attn_layernorm=int(mt in ("acmgcnp", "acmgcnpp")))
        dump_model_case(os.path.join(OUT, f"model_geometric_{tag}.npz"), GCN, cfg, graph, seed)
        seed += 1

    # AdamW trajectory (train.py:112-136) on the small graph, LayerNorm live, dropout 0
    torch.manual_seed(9)
    gen = torch.Generator().manual_seed(10)
    model = GCN(nfeat=7, nhid=64, nclass=2, nlayers=2, nnodes=n, dropout=0.0,
                model_type="acmgcnp", structure_info=0, variant=1)
    x = torch.randn(n, 7, generator=gen)
    labels = torch.randint(0, 2, (n,), generator=gen)
    train_idx = torch.randperm(n, generator=gen)[: n // 2]
    rec = {"cfg": json.dumps(dict(dialect="geometric", model_type="acmgcnp", structure_info=0,
                                  variant=1, hidden=64, lr=0.01, weight_decay=1e-3,
                                  optimizer="adamw", steps=10, attn_layernorm=1)),
           "x": x.numpy(), "labels": labels.numpy(), "train_idx": train_idx.numpy()}
    for name, p in model.named_parameters():
        if name not in ("fea_param", "xX_param"):
            rec["param:" + name] = p.detach().numpy().copy()
    with torch.no_grad():
        model.fea_param.zero_()
        model.xX_param.zero_()
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    losses = []
    import torch.nn.functional as F
    for _ in range(10):
        model.train()
        opt.zero_grad()
        out = F.log_softmax(model(x, adj_low, adj_high, None), dim=1)
        loss = torch.nn.NLLLoss()(out[train_idx], labels[train_idx])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    model.eval()
    with torch.no_grad():
        rec["final_logits"] = model(x, adj_low, adj_high, None).numpy()
    rec["losses"] = np.asarray(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "traj_geometric_acmgcnp_adamw.npz"), **rec)


# ---------------------------------------------------------------- real graph structures
def part_graphs():
    """Real structures (data, not code): Chameleon + Squirrel edge lists as
    undirected CSR of the raw adjacency; Squirrel features/labels rebuilt from
    the bundled MUSAE json/csv."""
    for name in ("chameleon", "squirrel"):
        edges = np.loadtxt(os.path.join(REF, "new_data", name, "out1_graph_edges.txt"),
                           skiprows=1, dtype=np.int64)
        n = int(edges.max()) + 1
        a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
        a = ((a + a.T) > 0).astype(np.float32).tocsr()       # nx.Graph semantics: undirected, simple
        a.sort_indices()
        rec = {"n": n, "adj_un_indptr": a.indptr.astype(np.int32),
               "adj_un_indices": a.indices.astype(np.int32)}
        if name == "squirrel":
            with open(os.path.join(REF, "new_data", name, "squirrel_features.json")) as f:
                fj = json.load(f)
            rows, cols = [], []
            for k, v in fj.items():
                rows += [int(k)] * len(v)
                cols += list(v)
            fdim = max(cols) + 1
            fx = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, fdim))
            fx.sum_duplicates()
            fx.data[:] = 1.0
            tgt = np.loadtxt(os.path.join(REF, "new_data", name, "squirrel_target.csv"),
                             delimiter="\t", skiprows=1, dtype=np.int64)
            with open(os.path.join(REF, "new_data", name, "class_map.json")) as f:
                cmap = json.load(f)
            labels = np.array([cmap[str(i)] for i in range(n)], dtype=np.int64)
            rec.update(feat_indptr=fx.indptr.astype(np.int32),
                       feat_indices=fx.indices.astype(np.int32), feat_dim=fdim, labels=labels,
                       target_raw=tgt[:, 1])
            for i in range(10):
                with np.load(os.path.join(REF, "ACM-Pytorch", "splits",
                                          f"squirrel_split_0.6_0.2_{i}.npz")) as s:
                    rec[f"train_mask_{i}"] = np.packbits(s["train_mask"].astype(bool))
                    rec[f"val_mask_{i}"] = np.packbits(s["val_mask"].astype(bool))
                    rec[f"test_mask_{i}"] = np.packbits(s["test_mask"].astype(bool))
        np.savez_compressed(os.path.join(OUT, f"graph_{name}.npz"), **rec)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", default="all", choices=["all", "pytorch", "geometric", "graphs"])
    a = ap.parse_args()
    if a.part == "all":
        for part in ("pytorch", "geometric", "graphs"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--part", part])
    else:
        {"pytorch": part_pytorch, "geometric": part_geometric, "graphs": part_graphs}[a.part]()
        print("golden part", a.part, "written to", OUT)
