#!/usr/bin/env python3
"""Reference accuracy per fixed split, recorded by TRAINING THE IMPORTED REFERENCE in the build
container (ACM-Pytorch dialect: dense A_low, attention LayerNorm dead), with seeded CPU init and
the deterministic dropout masks of tests/replay.py, so tests/test_gpu_accuracy.py can replay the
exact same experiment on the MI355X.

    python tests/golden/make_accuracy_golden.py cora      # ~6 min
    python tests/golden/make_accuracy_golden.py squirrel  # ~10 min

The loop follows ACM-Pytorch/train.py:95-139: train_model(), eval forward, keep test acc at the
lowest validation loss, early stop when val_loss > mean of the last `early_stopping` epochs.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from replay import SeededDropout  # noqa: E402

REF = "/root/reference"
CONFIGS = {
    "cora": dict(model="acmgcn", structure_info=0, variant=0, hidden=64, lr=0.01, weight_decay=5e-5, dropout=0.6,
                 epochs=300, early_stopping=200, splits=list(range(10))),
    "squirrel": dict(model="acmgcnp", structure_info=1, variant=0, hidden=64, lr=0.002, weight_decay=1e-4,
                     dropout=0.6, epochs=250, early_stopping=200, splits=[0, 1, 2]),
}


def main(name):
    cfg = CONFIGS[name]
    sys.path.insert(0, os.path.join(REF, "ACM-Pytorch"))
    sys.modules["google_drive_downloader"] = types.SimpleNamespace(GoogleDriveDownloader=object)
    os.chdir(os.path.join(REF, "ACM-Pytorch"))
    import torch.nn.functional as F
    from models.models import GCN
    import utils as U

    if name == "cora":
        adj_un, features, labels = U.load_full_data("cora")
    else:
        g = np.load(os.path.join(HERE, "graph_squirrel.npz"))
        n = int(g["n"])
        a = sp.csr_matrix((np.ones(len(g["adj_un_indices"]), np.float32), g["adj_un_indices"], g["adj_un_indptr"]),
                          shape=(n, n))
        adj_un = U.sparse_mx_to_torch_sparse_tensor(a)
        fx = sp.csr_matrix((np.ones(len(g["feat_indices"]), np.float32), g["feat_indices"], g["feat_indptr"]),
                           shape=(n, int(g["feat_dim"])))
        features = torch.FloatTensor(fx.toarray())
        labels = torch.LongTensor(g["labels"])
    if not (cfg["model"] in ("acmgcnp", "acmgcnpp") and cfg["structure_info"]):
        features = U.normalize_tensor(features)
    n = labels.shape[0]
    adj_low = U.normalize_tensor(torch.eye(n) + adj_un.to_dense())
    adj_high = (torch.eye(n) - adj_low).to_sparse()
    adj_unn = adj_un if cfg["structure_info"] else None
    out = {"cfg": json.dumps(dict(cfg, dataset=name, dialect="pytorch", attn_layernorm=0, optimizer="adam"))}
    accs = []
    for split in cfg["splits"]:
        tr, va, te = U.data_split(split, name)
        torch.manual_seed(1000 + split)
        model = GCN(nfeat=features.shape[1], nhid=cfg["hidden"], nclass=int(labels.max()) + 1, nlayers=1, nnodes=n,
                    dropout=cfg["dropout"], model_type=cfg["model"], structure_info=cfg["structure_info"],
                    variant=cfg["variant"])
        with torch.no_grad():
            model.fea_param.zero_()
            model.xX_param.zero_()
        opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        drop = SeededDropout(seed=split)
        real = F.dropout
        F.dropout = drop
        best_val, curr, hist = float("inf"), 0.0, []
        try:
            for epoch in range(cfg["epochs"]):
                drop.next_epoch()
                _, loss_train = U.train_model(model, opt, adj_low, adj_high, adj_unn, features, labels, tr,
                                              torch.nn.NLLLoss(), name)
                model.eval()
                with torch.no_grad():
                    o = F.log_softmax(model(features, adj_low, adj_high, adj_unn), dim=1)
                    val_loss = float(F.nll_loss(o[va], labels[va]))
                    test_acc = float(U.accuracy(labels[te], o[te]))
                hist.append((loss_train, val_loss, test_acc))
                if val_loss < best_val:
                    best_val, curr = val_loss, test_acc
                if cfg["early_stopping"] > 0 and epoch > cfg["early_stopping"]:
                    if val_loss > np.mean([h[1] for h in hist[epoch - cfg["early_stopping"]:epoch]]):
                        break
        finally:
            F.dropout = real
        accs.append(curr)
        out[f"hist_{split}"] = np.asarray(hist, dtype=np.float64)
        print(f"{name} split {split}: test acc {curr:.4f} after {len(hist)} epochs", flush=True)
    out["test_acc"] = np.asarray(accs)
    np.savez_compressed(os.path.join(HERE, f"accuracy_{name}.npz"), **out)
    print(f"{name}: {100 * np.mean(accs):.2f} +- {100 * np.std(accs):.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
