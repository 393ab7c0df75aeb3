#!/usr/bin/env python3
"""Host-side (Python) cost of one eager train step of the bench workload's model.

    python scripts/profile_host.py                 on the GPU: cProfile over 100 eager steps (twitch-shaped graph)
    python scripts/profile_host.py --null [...]    GPU-less: the package runs on CPU tensors over tests/fake_lib.py with every
                                                   compute entry point replaced by ``return 0`` -- what is timed is ONLY the host
                                                   path (argument marshalling, autograd, allocation), on a small graph
      --route trainstep | loop     train.TrainStep (default) or the reference's loop body (ACM-Geometric/train.py:119-137)
      --profile                    cProfile table instead of the wall-clock figure
      --model acmgcnp --structure 0|1 --variant 0|1
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class _Patch:
    """The two methods of pytest's monkeypatch that tests/fake_lib.install uses."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def null_library():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_lib
    fake = fake_lib.install(_Patch())
    keep = ("acm_version", "acm_tuning_get", "acm_tuning_set", "acm_shard_plan", "acm_last_error", "acm_csr_",
            "acm_acmii_table_bytes")
    for name in dir(fake):
        if name.startswith("acm_") and not name.startswith(keep) and not name.endswith("_workspace_bytes"):
            setattr(fake, name, lambda *a: 0)
    return fake


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--null", action="store_true")
    ap.add_argument("--route", default="trainstep")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--model", default="acmgcnp")
    ap.add_argument("--structure", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    if args.null:
        null_library()
    import acm_gnn_amd
    from acm_gnn_amd import data as D, distributed as DD, train as T
    dev = torch.device("cpu" if args.null else "cuda:0")
    sync = (lambda: None) if args.null else torch.cuda.synchronize
    if args.null:                                        # a small graph of the same class: 7 features, pattern-only operator
        import scipy.sparse as sp
        rng = np.random.default_rng(0)
        n = 4096
        a = sp.random(n, n, density=8.0 / n, random_state=1, format="csr", dtype=np.float32)
        adj = ((a + a.T) > 0).astype(np.float32).tocsr()
        x_np = np.abs(rng.standard_normal((n, 7))).astype(np.float32)
        y_np = rng.integers(0, 2, n).astype(np.int64)
        tr = np.sort(rng.permutation(n)[: n // 2])
    else:
        adj, x_np, y_np, (tr, va, te), n = D.synthetic_dataset("twitch-gamer")
    if not args.structure:
        x_np = D.row_normalize_features(x_np)
    low, deg = D.build_filters(adj)
    ops = DD.make_sharded_operators(low, deg, dev)
    x, y = torch.from_numpy(x_np).to(dev), torch.from_numpy(y_np).to(dev)
    torch.manual_seed(0)
    model = acm_gnn_amd.GCN(7, 64, 2, 2, n, 0.1, args.model, args.structure, variant=bool(args.variant)).to(dev)
    idx = torch.from_numpy(tr).to(dev)
    if args.route == "trainstep":
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
        step = T.TrainStep(model, opt, x, ops, y, T.row_weights(idx, n), small_step=False)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=0.05, weight_decay=1e-3)

        def step():                                      # ACM-Geometric/train.py:119-137
            model.train()
            opt.zero_grad()
            out = F.log_softmax(model(x, ops, None, None), dim=1)
            loss = F.nll_loss(out[idx], y[idx])
            loss.backward()
            opt.step()
            return loss
    for _ in range(10):
        step()
    sync()
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(100):
            step()
        pr.disable()
        sync()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(args.top)
        return
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        best = min(best, (time.perf_counter() - t) / args.steps * 1e3)
    print(f"{args.route} {args.model} structure={args.structure} variant={args.variant}"
          f"{' null library (host path only)' if args.null else ''}: {best:.3f} ms per eager step")


if __name__ == "__main__":
    main()
