#!/usr/bin/env python3
"""Run ON THE GPU BOX: what ONE GPU can measure of the N-rank row-sharded training step.

For every rank r of an N-rank plan of bench.py's workload (the twitch-gamer-shaped graph, degree ranking dealt to the
ranks like cards + equal blocks, replicated static input => the input pipeline runs sharded) this script builds rank r's
operators from ITS OWN rows and runs rank r's training step ALONE on the device, one rank after the other, with the
collectives replaced by local stand-ins of the same shapes (an all-gather copies the local block into every slot, an
all-reduce is a no-op): the kernels a rank launches, their HIP-event times, the launches per step and the bytes it would
send are exactly those of the real run -- only the values in the halo slots are not, which no kernel's time depends on --
and no other process shares the GPU while a rank is measured.  (Two ranks on one GPU against the single process are
checked for RESULTS in tests/test_gpu_sharded.py; RCCL itself needs N devices.)

    python scripts/shard8_per_rank.py [--world 8] [--steps 5] > gpurun_out/r04_shard8_per_rank.json

Round 5: ``--dataset arxiv-year|penn94 --method acmsgc --hops 3`` measures BASELINE config 5 (3-hop
ACM-SGC on the arXiv-year / Penn94 shapes: one linear ACM layer whose low channel goes through A_low three times -- three
halo exchanges forward, three backward), ``--dataset pokec`` the pokec-shaped graph (1.63 M nodes: the size at which a
rank's gathered table -- the FULL all-gathered halo -- no longer fits the Infinity Cache); ``--world`` takes a list.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acm_gnn_amd  # noqa: E402
from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T  # noqa: E402


class SoloGroup:
    """Stand-in for a process group of `world` ranks seen from rank `rank` (no peers)."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank


def install_stand_ins():
    real = {k: getattr(dist, k) for k in ("get_world_size", "get_rank", "get_backend", "all_gather_into_tensor", "all_reduce",
                                          "is_initialized")}

    def get_world_size(group=None):
        return group.world if isinstance(group, SoloGroup) else real["get_world_size"](group)

    def get_rank(group=None):
        return group.rank if isinstance(group, SoloGroup) else real["get_rank"](group)

    def get_backend(group=None):
        return "gloo" if isinstance(group, SoloGroup) else real["get_backend"](group)

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        if not isinstance(group, SoloGroup):
            return real["all_gather_into_tensor"](out, inp, group=group, async_op=async_op)
        out.view(group.world, -1).copy_(inp.reshape(1, -1).expand(group.world, -1))      # same bytes written as received

    def all_reduce(t, op=None, group=None, async_op=False):
        if not isinstance(group, SoloGroup):
            return real["all_reduce"](t, group=group, async_op=async_op)

    dist.get_world_size, dist.get_rank, dist.get_backend = get_world_size, get_rank, get_backend
    dist.all_gather_into_tensor, dist.all_reduce = all_gather_into_tensor, all_reduce


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="8", help="ranks of the plan; a comma-separated list measures every N (one JSON line each)")
    ap.add_argument("--method", default="acmgcnp")
    ap.add_argument("--hops", type=int, default=1)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dataset", default="twitch-gamer")
    ap.add_argument("--ranks", default="", help="comma-separated ranks to measure (default: all)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    install_stand_ins()
    for world in [int(w) for w in str(args.world).split(",")]:
        measure(args, world, dev)


def pokec_workload(world):
    """The pokec-shaped graph of scripts/bench_scale.py (1 632 803 nodes, 30.6 M edges, 65 features, 2 classes), padded to
    a multiple of `world` rows, in degree order."""
    import scipy.sparse as sp
    n, m, max_deg, f_in, classes = 1_632_803, 30_622_564, 14_854, 65, 2
    n += (-n) % world
    adj = D.chung_lu_graph(n, m, max_deg, seed=0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, f_in), dtype=np.float32)
    y = rng.integers(0, classes, n)
    tr = np.sort(rng.permutation(n)[: n // 2])
    perm = D.degree_order(adj)
    adj, x, y, (tr, va, te) = D.permute_dataset(adj, x, y, (tr, tr, tr), perm)
    return {"adj": adj, "x": x, "y": y, "splits": (tr, va, te)}


def measure(args, world, dev):
    wl = pokec_workload(world) if args.dataset == "pokec" else D.bench_workload(args.dataset, seed=0, node_order="degree", pad_to=world)
    adj, x_np, y_np, (tr, va, te) = wl["adj"], wl["x"], wl["y"], wl["splits"]
    n = adj.shape[0]
    adj, x_np, y_np, (tr, va, te) = D.permute_dataset(adj, x_np, y_np, (tr, va, te), DD.interleave_order(n, world))
    low, deg = D.build_filters(adj)
    plan = DD.equal_rows_plan(n, world)
    low = low.tocsr()
    low.sort_indices()
    per_rank = []
    for rank in ([int(r) for r in args.ranks.split(",")] if args.ranks else range(world)):
        b, e = plan.rows(rank)
        loc = low[b:e].tocsr()
        s = (1.0 / deg[b:e]).astype(np.float32)                      # A_low = D^-1 (A + I): one value per row
        group = SoloGroup(world, rank)
        ops = DD.make_sharded_operators_from_rows(loc.indptr, loc.indices, loc.data, deg[b:e], plan, rank, dev, group=group,
                                                  _form=(loc.indptr.astype(np.int32), loc.indices.astype(np.int32), s))
        assert ops.sharded and ops.uniform and ops.implicit
        ops.x_full = torch.from_numpy(np.ascontiguousarray(x_np)).to(dev)
        x = torch.from_numpy(np.ascontiguousarray(x_np[b:e])).to(dev)
        y = torch.from_numpy(np.ascontiguousarray(y_np[b:e])).to(dev)
        tr_loc = torch.from_numpy(DD.local_index(tr, plan, rank)).to(dev)
        torch.manual_seed(0)
        ops.hops = args.hops
        model = acm_gnn_amd.GCN(x.shape[1], 64, int(y_np.max()) + 1, 2, e - b, args.dropout, args.method, 0, variant=False,
                                attn_layernorm=True).to(dev)
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.05, weight_decay=1e-3)
        w = T.row_weights(tr_loc, e - b, n_train_total=len(tr), device=dev)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=False, fused_dropout=True)
        for _ in range(5):
            step()
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        for _ in range(args.steps):
            step()
        AF.set_kernel_timer(None)
        summ = timer.summary()
        coll = {k: v for k, v in summ.items() if k.startswith(("all_gather", "all_reduce"))}
        kern = {k: v for k, v in summ.items() if k not in coll}
        # the same step replayed as a hipGraph (the stand-in copies are captured like the collectives would be)
        gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=True)
        for _ in range(3):
            gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gstep()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / 50 * 1e3
        per_rank.append({
            "rank": rank, "rows": int(e - b), "nnz": int(loc.nnz), "input_pipeline": step.pipe is not None,
            "kernels_ms_per_step": round(sum(v[1] for v in kern.values()) / args.steps, 4),
            "kernel_launches_per_step": sum(v[0] for v in kern.values()) // args.steps,
            "kernel_us": {k: round(v[1] / v[0] * 1e3, 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
            "collective_calls_per_step": sum(v[0] for v in coll.values()) // args.steps,
            "collective_bytes_sent_per_step": int(sum(timer.bytes.values()) // args.steps),
            "stand_in_collectives_ms_per_step": round(sum(v[1] for v in coll.values()) / args.steps, 4),
            "captured_step_ms_with_stand_in_collectives": round(graph_ms, 4)})
        del step, gstep, model, opt, ops, x, y, w
        torch.cuda.empty_cache()
    ks = [r["kernels_ms_per_step"] for r in per_rank]
    print(json.dumps({"workload": f"{args.dataset}-shaped graph, {n} nodes, nnz(A_low) {low.nnz}; {args.method}"
                                  + (f" {args.hops}-hop" if args.hops > 1 else "") + f", hidden 64, dropout {args.dropout} "
                                  "(counter-based), AdamW (fused)",
                      "world": world, "plan": "degree ranking dealt to the ranks like cards, equal blocks, replicated static input",
                      "method": "each rank's step run alone on one MI355X, collectives replaced by local stand-ins of the same "
                                "shapes (scripts/shard8_per_rank.py)",
                      "kernels_ms_per_step": {"min": min(ks), "max": max(ks), "mean": round(float(np.mean(ks)), 4)},
                      "per_rank": per_rank}), flush=True)


if __name__ == "__main__":
    main()
