#!/usr/bin/env python3
"""Benchmark: edges/sec of one full-batch ACM-GCN training step on a twitch-gamer-shaped graph.

    python bench.py [--gpus N --steps K --warmup W]          # N > 1: starts its own ranks (one per GPU, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or is started as ranks

A "step" = zero_grad + 2-layer forward + log-softmax/NLL loss + backward + AdamW update on the
whole graph (the reference's hot loop, ACM-Geometric/train.py:119-136, without its per-epoch
eval pass).  value = nnz(A_low) / t_step summed over the job.  With N > 1
the CSR rows are sharded over the ranks (strong scaling: the graph is fixed) and the halo
features are all-gathered over RCCL.  Inputs are resident in HBM before the timed region.

Rank 0 prints one JSON line (contract in the task statement) that also carries
  roofline     -- algorithmic bytes / HIP-event time of the dominant kernel vs the 8 TB/s HBM peak
  cpu_baseline -- the oracle's literal restatement of the reference step on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured stream)
FP32_MFMA_PEAK_TF = 157.3
TRAFFIC_FILE = "r06_pmc_traffic.json"        # profiles/: HBM bytes per launch from the rocprofv3 --pmc passes (collect_profiles.sh)
WINDOWS = 5                    # timed windows of --steps replays each; ms_per_step is the median window


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dataset", default="twitch-gamer")
    ap.add_argument("--method", default="acmgcnp", choices=["acmgcn", "acmgcnp", "acmgcnpp"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--structure_info", type=int, default=0)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--weight_decay", type=float, default=1e-3)
    ap.add_argument("--uniform", action="store_true", help="uniform random graph instead of power-law")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout-impl", default="fused", choices=["fused", "torch"],
                    help="dropout masks: counter-based inside the layer kernels, or torch's F.dropout tensors")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="AdamW update: acm_adam_step (one launch) or torch.optim.AdamW (~80 launches)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", type=int, default=-1,
                    help="1 (default): also time hipGraph replays of the captured step and report those; 0: eager only")
    ap.add_argument("--node-order", default="degree", choices=["degree", "random"],
                    help="node relabelling applied to the whole dataset before training (data prep)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take the row-sharded path (RCCL collectives inside the captured step) even with one rank; needs a "
                         "torch.distributed launch")
    ap.add_argument("--steps-per-graph", type=int, default=-1,
                    help="optimizer steps captured per hipGraph (default 1; must divide --steps)")
    ap.add_argument("--traffic-json", default=None,
                    help="pmc_traffic.json of scripts/make_traffic_json.py to quote roofline.traffic from (default: the committed "
                         "profiles/ file; either is quoted only while the kernel sources hash to what it was collected on)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements of the single-GPU run (literal form, random node order)")
    ap.add_argument("--no-check", action="store_true", help="skip the comparisons with the CPU oracle (eval-mode logits on "
                    "sampled rows; one training-mode step: loss and every parameter gradient)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print (as one JSON line) the command and environment bench.py would start its ranks with, then exit; "
                         "{\"launch\": null} when this process would run the benchmark itself")
    return ap.parse_args()


def self_launch(args, argv=None, environ=None):
    """``python bench.py --gpus N`` WITHOUT a torch.distributed launcher around it (WORLD_SIZE unset): the command that starts
    one rank per GPU on this node -- torch.distributed.run over RCCL, rendezvous on 127.0.0.1 (the container's hostname may
    not resolve) on a free port -- with this very command line, or None when this process is the benchmark: already a rank
    (WORLD_SIZE set), or one GPU without --force-sharded.  Rank 0 of the started job prints the one JSON line; the launcher
    process is replaced by torch.distributed.run (os.execvpe), so its exit status is the job's.

    (Round 5: such a call exited with "must be launched through torch.distributed.run" -- a driver that runs the N = 1 command
    with another N would have got no scaling point.)"""
    environ = os.environ if environ is None else environ
    argv = [a for a in (sys.argv[1:] if argv is None else argv) if a != "--dry-run-launch"]
    if "WORLD_SIZE" in environ or not (args.gpus > 1 or args.force_sharded):
        return None
    import socket
    with socket.socket() as sock:                       # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max(args.gpus, 1)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),   # dmabuf IPC (RCCL needs it here)
           "OMP_NUM_THREADS": environ.get("OMP_NUM_THREADS", str(max((os.cpu_count() or 8) // max(args.gpus, 1), 1)))}
    return cmd, env


def algorithmic_bytes(label, n, nnz, implicit=False):
    """HBM bytes a launch must move if every operand is touched exactly once.
    Graph term: 4(N+1) + 8 nnz for (column id, value) pairs; 4(N+1) + 4 nnz + 4 N for the pattern-only
    form (ids + one scale per row) -- SURVEY.md section 8(d)."""
    per_edge, per_row = (4, 4) if implicit else (8, 0)
    kind, _, shape = label.partition("/")
    if kind == "nll_loss":
        rows, c = (int(v) for v in shape.split("x"))
        return rows * (4 * c + 8 + 4 + 4 * c)
    if kind.startswith("conv_agg"):
        f = int(shape[1:shape.index("k")])
        fi = int(shape[shape.index("i") + 1:])
        fp = 4 if fi <= 4 else (8 if fi <= 8 else 16)
        k = int(shape[shape.index("k") + 1:shape.index("i")])
        stats = 16 * k * n              # head_stats: mean | rstd | sigmoid | alpha per channel, written by the forward
        if kind == "conv_agg_fwd":      # graph + gathered X once + self X + out + agg + att + head_stats
            return 4 * (n + 1) + per_edge * nnz + per_row * n + 4 * n * fp * 2 + 4 * n * f + 4 * n * fp + 16 * n + stats
        if kind == "conv_agg_epi":      # the row-local stage alone (P given): agg + self X + out + att + head_stats
            return 4 * n * fp * 2 + 4 * n * f + 16 * n + stats
        bwd = 4 * n * (f + 2 * fp) + stats      # conv_agg_bwd: grad_out, agg, X, head_stats
        if "+proj" in kind:                     # the following layer's projection backward inside (acm_conv_agg_bwd_t.proj_*):
            bwd += 4 * n * 6                    # `out` (F floats per row) is read in place of grad_out, + proj_dz (6 floats)
        if "+gather" in kind:                   # + the next step's P = A_low dropout(x): graph, table once, P written
            return bwd + 4 * (n + 1) + per_edge * nnz + per_row * n + 4 * n * fp * 2
        return bwd
    if kind.startswith("gemm"):
        m, nn, k = (int(v) for v in shape.split("x"))
        return 4 * (m * k + k * nn + m * nn)
    f = int(shape[1:shape.index("k")])
    k = int(shape[shape.index("k") + 1:])
    graph = 4 * (n + 1) + per_edge * nnz + per_row * n
    if kind in ("conv_fwd", "conv_fwd_tail"):   # read Z [n,3F] (+S, deg) once; write out, pre, att
        fwd = graph + 4 * n * 3 * f + (4 * n * (f + 1) if k == 4 else 0) + 4 * n * f + 4 * n * (k - 1) * f + 16 * n
        if kind == "conv_fwd":
            return fwd
        # + the loss (labels, weights, dlogits) and the layer's K3 outputs (G_L, G_H, G_I) in the same row pass
        return fwd + n * (8 + 4 + 4 * f) + 4 * n * f * k
    if kind == "conv_bwd_spmm":  # read G [n,(k-1)F] once; write dZ_L, dZ_H (+dS)
        return graph + 4 * n * (k - 1) * f + 4 * n * (k - 1) * f
    if kind == "conv_bwd_local":  # read grad_out, pre, Z_I; write G_L, G_H, G_I (+G_S)
        return 4 * n * f * (1 + (k - 1) + 1) + 4 * n * f * k
    return 0


def main():
    args = parse()
    launch = self_launch(args)
    if args.dry_run_launch:
        print(json.dumps({"launch": None if launch is None else {"cmd": launch[0], "env": launch[1]}}))
        return
    if launch is not None:
        cmd, env = launch
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, dict(os.environ, **env))
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                     # started as ranks by someone else's launcher: its world size is the job's
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_sharded = bool(args.force_sharded)                              # exercise the RCCL path with one rank
    if world > 1 or (force_sharded and "RANK" in os.environ):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import acm_gnn_amd
    from acm_gnn_amd import data as D, distributed as DD, functional as AF, train as T

    # ---------------- data (synthetic, seeded; identical on every rank) ----------------
    t0 = time.time()
    wl = D.bench_workload(args.dataset, seed=args.seed, node_order=args.node_order, uniform=args.uniform, pad_to=world,
                          normalize_features=not (args.method in ("acmgcnp", "acmgcnpp") and args.structure_info))
    adj, x_np, y_np, (tr, va, te), n_real, low, deg = (wl[k] for k in ("adj", "x", "y", "splits", "n_real", "low", "deg"))
    n_glob = adj.shape[0]
    nnz = int(low.nnz)
    sharded = world > 1 or (force_sharded and dist.is_initialized())
    plan, shard = None, None
    if sharded:
        if args.node_order == "degree":
            # deal the degree ranking to the ranks like cards: equal contiguous blocks are then balanced in rows AND
            # in nnz (a contiguous cut of the ranking itself gives rank 0 the hubs: 5.1x the mean nnz at 8 ranks)
            adj, x_np, y_np, (tr, va, te) = D.permute_dataset(adj, x_np, y_np, (tr, va, te),
                                                              DD.interleave_order(n_glob, world))
            low, deg = D.build_filters(adj)
            plan = DD.equal_rows_plan(n_glob, world)
        else:
            plan = DD.shard_plan(low.indptr, world)          # acm_shard_plan: nnz + row_cost * rows balanced
        rows_r, nnz_r, _ = plan.work(low.indptr)
        r_nnz, r_work = plan.imbalance(low.indptr, DD.DEFAULT_ROW_COST)
        shard = {"plan": "equal blocks of the degree ranking dealt cyclically to the ranks" if args.node_order == "degree"
                 else f"acm_shard_plan(row_cost={DD.DEFAULT_ROW_COST})", "rows_per_rank": rows_r.tolist(),
                 "nnz_per_rank": nnz_r.tolist(), "nnz_max_over_mean": round(r_nnz, 4),
                 "work_max_over_mean": round(r_work, 4)}
    ops = DD.make_sharded_operators(low, deg, dev, with_structure=bool(args.structure_info), plan=plan,
                                    group=dist.group.WORLD if (force_sharded and dist.is_initialized()) else None)
    b, e = plan.rows(rank) if plan is not None else (0, n_glob)
    x = torch.from_numpy(np.ascontiguousarray(x_np[b:e])).to(dev)
    if ops.sharded and ops.uniform:
        ops.x_full = torch.from_numpy(np.ascontiguousarray(x_np)).to(dev)   # replicated static input (4.7 MB): no halo
                                                                             # all-gather for the first layer
    y = torch.from_numpy(np.ascontiguousarray(y_np[b:e])).to(dev)
    tr_loc = torch.from_numpy(DD.local_index(tr, plan, rank) if plan is not None else tr).to(dev)
    n_train = len(tr)
    n_cls = int(y_np.max()) + 1
    prep_s = time.time() - t0

    torch.manual_seed(args.seed)
    model = acm_gnn_amd.GCN(x.shape[1], args.hidden, n_cls, 2, e - b, args.dropout, args.method,
                            args.structure_info, variant=bool(args.variant), attn_layernorm=True).to(dev)
    use_graph = True if args.graph < 0 else bool(args.graph)
    if args.optimizer == "fused":               # acm_adam_step: the AdamW update as one launch
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay, capturable=use_graph)
    # mean NLL over the (global) training set; rows a rank does not own have weight 0
    w = T.row_weights(tr_loc, e - b, n_train_total=n_train, device=dev)
    fused_drop = args.dropout_impl == "fused"
    step = T.TrainStep(model, opt, x, ops, y, w, use_graph=False, fused_dropout=fused_drop)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- warm-up, then 3 calibration steps with per-launch HIP events on every kernel of the
    # library (outside the timed region) to find the dominant one ----------------
    for _ in range(max(args.warmup, 1)):
        loss = step()
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    for _ in range(3):
        loss = step()
    warm = timer.summary()
    AF.set_kernel_timer(None)
    # row-sharded runs: what each rank spent per step in the library's kernels and in the collectives (HIP events of the
    # three calibration steps), and the bytes it sent -- to read a scaling line against DESIGN.md section 7's estimate
    if shard is not None:
        coll = {k: v for k, v in warm.items() if k.startswith(("all_gather", "all_reduce"))}
        mine = {"kernels_ms_per_step": round(sum(v[1] for k, v in warm.items() if k not in coll) / 3, 4),
                "collectives_ms_per_step": round(sum(v[1] for v in coll.values()) / 3, 4),
                "collective_calls_per_step": sum(v[0] for v in coll.values()) // 3,
                "collective_bytes_sent_per_step": sum(timer.bytes.values()) // 3}
        per_rank = [None] * world
        if dist.is_initialized() and world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        shard["per_rank"] = per_rank
    warm = {k: v for k, v in warm.items() if not k.startswith(("all_gather", "all_reduce"))}
    dominant = max(warm, key=lambda k: warm[k][1])
    # ---------------- timed region ----------------
    focus = AF.KernelTimer(only=dominant)
    AF.set_kernel_timer(focus)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t1
    AF.set_kernel_timer(None)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eager_ms = dt / args.steps * 1e3
    ms_per_step = eager_ms
    eager_loss = float(loss.item())

    # ---------------- roofline of the dominant kernel (from the eager timed region) ----------------
    roofline, breakdown = None, None
    if rank == 0:
        launches, total_ms = focus.summary()[dominant]
        avg_ms = total_ms / launches
        alg = algorithmic_bytes(dominant, e - b, ops.low.nnz, ops.implicit)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        # PMC traffic cannot be sampled from inside this process; the figure measured for this kernel on this
        # workload by the committed rocprofv3 --pmc passes (scripts/collect_profiles.sh) is attached when it applies
        traffic, traffic_source = None, None
        tpath = args.traffic_json or os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if world == 1 and args.dataset == "twitch-gamer" and args.node_order == "degree" and os.path.exists(tpath):
            with open(tpath) as fh:
                rec = json.load(fh)
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from make_traffic_json import kernel_source_hash
            traffic_source = (os.path.relpath(tpath, ROOT) if args.traffic_json else f"profiles/{TRAFFIC_FILE}") + \
                (f"@{rec['_commit']}" if "_commit" in rec else "")
            if rec.get("_kernel_source_hash") == kernel_source_hash():
                traffic = rec.get(dominant, {}).get("hbm_bytes")
            else:                                   # counters collected for other kernels: not quoted
                traffic_source += " (STALE: the kernel sources changed since the counters were collected)"
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": dominant, "avg_ms": round(avg_ms, 4), "launches": launches,
                    "avg_ms_source": "HIP events around the kernel's eager launches in the timed region",
                    "algorithmic_bytes": alg}
        breakdown = {k: round(v[1] / v[0], 4) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])}

    def emit(ms, launch, final_loss, with_cpu, extras=None, check=None):
        cpu = cpu_baseline(args, model, wl) if (with_cpu and world == 1 and not args.no_cpu_baseline) else None
        config = {"workload": f"{args.dataset}-shaped Chung-Lu graph: {n_real} nodes, {adj.nnz // 2} undirected "
                              f"edges, nnz(A_low)={nnz}, F_in={x.shape[1]}, hidden={args.hidden}, classes={n_cls}; "
                              f"2-layer {args.method} (variant={args.variant}, structure_info={args.structure_info}, "
                              f"attention LayerNorm on), dropout {args.dropout} ({'counter-based, masks regenerated in the kernels' if fused_drop else 'F.dropout mask tensors'}), "
                              f"AdamW ({args.optimizer}); "
                              "step = fwd + NLL loss + bwd + optimizer update",
                  "parallelism": f"csr-row-shard x{world}" if world > 1 else "single-gpu",
                  "node_order": args.node_order, "launch": launch,
                  "operator_form": "pattern-only P + row scale (shared by A_low and A_low^T)" if ops.implicit
                  else "explicit (column id, value) CSR + transposed CSR",
                  "eager_ms_per_step": round(eager_ms, 4),
                  "file_edges_per_s": round((adj.nnz // 2) / (ms * 1e-3), 1),
                  "kernel_ms": breakdown, "prep_s": round(prep_s, 1), "final_loss": final_loss}
        if shard is not None:
            config["shard"] = shard
        config.update(extras or {})
        config.update(check or {"checked": False})
        result = {
            "metric": "edges/sec ACM-GCN fwd+bwd on twitch-gamer",
            "value": round(nnz / (ms * 1e-3), 1), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": config, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(result), flush=True)

    def timed_graph_steps(gstep, spread=None):
        """ms per step of `gstep`: WINDOWS windows of exactly --steps calls, each bracketed by barrier + synchronize and
        reduced with MAX over the ranks; the MEDIAN window is reported (``spread`` receives min / median / max)."""
        per_call = int(getattr(gstep, "steps_per_call", 1))      # optimizer steps one call runs (TrainStep(steps_per_graph=...))
        assert args.steps % per_call == 0
        for _ in range(max(-(-args.warmup // per_call), 1)):
            out = gstep()
        wins = []
        for _ in range(WINDOWS):
            fence()
            t = time.perf_counter()
            for _ in range(args.steps // per_call):                # exactly --steps optimizer steps per window
                out = gstep()
            fence()
            dtg = time.perf_counter() - t
            if world > 1:
                tt = torch.tensor([dtg], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtg = float(tt.item())
            wins.append(dtg / args.steps * 1e3)
        wins.sort()
        if spread is not None:
            spread.update({"windows": WINDOWS, "steps_per_window": args.steps, "min": round(wins[0], 4),
                           "median": round(wins[len(wins) // 2], 4), "max": round(wins[-1], 4)})
        return wins[len(wins) // 2], out

    # ---------------- second timed region: the same K steps as replays of one captured HIP graph ----------------
    # (collectives included when sharded).  A watchdog makes the run fall back to the eager measurement if the
    # captured path does not finish: rank 0 then reports the eager numbers instead of hanging the job.
    graph_ok = False
    spread = {}
    if use_graph:
        import threading
        state = {"done": False}

        def fallback():
            if state["done"]:
                return
            if rank == 0:
                emit(eager_ms, "eager launches (hipGraph path timed out)", eager_loss, with_cpu=False)
            os._exit(0)

        timer_t = threading.Timer(120.0, fallback)
        timer_t.daemon = True
        timer_t.start()
        try:
            # one optimizer step per captured graph.  (--steps-per-graph K captures K consecutive steps in one graph: measured
            # SLOWER on the same box, 0.2842 ms against 0.2785 / 0.2798 at K = 5 / 1 -- back-to-back replays of a small graph
            # already overlap the next launch with the running one; profiles/r04_steps_per_graph.txt)
            spg = 1 if args.steps_per_graph < 0 else args.steps_per_graph
            gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=fused_drop, steps_per_graph=spg)
            ms_per_step, loss = timed_graph_steps(gstep, spread)
            spread["steps_per_graph"] = spg
            graph_ok = True
        except Exception as exc:                      # capture refused: keep the eager measurement
            sys.stderr.write(f"bench.py: hipGraph capture failed ({exc!r}); reporting eager launches\n")
        # the dominant kernel's duration INSIDE the replayed graph (the number that belongs next to a hipGraph-replay
        # ms_per_step): a second capture of the same step with external HIP events around that kernel, replayed after --
        # never inside -- the timed windows.  Falls back to the eager-region figure (and says so) where the runtime
        # refuses event-record nodes.
        if graph_ok and rank == 0 and roofline is not None and world == 1:
            try:
                got = replayed_kernel_ms(dominant, lambda: T.TrainStep(model, opt, x, ops, y, w, use_graph=True,
                                                                       fused_dropout=fused_drop, steps_per_graph=1))
                if got is not None:
                    r_ms, r_n = got
                    roofline.update({"eager_avg_ms": roofline["avg_ms"], "avg_ms": round(r_ms, 4), "launches": r_n,
                                     "achieved": round(roofline["algorithmic_bytes"] / (r_ms * 1e-3) / 1e9, 1),
                                     "avg_ms_source": "external HIP events (event-record nodes) around the kernel inside the "
                                                      "replayed hipGraph, on the replay stream"})
                    roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 4)
            except Exception as exc:
                sys.stderr.write(f"bench.py: in-graph kernel timing unavailable ({exc!r}); roofline.avg_ms is the eager figure\n")
        state["done"] = True
        timer_t.cancel()
    if world > 1:                                  # each rank holds the loss over its own rows
        loss = loss.detach().clone()
        dist.all_reduce(loss)
    final_loss = float(loss.item())

    # ---------------- parity of what was just timed: eval-mode logits of the trained model against the CPU oracle
    # on sampled rows (outside the timed regions; rank 0, its own rows) ----------------
    check = None
    if not args.no_check:
        check = sampled_check(args, model, wl_now=(adj, x_np, low), ops=ops, x=x, rows=(b, e), rank=rank, world=world)
        if check is not None and world == 1 and fused_drop and args.optimizer == "fused":
            # ... and of the step as it is timed: one TRAINING-mode step (through the input pipeline where it applies)
            tcheck = training_step_check(args, model, (adj, x_np, y_np, tr, low), ops, x, y, w)
            check["check_train"] = tcheck
            check["checked"] = bool(check["checked"] and tcheck["ok"])
        if check is not None and not check["checked"]:
            sys.stderr.write(f"bench.py: PARITY CHECK FAILED: {check}\n")

    # ---------------- side measurements (single GPU): the same step in the literal project-then-aggregate form, and
    # on the generator's random node ids (no degree relabelling) -- SURVEY.md section 7 asks for both ----------------
    extras = {}
    if world == 1 and use_graph and graph_ok and not args.no_extras:
        try:
            extras = side_measurements(args, timed_graph_steps, model, opt, x, ops, y, w, fused_drop, dev, (tr, va, te))
        except Exception as exc:
            extras = {"extras_error": repr(exc)}

    # which form of the step was timed: with the first layer's input aggregation of step t + 1 inside that layer's backward
    # of step t (train.TrainStep's default where it qualifies; ACM_TUNING=pipeline=0 switches it off) or without
    extras["input_pipeline"] = step.pipe is not None
    if spread:
        extras["ms_per_step_windows"] = spread
    if rank == 0:
        emit(ms_per_step, (f"hipGraph replay of the captured step ({spread.get('steps_per_graph', 1)} consecutive steps per graph)"
                           if graph_ok else "eager launches"), final_loss,
             with_cpu=True, extras=extras, check=check)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if check is not None and rank == 0 and not check["checked"]:
        sys.exit(1)


def replayed_kernel_ms(label, make_step, replays=20):
    """(average ms, launches) of the kernel(s) labelled ``label`` inside the hipGraph that ``make_step()`` captures: the
    capture runs under a functional.KernelTimer(external=True), which brackets those launches with event-record nodes; every
    replay re-records them.  None when nothing under that label was captured."""
    import torch
    from acm_gnn_amd import functional as AF
    probe = AF.KernelTimer(only=label, external=True)
    AF.set_kernel_timer(probe)
    try:
        pstep = make_step()
    finally:
        AF.set_kernel_timer(None)
    if not probe.captured.get(label):
        return None
    for _ in range(3):
        pstep()
    vals = []
    for _ in range(replays):
        pstep()
        vals += probe.captured_ms(label)
    torch.cuda.synchronize()
    return sum(vals) / len(vals), len(vals)


def side_measurements(args, timed_graph_steps, model, opt, x, ops, y, w, fused_drop, dev, splits=None):
    """{literal_ms_per_step, random_order_ms_per_step, ...}: hipGraph replays of the same training step (a) with the
    aggregate-first rewrite switched off (tuning rewrites=0: project, then gather the 2F-wide rows, as the reference's op
    order does), (b) on the same graph with the generator's random node ids instead of the degree relabelling."""
    import torch
    import acm_gnn_amd
    from acm_gnn_amd import data as D, distributed as DD, train as T
    out = {}
    if splits is not None:
        # an EPOCH of the reference's loops = the training step + the evaluation pass (eval-mode forward, accuracy on the
        # three index sets, validation NLL, one host copy: ACM-Geometric/train.py:119-140) -- the unit of the paper's
        # ms/epoch tables; both halves as hipGraph replays
        ev = T.EvalStep(model, x, ops, y, tuple(torch.from_numpy(np.asarray(s_)).to(dev) for s_ in splits), loss_set=1,
                        use_graph=True)
        gstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=fused_drop,
                            pipeline_input=False)       # as train.fit does: an evaluation pass follows every step

        def epoch():
            loss = gstep()
            ev()
            return loss
        ms, _ = timed_graph_steps(epoch)
        out["train_plus_eval_ms_per_epoch"] = round(ms, 4)
        # ... and with the input pipeline left on in the training half (fit(pipeline_input=None))
        pstep = T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=fused_drop)
        if pstep.pipe is not None:
            def epoch_p():
                loss = pstep()
                ev()
                return loss
            ms, _ = timed_graph_steps(epoch_p)
            out["train_plus_eval_pipelined_ms_per_epoch"] = round(ms, 4)
        del pstep
        # the same captured step alone: what train.fit() and the row-sharded runs execute (no input pipeline)
        ms, _ = timed_graph_steps(gstep)
        out["plain_ms_per_step"] = round(ms, 4)
    from acm_gnn_amd import tuning
    if not args.variant and (tuning.HOST.rewrites & tuning.REWRITE_AGG_FIRST):
        with tuning.override(rewrites=tuning.HOST.rewrites & ~tuning.REWRITE_AGG_FIRST):
            ms, _ = timed_graph_steps(T.TrainStep(model, opt, x, ops, y, w, use_graph=True, fused_dropout=fused_drop))
            out["literal_ms_per_step"] = round(ms, 4)
    if not args.variant and x.shape[1] <= 8 and args.hidden == 64:
        # the reference's DEFAULT variant (ACM-Geometric/parse.py:57: --variant 1, ACMII: the ReLU between projection and filter)
        # on the same graph and split: the mask form on the bf16 matrix pipe, and the round-3 path (fp32-MFMA recompute
        # forward + two transposed 64-wide gathers in the backward) it replaces
        torch.manual_seed(args.seed)
        n_cls = int(y.max().item()) + 1
        model_v = acm_gnn_amd.GCN(x.shape[1], args.hidden, n_cls, 2, x.shape[0], args.dropout, args.method, args.structure_info,
                                  variant=True, attn_layernorm=True).to(dev)
        opt_v = acm_gnn_amd.FusedAdamW(model_v.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        ms, _ = timed_graph_steps(T.TrainStep(model_v, opt_v, x, ops, y, w, use_graph=True, fused_dropout=fused_drop))
        out["acmii_ms_per_step"] = round(ms, 4)
        with tuning.override(rewrites=tuning.HOST.rewrites & ~tuning.REWRITE_ACMII_MASK):
            ms, _ = timed_graph_steps(T.TrainStep(model_v, opt_v, x, ops, y, w, use_graph=True, fused_dropout=fused_drop))
            out["acmii_fp32_mfma_ms_per_step"] = round(ms, 4)
        del model_v, opt_v
    other = "random" if args.node_order == "degree" else "degree"
    wl = D.bench_workload(args.dataset, seed=args.seed, node_order=other, uniform=args.uniform,
                          normalize_features=not (args.method in ("acmgcnp", "acmgcnpp") and args.structure_info))
    n = wl["adj"].shape[0]
    x2, y2 = torch.from_numpy(wl["x"]).to(dev), torch.from_numpy(wl["y"]).to(dev)
    w2 = T.row_weights(torch.from_numpy(wl["splits"][0]).to(dev), n, device=dev)
    # the generator's random ids, (a) as they are, (b) with the degree relabelling done INSIDE the operator
    # (graph.relabel_by_degree: what operators_for does for the drop-in route; TrainStep moves x / labels once)
    for key, relabel in ((f"{other}_order_ms_per_step", False), (f"{other}_order_relabelled_in_operator_ms_per_step", True)):
        if relabel and other != "random":
            continue
        ops2 = DD.make_sharded_operators(wl["low"], wl["deg"], dev, with_structure=bool(args.structure_info), relabel=relabel)
        torch.manual_seed(args.seed)
        model2 = acm_gnn_amd.GCN(x2.shape[1], args.hidden, int(wl["y"].max()) + 1, 2, n, args.dropout, args.method,
                                 args.structure_info, variant=bool(args.variant), attn_layernorm=True).to(dev)
        opt2 = acm_gnn_amd.FusedAdamW(model2.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        ms, _ = timed_graph_steps(T.TrainStep(model2, opt2, x2, ops2, y2, w2, use_graph=True, fused_dropout=fused_drop))
        out[key] = round(ms, 4)
    return out


def _oracle_params(model, n_rows=None):
    params = {}
    for k, v in model.named_parameters():
        if k in ("fea_param", "xX_param"):
            continue
        params[k] = v.detach().cpu().clone()
    return params


def _csr_operands(low, adj):
    """(A_low, I - A_low, A) as torch CSR tensors -- the reference's filters (ACM-Geometric/train.py:76-81)."""
    import scipy.sparse as sp
    import torch

    def t(m):
        m = m.tocsr()
        m.sort_indices()
        return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                       torch.from_numpy(m.data.astype(np.float32)), size=m.shape)
    n = low.shape[0]
    return t(low), t(sp.identity(n, dtype=np.float32, format="csr") - low), t(adj.astype(np.float32))


def sampled_check(args, model, wl_now, ops, x, rows, rank, world, n_sample=4096):
    """Eval-mode logits of the model that was just trained and timed, on `n_sample` sampled rows of rank 0's block,
    against oracle.gcn_forward on the host (CSR operands, the same parameters).  Tolerance per element: 1e-6 of the
    logit range (a few rows of this workload carry logits 1e4 times the typical one: row-normalised N(0,1) features,
    train.py:69-73) plus 1e-4 of the element's own magnitude -- fp32 sums over up to 21 k neighbours in different
    orders; measured: max |err| 5e-4 at a logit range of 2.6e4."""
    import torch
    from oracle import acm_oracle as O
    if world > 1 and args.structure_info:
        return {"checked": False, "check_skipped": "struc_low is sharded: rank 0 holds only its rows"} if rank == 0 else None
    adj, x_np, low = wl_now
    was_training = model.training
    model.eval()
    with torch.no_grad():
        got = model(x, ops).float().cpu()
    model.train(was_training)
    if rank != 0:
        return None
    torch.set_num_threads(min(os.cpu_count(), 32))
    a_low, a_high, a_un = _csr_operands(low, adj)
    with torch.no_grad():
        ref = O.gcn_forward(_oracle_params(model), torch.from_numpy(x_np), a_low, a_high,
                            a_un if args.structure_info else None, model_type=args.method, variant=bool(args.variant),
                            structure_info=args.structure_info, attn_layernorm=True, dropout=args.dropout, training=False)
    b, e = rows
    ref = ref[b:e]
    pick = torch.from_numpy(np.random.default_rng(args.seed).choice(e - b, size=min(n_sample, e - b), replace=False))
    d = (got[pick] - ref[pick]).abs()
    tol = 1e-6 * float(ref.abs().max()) + 1e-4 * ref[pick].abs()
    ok = bool((d <= tol).all()) and bool(torch.isfinite(got).all())
    return {"checked": ok, "check": {"rows": int(pick.numel()), "max_abs_err": float(d.max()),
                                     "logit_range": float(ref.abs().max()),
                                     "max_err_over_tol": float((d / tol).max()), "against": "oracle.gcn_forward (CPU, CSR operands), eval mode"}}


def training_step_check(args, model, wl_now, ops, x, y, w):
    """One TRAINING-mode step of the model that was just timed, as it was timed (counter-based dropout inside the kernels;
    through the input pipeline when the configuration qualifies, so that the first layer's P is the one the previous
    step's backward kernel carried), against the oracle's step on the host with the masks regenerated in numpy
    (oracle/philox.py): the loss and EVERY parameter gradient.  lr = 0, so the parameters stay what the timed run left.
    Tolerances as in tests/test_gpu_fullsize.py: loss 3e-5 relative; a gradient within 3e-4 of its range (+1e-6), or --
    the first layer's weight gradients sum products with feature values up to 1e4 -- within 5x the fp32 oracle's own
    distance from a float64 run of the oracle."""
    import torch
    import acm_gnn_amd
    from acm_gnn_amd import train as T
    from oracle import acm_oracle as O
    from oracle.philox import dropout_factors
    adj, x_np, y_np, tr, low = wl_now
    n = x_np.shape[0]
    opt0 = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    chk = T.TrainStep(model, opt0, x, ops, y, w, use_graph=False, fused_dropout=True)
    st = model.dropout_state
    chk()                                            # a whole step: its backward carries the next step's P (pipeline)
    opt0.zero_grad(set_to_none=True)
    counter = int(st.step.item())
    loss = chk._forward_backward()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    masks = {"x": torch.from_numpy(dropout_factors(st.seed, counter, 0, args.dropout, n, x_np.shape[1]) > 0).float(),
             "hidden": torch.from_numpy(dropout_factors(st.seed, counter, 1, args.dropout, n, args.hidden) > 0).float()}
    torch.set_num_threads(min(os.cpu_count(), 32))
    operands = _csr_operands(low, adj)
    xt, yt, idx = torch.from_numpy(x_np), torch.from_numpy(y_np), torch.from_numpy(tr)

    def oracle_step(dtype):
        ops_t = operands
        if dtype != torch.float32:
            ops_t = tuple(torch.sparse_csr_tensor(t.crow_indices(), t.col_indices(), t.values().to(dtype), size=t.shape)
                          for t in operands)
        ps = {k: v.to(dtype).requires_grad_(True) for k, v in _oracle_params(model).items()}
        ref = O.gcn_forward(ps, xt.to(dtype), ops_t[0], ops_t[1], ops_t[2] if args.structure_info else None,
                            model_type=args.method, variant=bool(args.variant), structure_info=args.structure_info,
                            attn_layernorm=True, dropout=args.dropout, training=True,
                            masks={k: v.to(dtype) for k, v in masks.items()})
        ls = O.nll_loss_on(ref, yt, idx)
        ls.backward()
        return float(ls), {k: v.grad for k, v in ps.items() if v.grad is not None}

    ref_loss, ref_g = oracle_step(torch.float32)
    ok = abs(float(loss) - ref_loss) <= 3e-5 * max(1.0, abs(ref_loss))
    worst, needs64 = 0.0, []
    for k, rg in ref_g.items():
        if k not in got:
            ok = False
            continue
        d = float((got[k].double() - rg.double()).abs().max())
        tol = 3e-4 * float(rg.abs().max()) + 1e-6
        worst = max(worst, d / max(float(rg.abs().max()), 1e-30))
        if d >= tol:
            needs64.append((k, d, tol))
    used64 = {}
    if needs64:
        _, g64 = oracle_step(torch.float64)
        for k, d, tol in needs64:
            d64 = float((got[k].double() - g64[k]).abs().max())
            e_ref = float((ref_g[k].double() - g64[k]).abs().max())
            used64[k] = [d64, e_ref]
            ok = ok and d64 <= max(5.0 * e_ref, tol)
    return {"ok": bool(ok), "input_pipeline": chk.pipe is not None, "dropout_counter": counter, "loss": float(loss),
            "loss_oracle": ref_loss, "gradients": len(ref_g), "worst_grad_err_over_range_vs_fp32_oracle": worst,
            "vs_fp64_oracle_[gpu_err, fp32_oracle_err]": used64,
            "against": "oracle.gcn_forward + NLL + backward (CPU, CSR operands), training mode, masks regenerated (oracle/philox.py)"}


def cpu_baseline(args, model, wl):
    """The oracle's literal torch-CPU restatement of the reference step (sparse COO operands, same op order as
    ACM-Geometric/layers.py:78-116) timed on the host cores, ON THE BENCHMARK GRAPH ITSELF: one warm-up step, then the
    median of `reps` full train steps (forward + loss + backward + AdamW); edges/s = nnz(A_low) / t.  `csr_value`
    is the same math with CSR operands ("best effort" CPU).  Both legs use min(host cores, 32)
    threads: torch's sparse kernels get slower beyond a few dozen threads."""
    import torch
    from oracle import acm_oracle as O
    adj, x_np, y_np, (tr, _, _), low = wl["adj"], wl["x"], wl["y"], wl["splits"], wl["low"]
    nnz = int(low.nnz)
    x, y, idx = torch.from_numpy(x_np), torch.from_numpy(y_np), torch.from_numpy(tr)
    kw = dict(model_type=args.method, variant=bool(args.variant), structure_info=args.structure_info,
              attn_layernorm=True, dropout=args.dropout, training=True)
    cores = min(os.cpu_count(), 32)
    torch.set_num_threads(cores)
    reps = 3

    def run(a_low, a_high, a_un):
        params = {k: v.requires_grad_(True) for k, v in _oracle_params(model).items()}
        opt = torch.optim.AdamW(list(params.values()), lr=args.lr, weight_decay=args.weight_decay)
        times = []
        for _ in range(reps + 1):                     # the first step is the warm-up
            t = time.perf_counter()
            opt.zero_grad()
            out = O.gcn_forward(params, x, a_low, a_high, a_un if args.structure_info else None, **kw)
            loss = O.nll_loss_on(out, y, idx)
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t)
        return float(np.median(times[1:]))

    t_csr = run(*_csr_operands(low, adj))
    coo = O.filters_linkx(adj)                         # the reference's operand format: un-coalesced sparse COO
    t_coo = run(*coo)
    return {"value": round(nnz / t_coo, 1), "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"median of {reps} full train steps (fwd+loss+bwd+AdamW) after 1 warm-up, same model, on the "
                      f"benchmark graph itself ({x.shape[0]} nodes, nnz(A_low)={nnz}); operands in the reference's "
                      f"format (un-coalesced sparse COO), torch CPU, {cores} threads of {os.cpu_count()} host cores",
            "ms_per_step": round(t_coo * 1e3, 1),
            "csr_value": round(nnz / t_csr, 1), "csr_ms_per_step": round(t_csr * 1e3, 1),
            "host_cores": os.cpu_count()}


if __name__ == "__main__":
    main()
