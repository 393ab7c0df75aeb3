"""End-to-end accuracy parity: replay on the MI355X the exact experiments recorded from the
imported reference (tests/golden/make_accuracy_golden.py) -- same fixed splits, same seeded CPU
initialisation, same dropout masks (tests/replay.py), same optimizer and model-selection rule --
and compare the selected test accuracy per split.  Target: mean within
+-0.2 pp of the reference run; single splits may differ by a few test nodes."""
import os
import threading

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from conftest import GOLDEN, csr_to_coo_tensor, load_npz
from replay import SeededDropout

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(name):
    if name == "cora":
        g = load_npz(os.path.join(GOLDEN, "graph_cora.npz"))
        n = int(g["n"])
        x = sp.csr_matrix((g["feat_vals"], g["feat_indices"], g["feat_indptr"]), shape=(n, int(g["feat_dim"]))).toarray()
        masks = [(g["train_mask"], g["val_mask"], g["test_mask"])]
        return n, torch.from_numpy(x.astype(np.float32)), torch.from_numpy(g["labels"]), g, masks
    g = load_npz(os.path.join(GOLDEN, f"graph_{name}.npz"))          # squirrel, film: binary bag-of-words features
    n = int(g["n"])
    x = sp.csr_matrix((np.ones(len(g["feat_indices"]), np.float32), g["feat_indices"], g["feat_indptr"]),
                      shape=(n, int(g["feat_dim"]))).toarray()
    masks = [tuple(np.unpackbits(g[f"{k}_mask_{i}"])[:n].astype(bool) for k in ("train", "val", "test"))
             for i in range(10)]
    return n, torch.from_numpy(x), torch.from_numpy(g["labels"]), g, masks


def _cora_masks(split):
    # Cora's ten fixed splits are not all shipped as fixtures; split 0 is in graph_cora.npz, the
    # others are re-derived from the recorded reference run only if present.
    return None


# name -> (mean |selected acc - reference run| bound, per-split bound); see the criterion below.  The mean bound is
# max(0.2 pp, the distance between two runs of the REFERENCE ITSELF that differ only in fp32 summation order) when the
# second reference run is recorded (accuracy_<name>_b.npz, make_accuracy_golden.py --b: Squirrel 66.11 vs 65.81 %, 0.30 pp),
# the listed value otherwise.
REPLAYS = {"cora": (0.002, 0.01), "squirrel": (0.002, 0.025), "film_v0": (0.004, 0.025), "film_v1": (0.004, 0.025)}
# recorded with the library's counter-based masks injected into the reference (--philox): replayed on the fused small-graph step
# (mean selected bound, per-split selected bound[, per-split bound of the second-half curve gap: 0.4 pp unless the reference's
#  own second run of the experiment -- accuracy_<name>_b_philox.npz, summation order only -- shows a wider band])
PHILOX_REPLAYS = {"cora": (0.002, 0.01), "squirrel": (0.002, 0.025), "chameleon_syn": (0.002, 0.025)}


def _reference_band(name, splits):
    """|mean selected accuracy of reference run A - of reference run B| over `splits` (and the per-split differences), or
    None when run B was not recorded."""
    path = os.path.join(GOLDEN, f"accuracy_{name}_b.npz")
    if not os.path.exists(path):
        return None
    a, b = load_npz(os.path.join(GOLDEN, f"accuracy_{name}.npz")), load_npz(path)
    ia = {s: i for i, s in enumerate(a["cfg"]["splits"])}
    ib = {s: i for i, s in enumerate(b["cfg"]["splits"])}
    common = [s for s in splits if s in ia and s in ib]
    if not common:
        return None
    da = np.asarray([a["test_acc"][ia[s]] for s in common], dtype=np.float64)
    db = np.asarray([b["test_acc"][ib[s]] for s in common], dtype=np.float64)
    return abs(float(da.mean() - db.mean())), (db - da)


def _prepare(name, suffix=""):
    """Everything a replay needs, on the host: config, recorded run, features / labels, the small-graph dialect's
    filters (ACM-Pytorch/utils.py:612-629, built with the torch ops the reference uses), the fixed splits."""
    rec = load_npz(os.path.join(GOLDEN, f"accuracy_{name}{suffix}.npz"))
    cfg = rec["cfg"]
    dataset = cfg.get("dataset", name)
    n, x, labels, g, masks = _load(dataset)
    splits_path = os.path.join(GOLDEN, f"splits_{dataset}.npz")
    if len(masks) >= len(cfg["splits"]):             # the graph fixture carries every fixed split (squirrel, film)
        masks = dict(enumerate(masks))
    elif os.path.exists(splits_path):
        sp_rec = load_npz(splits_path)
        masks = {int(k.split("_")[1]): None for k in sp_rec if k.startswith("train_")}
        masks = {i: tuple(np.unpackbits(sp_rec[f"{k}_{i}"])[:n].astype(bool) for k in ("train", "val", "test"))
                 for i in masks}
    else:
        masks = dict(enumerate(masks))
    a_un = csr_to_coo_tensor(g, "adj_un")
    if not (cfg["model"] in ("acmgcnp", "acmgcnpp") and cfg["structure_info"]):
        rs = x.sum(1)
        inv = torch.pow(rs, -1)
        inv[torch.isinf(inv)] = 0.0
        x = torch.mm(torch.diag(inv), x)
    rowsum = (torch.eye(n) + a_un.to_dense()).sum(1)
    inv = torch.pow(rowsum, -1)
    inv[torch.isinf(inv)] = 0.0
    adj_low = torch.mm(torch.diag(inv), torch.eye(n) + a_un.to_dense())
    adj_high = (torch.eye(n) - adj_low).to_sparse()
    return rec, cfg, dataset, n, x, labels, masks, a_un, adj_low, adj_high


_REPLAYS_DONE = {}        # (dataset, gather dtype) -> {split: result}: the fp32 replays serve both tests below (gate time)
_REPLAY_LOCKS = {}        # (dataset, gather dtype) -> lock: a test waits for the prefetch thread's batch instead of repeating it
_REPLAY_LOCKS_GUARD = threading.Lock()
BF16_REPLAYS = ["film_v1", "squirrel"]


def _fixed_split_todo(name):
    """The splits test_fixed_split_accuracy_matches_reference_run replays (None: fixture not generated)."""
    path = os.path.join(GOLDEN, f"accuracy_{name}.npz")
    if not os.path.exists(path):
        return None
    cfg = load_npz(path)["cfg"]
    _, _, _, _, _, _, masks, *_ = _prepare(name)
    return [s for s in cfg["splits"] if s in masks]


def prefetch_replays():
    """Gate time: the replays are bound by the CPU generation of the recorded masks and leave the GPU mostly idle, so
    conftest.py starts them in a background thread when the session begins and moves this module's tests to the END of
    the run -- the worker processes then train underneath the other GPU tests.  Same batches, same worker count, same
    results (every worker is its own process with its own HIP context); a batch that fails here is simply run again, and
    reported, by the test that needs it."""
    def work():
        plan = [(name, "fp32") for name in REPLAYS] + [(name, "bf16") for name in BF16_REPLAYS]
        for name, dt in plan:
            try:
                if dt == "fp32":
                    todo = _fixed_split_todo(name)
                else:
                    path = os.path.join(GOLDEN, f"accuracy_{name}.npz")
                    todo = list(load_npz(path)["cfg"]["splits"])[:5] if os.path.exists(path) else None
                if todo:
                    _replay_parallel(name, todo, dt)
            except Exception:                      # noqa: BLE001 -- the test repeats the batch and shows the error
                pass
    t = threading.Thread(target=work, name="accuracy-replay-prefetch", daemon=True)
    t.start()
    return t


def _replay_parallel(name, todo, gather_dtype="fp32"):
    """The splits of one recorded experiment side by side in worker processes (spawned: each gets its own HIP context on the
    same GPU); results are kept for the other tests of this module."""
    import concurrent.futures as cf
    import multiprocessing as mp
    with _REPLAY_LOCKS_GUARD:
        lock = _REPLAY_LOCKS.setdefault((name, gather_dtype), threading.Lock())
    with lock:
        have = _REPLAYS_DONE.setdefault((name, gather_dtype), {})
        missing = [s for s in todo if s not in have]
        if missing:
            workers = min(5, len(missing))        # (ten processes on the one GPU were measured slower: 48 s against 27 s for Film)
            chunks = [missing[i::workers] for i in range(workers)]
            with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
                for part in pool.map(_replay_splits, [name] * workers, chunks, [gather_dtype] * workers):
                    have.update(part)
        return {s: have[s] for s in todo}


def _replay_splits(name, splits, gather_dtype="fp32"):
    """Worker (its own process: the replay is bound by the CPU generation of the recorded dropout masks, 80 ms per
    epoch on Squirrel, so the splits of one dataset run side by side on the one GPU): train the given splits exactly as
    the recorded reference run did and return {split: (selected test acc, val-loss curve, test-acc curve)}."""
    from acm_gnn_amd import GCN, layers, train as T
    from acm_gnn_amd.graph import clear_cache
    rec, cfg, dataset, n, x, labels, masks, a_un, adj_low, adj_high = _prepare(name)
    torch.set_num_threads(4)
    xd, yd = x.to(DEV), labels.to(DEV)
    low_d, high_d = adj_low.to(DEV), adj_high.to(DEV)
    un_d = a_un.to(DEV) if cfg["structure_info"] else None
    layers._default_device = lambda: torch.device("cpu")            # seeded CPU initialisation, like the reference run
    real_dropout = F.dropout
    out = {}
    try:
        for split in splits:
            tr, va, te = (torch.from_numpy(np.nonzero(m)[0]).to(DEV) for m in masks[split])
            clear_cache()
            torch.manual_seed(1000 + split)
            model = GCN(x.shape[1], cfg["hidden"], int(labels.max()) + 1, 1, n, cfg["dropout"], cfg["model"],
                        cfg["structure_info"], variant=bool(cfg["variant"]), attn_layernorm=False,
                        gather_dtype=gather_dtype).to(DEV)            # storage type of the gathered operands
            opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
            drop = SeededDropout(seed=split)
            F.dropout = drop
            w = T.row_weights(tr, n)
            step = T.TrainStep(model, opt, xd, low_d, yd, w, high_d, un_d)
            best_val, curr, vals, accs = float("inf"), 0.0, [], []
            for epoch in range(cfg["epochs"]):
                drop.next_epoch()
                step()
                o, (acc_te,) = T.evaluate(model, xd, low_d, yd, (te,), high_d, un_d)
                val_loss = float(F.nll_loss(F.log_softmax(o, 1)[va], yd[va]))
                vals.append(val_loss)
                accs.append(acc_te)
                if val_loss < best_val:
                    best_val, curr = val_loss, acc_te
                if cfg["early_stopping"] > 0 and epoch > cfg["early_stopping"]:
                    if val_loss > np.mean(vals[epoch - cfg["early_stopping"]:epoch]):
                        break
            out[split] = (curr, vals, accs)
    finally:
        F.dropout = real_dropout
    return out


@pytest.mark.parametrize("name", list(REPLAYS))
def test_fixed_split_accuracy_matches_reference_run(name):
    path = os.path.join(GOLDEN, f"accuracy_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    rec = load_npz(path)
    cfg = rec["cfg"]
    _, _, _, _, _, _, masks, *_ = _prepare(name)
    todo = [s for s in cfg["splits"] if s in masks]
    results = _replay_parallel(name, todo)
    _judge_replay(name, rec, results, f"accuracy_replay_{name}.json")


def _judge_replay(name, rec, results, out_name, path_label="general path", bounds=None):
    """The parity criterion of a replayed reference run (``results``: {split: (selected test acc, val-loss curve, test-acc
    curve)}) against the recorded one (``rec``); writes the evidence file gpurun_out/<out_name>."""
    cfg = rec["cfg"]
    dataset = cfg.get("dataset", name)
    got, ref, curve_gap, curves, at_ref_epoch = [], [], [], [], []
    for si, split in enumerate(cfg["splits"]):
        if split not in results:
            continue
        curr, vals, accs = results[split]
        got.append(curr)
        ref.append(float(rec["test_acc"][si]))
        hist = rec[f"hist_{split}"]
        m = min(len(vals), len(hist))
        curves.append({"split": int(split), "val_loss": [round(v, 6) for v in vals], "test_acc": [round(a, 5) for a in accs]})
        k_ref = int(np.argmin(hist[:, 1]))                 # the epoch the reference run selected
        at_ref_epoch.append(accs[min(k_ref, len(accs) - 1)] - float(hist[k_ref, 2]))
        # same init + same masks: the validation-loss curve tracks the reference's (tightly at first, then within
        # fp32 chaos) and so does the per-epoch test accuracy
        np.testing.assert_allclose(vals[:5], hist[:5, 1], rtol=2e-4)
        # Film trains without dropout at lr 0.05: the loss curve has isolated spikes whose height is chaotic
        # (and so does the synthetic-label Chameleon run: lr 0.05 under dropout 0.7 -- its curve follows the reference's to six
        # digits for ten epochs, then to 13 % while the loss wanders around its minimum)
        np.testing.assert_allclose(vals[:m], hist[:m, 1], rtol=0.12 if dataset == "film" else (0.2 if cfg["lr"] >= 0.05 else 3e-2))
        if cfg["lr"] >= 0.05:
            np.testing.assert_allclose(vals[:10], hist[:10, 1], rtol=1e-3)
        curve_gap.append(float(np.mean(accs[m // 2:m]) - np.mean(hist[m // 2:m, 2])))
    got, ref = np.asarray(got), np.asarray(ref)
    print(f"\n{name} ({path_label}): reference-run {100 * ref.mean():.2f} +- {100 * ref.std():.2f}  |  MI355X {100 * got.mean():.2f} "
          f"+- {100 * got.std():.2f}  | selected, per split {np.round(100 * (got - ref), 2).tolist()}"
          f"  | mean test-acc over the 2nd half of training, per split {np.round(100 * np.asarray(curve_gap), 2).tolist()} pp")
    out_dir = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, out_name), "w") as fh:
        import json
        band0 = _reference_band(name, [s for s in cfg["splits"] if s in results])
        json.dump({"config": cfg, "path": path_label, "reference_run": ref.tolist(), "mi355x": got.tolist(),
                   "reference_run_b_mean_distance_pp": None if band0 is None else 100 * band0[0],
                   "reference_run_b_minus_a_pp": None if band0 is None else (100 * band0[1]).tolist(),
                   "mean_diff_pp": float(100 * (got.mean() - ref.mean())), "curve_gap_pp": (100 * np.asarray(curve_gap)).tolist(),
                   "at_reference_epoch_diff_pp": (100 * np.asarray(at_ref_epoch)).tolist(),
                   "curves": curves}, fh)
    print(f"   test accuracy at the epoch the reference selected, per split: {np.round(100 * np.asarray(at_ref_epoch), 2).tolist()} pp "
          f"(mean {100 * np.mean(at_ref_epoch):+.2f})")
    # Parity criterion (+-0.2 pp), in three strengths:
    #  (a) the test-accuracy CURVES, averaged over the second half of training, agree to 0.2 pp on the mean of the splits
    #      (0.4 pp on every single split);
    #  (b) the test accuracy AT THE EPOCH THE REFERENCE RUN SELECTED agrees to 0.2 pp on the mean of the splits -- the
    #      reference's reported number with the model-selection jitter taken out;
    #  (c) the accuracy this run SELECTS ITSELF (test acc at the arg-min of its own validation loss).  The validation
    #      loss is flat to 0.5 % over 10-30 epochs around its minimum while the curves of the two runs differ by 0.2 %
    #      there, so the arg-min lands on a different epoch of that band and the functional moves by up to 2 pp per split
    #      in either direction (recorded Squirrel replay: per-epoch accuracies equal to < 0.1 pp, selected -0.53 pp on
    #      the mean of ten splits; the reference's own split-to-split std is 1.7 pp) -- bounded per split and on the
    #      mean (REPLAYS).
    mean_bound, split_bound = bounds if bounds is not None else REPLAYS[name]
    band = _reference_band(name, [s for s in cfg["splits"] if s in results])
    if band is not None:                               # the reference's own run-to-run distance, measured
        print(f"   two runs of the reference itself (summation order only): mean {100 * band[0]:.2f} pp apart, per split "
              f"{np.round(100 * band[1], 2).tolist()}")
        mean_bound = max(0.002, band[0])
    gap_bound = 0.004
    band_path = os.path.join(GOLDEN, f"accuracy_{name}_b_philox.npz")
    if bounds is not None and os.path.exists(band_path):
        # the REFERENCE's own second run of this experiment (same masks, init, splits; CSR operands + 3 threads): how far its
        # second-half test-accuracy curves sit from the first run's, per split -- no replay can be asked to do better
        rb = load_npz(band_path)
        own = []
        for split in cfg["splits"]:
            if f"hist_{split}" in rb and f"hist_{split}" in rec:
                ha, hb = rec[f"hist_{split}"], rb[f"hist_{split}"]
                mm = min(len(ha), len(hb))
                own.append(abs(float(np.mean(hb[mm // 2:mm, 2]) - np.mean(ha[mm // 2:mm, 2]))))
        if own:
            sel_band = abs(float(np.mean(rb["test_acc"]) - np.mean(rec["test_acc"])))
            print(f"   the reference's own two runs, second-half curve gap per split: {np.round(100 * np.asarray(own), 2).tolist()} pp; "
                  f"selected accuracy, means {100 * sel_band:.2f} pp apart")
            gap_bound = max(gap_bound, 2.0 * max(own))          # (a sanity bound per split; the MEAN over the splits is the criterion)
            mean_bound = max(mean_bound, sel_band)
    assert abs(np.mean(curve_gap)) <= 0.002 and np.all(np.abs(curve_gap) <= gap_bound), (curve_gap, gap_bound)
    assert abs(np.mean(at_ref_epoch)) <= 0.002, at_ref_epoch
    assert np.all(np.abs(got - ref) <= split_bound), (got - ref)
    assert abs(got.mean() - ref.mean()) <= mean_bound, (got.mean(), ref.mean())


def _replay_small_step(name, splits, use_graph=True):
    """The recorded Philox-mask reference run (accuracy_<name>_philox.npz) replayed on the path that RUNS these graphs by
    default: TrainStep handed the reference's adjacency TENSORS and dense features builds the operators and the CSR twin,
    takes the fused small-graph step (acm_small_step: six launches, masks drawn inside the kernels from the same
    (seed, step) the recording injected into the reference, FusedAdam's update applied where the gradients finish) and the
    three-launch evaluation pass (EvalStep) -- ACM-Pytorch/train.py:95-139.  In-process: a captured step + evaluation takes
    ~0.3 ms, there is no CPU mask generation to wait for."""
    from acm_gnn_amd import GCN, FusedAdam, functional as AF, layers, train as T
    from acm_gnn_amd.graph import clear_cache
    rec = load_npz(os.path.join(GOLDEN, f"accuracy_{name}_philox.npz"))
    cfg = rec["cfg"]
    _, _, dataset, n, x, labels, masks, a_un, adj_low, adj_high = _prepare(name, "_philox")
    assert F.dropout is T._TORCH_DROPOUT               # nobody's mask-replay patch is active in this process
    xd, yd = x.to(DEV), labels.to(DEV)
    low_d, high_d = adj_low.to(DEV), adj_high.to(DEV)
    un_d = a_un.to(DEV) if cfg["structure_info"] else None
    default_device = layers._default_device
    layers._default_device = lambda: torch.device("cpu")            # seeded CPU initialisation, like the reference run
    out = {}
    try:
        for split in splits:
            tr, va, te = (torch.from_numpy(np.nonzero(m)[0]).to(DEV) for m in masks[split])
            clear_cache()
            torch.manual_seed(1000 + split)
            model = GCN(x.shape[1], cfg["hidden"], int(labels.max()) + 1, 1, n, cfg["dropout"], cfg["model"],
                        cfg["structure_info"], variant=bool(cfg["variant"]), attn_layernorm=False).to(DEV)
            model.fused_dropout = True
            model.dropout_state = AF.DropoutState(torch.device(DEV), seed=int(cfg["philox_seed"]) + split)
            opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
            step = T.TrainStep(model, opt, xd, low_d, yd, T.row_weights(tr, n), high_d, un_d, use_graph=use_graph)
            assert step.small is not None, step.small_refused
            ev = T.EvalStep(model, xd, low_d, yd, (va, te), high_d, un_d, loss_set=0, use_graph=use_graph)
            assert ev.small is not None, ev.small_refused
            best_val, curr, vals, accs = float("inf"), 0.0, [], []
            for epoch in range(cfg["epochs"]):
                step()
                _, (_, acc_te), val_loss = ev()
                vals.append(val_loss)
                accs.append(acc_te)
                if val_loss < best_val:
                    best_val, curr = val_loss, acc_te
                if cfg["early_stopping"] > 0 and epoch > cfg["early_stopping"]:
                    if val_loss > np.mean(vals[epoch - cfg["early_stopping"]:epoch]):
                        break
            assert int(model.dropout_state.step.item()) == len(vals)
            out[split] = (curr, vals, accs)
    finally:
        layers._default_device = default_device
    return rec, out


@pytest.mark.parametrize("name", list(PHILOX_REPLAYS))
def test_small_step_accuracy_matches_reference_run_with_its_own_masks(name):
    """VERDICT r05 item 1 / SURVEY 8 row g on the DEFAULT path of BASELINE configs 1-3: the reference itself was trained with
    the library's counter-based masks injected (make_accuracy_golden.py --philox: PhiloxDropout over oracle/philox.py), so
    the fused small-graph step -- which cannot be fed mask tensors -- replays the very same experiment: ten fixed splits,
    seeded init, torch.optim.Adam semantics (FusedAdam), the min-val-loss selection rule; selected test accuracy within
    +-0.2 pp on the mean of the splits (same three-strength criterion as the general path's replay above)."""
    path = os.path.join(GOLDEN, f"accuracy_{name}_philox.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    cfg = load_npz(path)["cfg"]
    _, _, _, _, _, _, masks, *_ = _prepare(name, "_philox")
    rec, results = _replay_small_step(name, [s for s in cfg["splits"] if s in masks])
    _judge_replay(name, rec, results, f"accuracy_replay_small_{name}.json", path_label="fused small-graph step", bounds=PHILOX_REPLAYS[name])
    if name == "cora":                                  # the eager step is the same six launches: same run, bit for bit
        _, eager = _replay_small_step(name, cfg["splits"][:1], use_graph=False)
        s0 = cfg["splits"][0]
        assert eager[s0][1] == results[s0][1] and eager[s0][2] == results[s0][2]


@pytest.mark.parametrize("name", BF16_REPLAYS)
def test_bf16_gathered_operands_keep_the_accuracy(name):
    """the opt-in bf16 storage of the gathered operands (gather_dtype = "bf16": the wide forward
    gathers, the structure-channel gathers, and -- new -- the transposed products of the backward, acm_conv_bwd_spmm_t.
    gather_bf16 / spmm_sub) must not cost accuracy.  The recorded reference experiments (Film ACMII-GCN+, Squirrel ACM-GCN+
    with A: same init, masks, optimizer, selection rule) replayed twice on the MI355X, fp32 and bf16: the test-accuracy
    curves over the second half of training agree to 0.2 pp on the mean of the splits, and the selected accuracy to
    max(0.2 pp, the reference's own run-to-run band) -- the selection functional moves by that much under ANY change of
    fp32 summation order."""
    path = os.path.join(GOLDEN, f"accuracy_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    rec = load_npz(path)
    todo = list(rec["cfg"]["splits"])[:5]              # five splits (one batch of workers); the fp32 replays of
    runs = {dt: _replay_parallel(name, todo, dt) for dt in ("fp32", "bf16")}    # test_fixed_split_... are reused
    sel = {dt: np.asarray([runs[dt][s][0] for s in todo]) for dt in runs}
    half = []
    for s in todo:
        a32, a16 = np.asarray(runs["fp32"][s][2]), np.asarray(runs["bf16"][s][2])
        m = min(len(a32), len(a16))
        half.append(float(a16[m // 2:m].mean() - a32[m // 2:m].mean()))
    # anatomy of the selection: the selected accuracy is the test accuracy at the arg-min of the
    # validation loss.  Storing the gathered operands in bf16 perturbs the validation-loss curve by delta = max_e |v16 - v32|;
    # the arg-min of the perturbed curve then lies anywhere on the fp32 curve's 2-delta plateau (v32[e16] <= v16[e16] + delta
    # <= v16[e32] + delta <= v32[e32] + 2 delta), and the selected accuracy anywhere in the range of test accuracies over
    # that plateau (+ the difference of the two test-accuracy curves there).
    anatomy = []
    for s in todo:
        v32, a32 = np.asarray(runs["fp32"][s][1]), np.asarray(runs["fp32"][s][2])
        v16, a16 = np.asarray(runs["bf16"][s][1]), np.asarray(runs["bf16"][s][2])
        m = min(len(v32), len(v16))
        e32, e16 = int(np.argmin(v32)), int(np.argmin(v16))
        delta = float(np.abs(v16[:m] - v32[:m]).max())
        plateau = np.nonzero(v32[:m] <= v32.min() + 2 * delta)[0]
        anatomy.append({"split": int(s), "epochs_fp32": len(v32), "epochs_bf16": len(v16), "argmin_fp32": e32, "argmin_bf16": e16,
                        "val_min_fp32": float(v32.min()), "val_min_bf16": float(v16.min()), "max_val_curve_diff": delta,
                        "plateau_epochs": int(len(plateau)), "plateau_first_last": [int(plateau.min()), int(plateau.max())],
                        "fp32_test_acc_over_plateau_pp": [float(100 * a32[plateau].min()), float(100 * a32[plateau].max())],
                        "test_acc_pp": {"fp32@fp32": float(100 * a32[e32]), "bf16@fp32": float(100 * a16[min(e32, len(a16) - 1)]),
                                        "fp32@bf16": float(100 * a32[min(e16, len(a32) - 1)]), "bf16@bf16": float(100 * a16[e16])},
                        "max_test_curve_diff_on_plateau_pp": float(100 * np.abs(a16[plateau] - a32[plateau]).max())})
    band = _reference_band(name, todo)
    bound = max(0.002, band[0]) if band is not None else 0.004
    print(f"\n{name}: fp32 {100 * sel['fp32'].mean():.2f} % | bf16 {100 * sel['bf16'].mean():.2f} % | selected, per split "
          f"{np.round(100 * (sel['bf16'] - sel['fp32']), 2).tolist()} | 2nd-half curve mean, per split {np.round(100 * np.asarray(half), 2).tolist()} pp")
    out_dir = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"accuracy_bf16_{name}.json"), "w") as fh:
        import json
        json.dump({"fp32_selected": sel["fp32"].tolist(), "bf16_selected": sel["bf16"].tolist(),
                   "selected_mean_diff_pp": float(100 * (sel["bf16"].mean() - sel["fp32"].mean())),
                   "curve_second_half_diff_pp": (100 * np.asarray(half)).tolist(), "selected_bound_pp": 100 * bound,
                   "at_fp32_epoch_diff_pp": [a["test_acc_pp"]["bf16@fp32"] - a["test_acc_pp"]["fp32@fp32"] for a in anatomy],
                   "selection_anatomy": anatomy}, fh, indent=1)
    # the curves agree; the SELECTED accuracy must not be lower by more than the bound (measured: Film ACMII-GCN+ selects
    # +0.44 pp HIGHER under bf16 -- per split -0.13 ... +1.18 pp, the arg-min of a flat validation loss landing elsewhere --
    # with the curves 0.06 pp apart; Squirrel see the printed line); a two-sided sanity bound of 0.8 pp on top
    assert abs(np.mean(half)) <= 0.002 and np.all(np.abs(half) <= 0.006), half
    # two-sided, at the SAME epoch (the one the fp32 run selected): the two trajectories agree to 0.3 pp on the mean of the
    # splits (a single epoch's test accuracy is noisier than the half-run mean above)
    same_epoch = np.asarray([a["test_acc_pp"]["bf16@fp32"] - a["test_acc_pp"]["fp32@fp32"] for a in anatomy])
    assert abs(same_epoch.mean()) <= 0.3, same_epoch
    assert sel["bf16"].mean() >= sel["fp32"].mean() - bound, (sel["bf16"].mean(), sel["fp32"].mean(), bound)
    assert abs(sel["bf16"].mean() - sel["fp32"].mean()) <= 0.008, (sel["bf16"].mean(), sel["fp32"].mean())


@pytest.mark.parametrize("name,n_splits", [("cora", 5), ("squirrel", 3)])
def test_fused_training_stack_reaches_the_reference_accuracy_band(name, n_splits):
    """The whole fused step -- counter-based dropout inside the kernels, acm_adam_step, the captured graph -- cannot
    replay the reference's CPU dropout masks, so it is checked statistically: trained on the splits of the recorded
    reference run with the reference's hyper-parameters and selection rule, its mean test accuracy must land in the
    reference's band (Cora ACM-GCN 87.59 +- 1.13 % over ten splits, Squirrel ACM-GCN+ with A 66.79 +- 0.86 % over three;
    +-1.5 pp on the mean of the splits used here)."""
    path = os.path.join(GOLDEN, f"accuracy_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"accuracy_{name}.npz not generated")
    rec = load_npz(path)
    cfg = rec["cfg"]
    from acm_gnn_amd import GCN, FusedAdam, train as T
    from acm_gnn_amd.graph import clear_cache
    n, x, labels, g, _ = _load(name)
    sp_rec = load_npz(os.path.join(GOLDEN, f"splits_{name}.npz"))
    a_un = csr_to_coo_tensor(g, "adj_un")
    if not (cfg["model"] in ("acmgcnp", "acmgcnpp") and cfg["structure_info"]):
        rs = x.sum(1)
        inv = torch.pow(rs, -1)
        inv[torch.isinf(inv)] = 0.0
        x = torch.mm(torch.diag(inv), x)
    rowsum = (torch.eye(n) + a_un.to_dense()).sum(1)
    adj_low = torch.mm(torch.diag(torch.pow(rowsum, -1)), torch.eye(n) + a_un.to_dense())
    adj_high = (torch.eye(n) - adj_low).to_sparse()
    xd, yd, low_d, high_d = x.to(DEV), labels.to(DEV), adj_low.to(DEV), adj_high.to(DEV)
    un_d = a_un.to(DEV) if cfg["structure_info"] else None
    got, ref = [], []
    for si, split in enumerate(cfg["splits"][:n_splits]):
        tr, va, te = (torch.from_numpy(np.nonzero(np.unpackbits(sp_rec[f"{k}_{split}"])[:n].astype(bool))[0]).to(DEV)
                      for k in ("train", "val", "test"))
        clear_cache()
        torch.manual_seed(1000 + split)
        model = GCN(x.shape[1], cfg["hidden"], int(labels.max()) + 1, 1, n, cfg["dropout"], cfg["model"],
                    cfg["structure_info"], variant=bool(cfg["variant"]), attn_layernorm=False).to(DEV)
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        step = T.TrainStep(model, opt, xd, low_d, yd, T.row_weights(tr, n), high_d, un_d, use_graph=True)
        assert model.fused_dropout
        assert step.small is not None, step.small_refused      # tensor adjacencies reach the six-launch step (r06)
        best_val, curr, vals = float("inf"), 0.0, []
        for epoch in range(cfg["epochs"]):
            step()
            out, (acc_te,) = T.evaluate(model, xd, low_d, yd, (te,), high_d, un_d)
            val_loss = float(F.nll_loss(F.log_softmax(out, 1)[va], yd[va]))
            vals.append(val_loss)
            if val_loss < best_val:
                best_val, curr = val_loss, acc_te
            if cfg["early_stopping"] > 0 and epoch > cfg["early_stopping"]:
                if val_loss > np.mean(vals[epoch - cfg["early_stopping"]:epoch]):
                    break
        got.append(curr)
        ref.append(float(rec["test_acc"][si]))
    got, ref = np.asarray(got), np.asarray(ref)
    print(f"\n{name}, fused stack: {100 * got.mean():.2f} +- {100 * got.std():.2f}  (reference run, same splits: "
          f"{100 * ref.mean():.2f} +- {100 * ref.std():.2f})")
    assert abs(got.mean() - ref.mean()) <= 0.015
    assert np.all(np.abs(got - ref) <= 0.04)
