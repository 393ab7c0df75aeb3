"""Synthetic graphs of the benchmark shapes and host-side filter construction.

No dataset can be downloaded (no network), so the LINKX-scale workloads are
seeded synthetic graphs with the node / edge / feature counts of the real
datasets.  Filter construction restates
ACM-Geometric/train.py:66-81 (symmetrise, A_low = D^-1 (I + A) in float64,
cast to float32) with scipy; it is one-off preprocessing, not the hot path.
"""
import numpy as np
import scipy.sparse as sp

# (nodes, undirected edges, input features, classes) -- LINKX dataset facts
SHAPES = {
    "twitch-gamer": (168_114, 6_797_557, 7, 2),
    "arxiv-year": (169_343, 1_166_243, 128, 5),
    "penn94": (41_554, 1_362_229, 4_814, 2),
    "squirrel": (5_201, 198_353, 2_089, 5),
    "tiny": (2_000, 20_000, 7, 2),
}


def _powerlaw_weights(n, mean_deg, max_deg, gamma=2.1):
    """Chung-Lu expected degrees w_i ~ (i + i0)^(-1/(gamma-1)) scaled to mean_deg with w_0 ~ max_deg."""
    alpha = 1.0 / (gamma - 1.0)
    target = max_deg / mean_deg
    lo, hi = 1e-3, float(n)
    i = np.arange(n, dtype=np.float64)
    for _ in range(60):
        mid = np.sqrt(lo * hi)
        w = (i + mid) ** (-alpha)
        if w[0] / w.mean() > target:
            lo = mid
        else:
            hi = mid
    w = (i + np.sqrt(lo * hi)) ** (-alpha)
    return w * (mean_deg / w.mean())


def chung_lu_graph(n, n_edges, max_deg, seed=0, gamma=2.1, uniform=False):
    """Simple undirected graph with exactly n_edges edges; returns symmetric CSR (0/1, no self loops).

    Node ids are randomly permuted so that degree does not correlate with id (as in
    crawled social graphs)."""
    rng = np.random.default_rng(seed)
    if uniform:
        p = None
    else:
        w = _powerlaw_weights(n, 2.0 * n_edges / n, max_deg, gamma)
        p = w / w.sum()
        cdf = np.cumsum(p)
    keys = np.empty(0, dtype=np.int64)
    while keys.size < n_edges:
        need = int((n_edges - keys.size) * 1.25) + 1024
        if p is None:
            u = rng.integers(0, n, need)
            v = rng.integers(0, n, need)
        else:
            u = np.searchsorted(cdf, rng.random(need)).clip(0, n - 1)
            v = np.searchsorted(cdf, rng.random(need)).clip(0, n - 1)
        ok = u != v
        a, b = np.minimum(u[ok], v[ok]), np.maximum(u[ok], v[ok])
        keys = np.unique(np.concatenate([keys, a.astype(np.int64) * n + b]))
    if keys.size > n_edges:
        keys = rng.choice(keys, n_edges, replace=False)
    u, v = keys // n, keys % n
    perm = rng.permutation(n)
    u, v = perm[u], perm[v]
    rows = np.concatenate([u, v])
    cols = np.concatenate([v, u])
    a = sp.csr_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(n, n))
    a.sum_duplicates()
    a.data[:] = 1.0
    a.sort_indices()
    return a


def synthetic_dataset(name, seed=0, uniform=False, pad_to=1):
    """(adjacency CSR, features f32 [n, f_in], labels int64 [n], train/val/test index arrays, n_real).

    ``pad_to`` > 1 appends isolated dummy nodes so that n is a multiple of it (row sharding)."""
    n, e, f_in, c = SHAPES[name]
    max_deg = {"twitch-gamer": 35_000, "arxiv-year": 13_000, "penn94": 4_400, "squirrel": 1_900,
               "tiny": 300}[name]
    adj = chung_lu_graph(n, e, max_deg, seed=seed, uniform=uniform)
    rng = np.random.default_rng(seed + 1)
    if name == "penn94":                                   # one-hot style sparse binary features
        x = np.zeros((n, f_in), np.float32)
        for j in range(5):
            x[np.arange(n), rng.integers(0, f_in, n)] = 1.0
    else:
        x = rng.standard_normal((n, f_in)).astype(np.float32)   # standardised, dataset.py:380-382
    y = rng.integers(0, c, n).astype(np.int64)
    order = rng.permutation(n)
    n_tr, n_va = int(0.5 * n), int(0.25 * n)                # parse.py:46-49 (50/25/25)
    splits = (np.sort(order[:n_tr]), np.sort(order[n_tr:n_tr + n_va]), np.sort(order[n_tr + n_va:]))
    n_real = n
    if pad_to > 1 and n % pad_to:
        n_pad = (n + pad_to - 1) // pad_to * pad_to
        adj = sp.csr_matrix((adj.data, adj.indices, np.concatenate([adj.indptr, np.full(n_pad - n, adj.indptr[-1])])),
                            shape=(n_pad, n_pad))
        x = np.concatenate([x, np.zeros((n_pad - n, f_in), np.float32)])
        y = np.concatenate([y, np.zeros(n_pad - n, np.int64)])
    return adj, x, y, splits, n_real


def degree_order(adj):
    """Permutation `perm` (new id -> old id) that lists nodes by decreasing degree, ties by old id.
    Relabelling a graph this way makes the rows that most edges point at (hubs) contiguous, so the
    gathered operand's hot part stays cache/LDS resident.  Pure relabelling: results are the same
    up to the permutation."""
    deg = np.diff(adj.indptr)
    return np.lexsort((np.arange(adj.shape[0]), -deg)).astype(np.int64)


def permute_dataset(adj, x, y, splits, perm):
    """Apply new id i <- old id perm[i] consistently to the adjacency, features, labels, splits."""
    n = adj.shape[0]
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    a = adj[perm][:, perm].tocsr()
    a.sort_indices()
    new_splits = tuple(np.sort(inv[s]) for s in splits)
    return a, x[perm], y[perm], new_splits


def build_filters(adj):
    """A_low = D^-1 (I + A) computed in float64, returned as float32 CSR, plus d = rowsum(I + A)
    (ACM-Geometric/train.py:76-81, utils.py:5-19).  A_high = I - A_low is implied."""
    n = adj.shape[0]
    m = (sp.identity(n, format="csr", dtype=np.float64) + adj.astype(np.float64)).tocsr()
    deg = np.asarray(m.sum(1)).flatten()
    with np.errstate(divide="ignore"):
        inv = np.power(deg, -1.0)
    inv[np.isinf(inv)] = 0.0
    low = sp.diags(inv, 0).dot(m).tocsr().astype(np.float32)
    low.sort_indices()
    return low, deg.astype(np.float32)


def row_normalize_features(x):
    """Divide each feature row by its sum, inf -> 0, in the input precision and through scipy
    exactly as the reference does (train.py:69-73 -> utils.py:5-19; skipped for
    acmgcnp + structure_info)."""
    m = sp.csr_matrix(x)
    rowsum = np.asarray(m.sum(1)).flatten()
    with np.errstate(divide="ignore"):
        inv = np.power(rowsum, -1.0)
    inv[np.isinf(inv)] = 0.0
    return np.asarray(sp.diags(inv, 0).dot(m).todense()).astype(np.float32)


def bench_workload(dataset="twitch-gamer", seed=0, node_order="degree", normalize_features=True, uniform=False,
                   pad_to=1):
    """The synthetic workload of bench.py as one call (the full-size parity tests build exactly this):
    Chung-Lu graph of the dataset's shape, optional relabelling of the real nodes by decreasing degree (data
    preparation: results are identical up to the permutation), feature row normalisation (train.py:69-73; the
    caller passes normalize_features=False for acmgcnp/acmgcnpp with structure_info), A_low and d.

    Returns a dict: adj (scipy CSR, 0/1), x, y, splits (train, val, test), n_real, low (scipy CSR fp32), deg."""
    adj, x, y, splits, n_real = synthetic_dataset(dataset, seed=seed, uniform=uniform, pad_to=pad_to)
    if node_order == "degree":
        perm = degree_order(adj[:n_real][:, :n_real].tocsr())
        full = np.concatenate([perm, np.arange(n_real, adj.shape[0])])
        adj, x, y, splits = permute_dataset(adj, x, y, splits, full)
    elif node_order != "random":
        raise ValueError("node_order: 'degree' or 'random'")
    if normalize_features:
        x = row_normalize_features(x)
    low, deg = build_filters(adj)
    return dict(adj=adj, x=x, y=y, splits=splits, n_real=n_real, low=low, deg=deg)
