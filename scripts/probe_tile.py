import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from acm_gnn_amd import data as D, functional as AF
from acm_gnn_amd.graph import CsrGraph
DEV = torch.device("cuda:0")
adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset("twitch-gamer")
perm = D.degree_order(adj)
adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
low, deg = D.build_filters(adj)
low = low.tocsr(); low.sort_indices()
g = CsrGraph.from_csr(torch.from_numpy(low.indptr.astype(np.int32)).to(DEV), torch.from_numpy(low.indices.astype(np.int32)).to(DEV), None, n)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for w in (64, 32, 16, 8, 4):
    x = torch.randn(n, w, device=DEV); y = torch.empty(n, w, device=DEV)
    t = timeit(lambda: AF.spmm(g, x, out=y))
    print(f"width {w:3d}: {t:7.1f} us per pass, x{64 // w} passes = {t * 64 / w:7.1f} us for 64 columns", flush=True)
# strided slices of one 64-wide table (what a column-tiled pass over an existing activation would read)
x = torch.randn(n, 64, device=DEV); y = torch.empty(n, 64, device=DEV)
for w in (32, 16, 8):
    def run():
        for c in range(0, 64, w):
            AF.spmm(g, x[:, c:c + w], out=y[:, c:c + w])
    try:
        print(f"slices of width {w} of a 64-wide table: {timeit(run):7.1f} us", flush=True)
    except Exception as e:
        print("slices", w, "failed:", repr(e)[:200])
