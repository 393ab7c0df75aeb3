"""Row-sharded ACM operators: one process per GPU, RCCL (torch.distributed "nccl")
over xGMI for the halo exchange.

Partition: P contiguous row blocks balanced by WORK, not by row count --
``shard_plan`` cuts the prefix sum of ``nnz(row) + row_cost`` into equal shares (the C ABI's
``acm_shard_plan``; equal rows put 64 % of the edges of a degree-ordered power-law graph on rank 0 of
8).  Blocks may therefore differ in length.  Rank p owns rows [bounds[p], bounds[p+1]) of A_low, of
A_low^T, of X / Z / H / out, of struc_low and of the labels.

Halo layout: every all-gather moves ``n_max = max block length`` rows per rank (shorter blocks are
zero-padded), so the gathered table has P * n_max rows and node j of rank r sits at row
``r * n_max + (j - bounds[r])``.  The local operators are built with their column ids already in that
padded numbering (``ShardPlan.padded_ids``): the gather kernels index the all-gather output directly,
no compaction pass, and with equal blocks the numbering is the identity.

Per layer the only data-path collectives are

    forward : all-gather of the projected features [Z_L | Z_H] (and struc_low rows)
    backward: all-gather of the row-local gradients [G_L | G_H] (and D*G_S)
              + one all-reduce of the (tiny) replicated-parameter gradients

(functional.AcmConvFunction issues them through ``FilterOperators.group``).
The reference is single-process (no collective call sites), so
this module has no reference counterpart; its contract is "N-rank result ==
1-rank result", tested with world_size 2 and 4 on gloo (CPU) with the kernel launches
replaced by a test double, with two ranks on one GPU, and by construction on RCCL.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, tuning
from .graph import CsrGraph, FilterOperators, as_implicit, implicit_form, relabel_by_degree

# Row-local work of a training step expressed in "edges per row": on the twitch-shaped graph the per-row kernels
# (projections, head, K3a, output-layer tail, optimizer) take ~150 us for 168 k rows, the gathers ~180 us for 13.8 M
# edges (profiles/r01_q_step_timeline.txt) => one row costs about as much as 68 gathered edges.
DEFAULT_ROW_COST = 64


class ShardPlan:
    """Contiguous row blocks [bounds[p], bounds[p+1]) for P ranks and the padded halo numbering."""

    def __init__(self, bounds):
        self.bounds = np.asarray(bounds, dtype=np.int64)
        if self.bounds.ndim != 1 or self.bounds.size < 2 or self.bounds[0] != 0 or np.any(np.diff(self.bounds) < 0):
            raise ValueError("ShardPlan: bounds must be a non-decreasing vector starting at 0")
        self.world = self.bounds.size - 1
        self.n_global = int(self.bounds[-1])
        self.n_max = int(np.diff(self.bounds).max())

    def rows(self, rank):
        return int(self.bounds[rank]), int(self.bounds[rank + 1])

    @property
    def uniform(self):
        """Equal blocks: the padded numbering is the identity and the gathered table has n_global rows."""
        return bool(np.all(np.diff(self.bounds) == self.n_max))

    @property
    def n_gathered(self):
        return self.world * self.n_max

    def owner(self, ids):
        return np.searchsorted(self.bounds, np.asarray(ids), side="right") - 1

    def padded_ids(self, ids):
        """Global node ids -> row in the all-gathered (padded) table."""
        ids = np.asarray(ids, dtype=np.int64)
        if self.uniform:
            return ids
        r = np.minimum(self.owner(ids), self.world - 1)
        return r * self.n_max + (ids - self.bounds[r])

    def pad_rows(self, array):
        """A global [n_global, ...] array laid out like the gathered table ([world * n_max, ...], zero padding)."""
        if self.uniform:
            return array
        out = np.zeros((self.n_gathered,) + array.shape[1:], dtype=array.dtype)
        out[self.padded_ids(np.arange(self.n_global))] = array
        return out

    def work(self, indptr, row_cost=0):
        """Per-rank (rows, nnz, nnz + row_cost * rows)."""
        ip = np.asarray(indptr, dtype=np.int64)
        rows = np.diff(self.bounds)
        nnz = ip[self.bounds[1:]] - ip[self.bounds[:-1]]
        return rows, nnz, nnz + row_cost * rows

    def imbalance(self, indptr, row_cost=0):
        """max / mean of the per-rank nnz and of the per-rank work: the factor by which the slowest rank's gathers /
        whole step exceed the average."""
        _, nnz, cost = self.work(indptr, row_cost)
        return float(nnz.max() / max(nnz.mean(), 1e-30)), float(cost.max() / max(cost.mean(), 1e-30))

    def __repr__(self):
        return f"ShardPlan(world={self.world}, rows={np.diff(self.bounds).tolist()})"


def equal_rows_plan(n_global, world):
    """Equal row counts (the node count must be a multiple of the world size: pad the graph)."""
    if n_global % world:
        raise ValueError(f"node count {n_global} is not a multiple of the world size {world}; pad the graph")
    return ShardPlan(np.arange(world + 1, dtype=np.int64) * (n_global // world))


def shard_plan(indptr, world, row_cost=DEFAULT_ROW_COST):
    """Work-balanced contiguous blocks: bounds[p] = the row at which the prefix sum of (nnz(row) + row_cost) crosses
    p / world of its total (acm_shard_plan, host code of the C ABI; binary search over indptr)."""
    ip = np.ascontiguousarray(indptr, dtype=np.int64)
    n = ip.size - 1
    bounds = np.zeros(world + 1, dtype=np.int64)
    st = _lib.load().acm_shard_plan(n, ip.ctypes.data_as(C.c_void_p), int(world), int(row_cost),
                                    bounds.ctypes.data_as(C.c_void_p))
    _lib.check(st, "acm_shard_plan")
    return ShardPlan(bounds)


def shard_bounds(n_global, world, rank):
    """Row range of `rank` under the equal-rows plan."""
    return equal_rows_plan(n_global, world).rows(rank)


def interleave_order(n, world):
    """Permutation (new id -> old id) that deals rows 0, 1, 2, ... to the ranks like cards: new block p = old rows
    p, p + P, p + 2P, ....  Applied to a degree-sorted graph (data.degree_order) it makes EQUAL contiguous blocks
    balanced in rows and in nnz at once (each rank gets every P-th row of the degree ranking, itself still sorted by
    degree), which a contiguous cut of the sorted ranking cannot do: there rank 0 holds the hubs and few rows, the last
    rank many short rows.  n must be a multiple of world."""
    if n % world:
        raise ValueError("interleave_order: pad the node count to a multiple of the world size")
    return np.arange(n, dtype=np.int64).reshape(n // world, world).T.reshape(-1).copy()


def shard_filter_arrays(low_csr, deg, plan, rank):
    """Host-side split of a global scipy CSR A_low (and d) into the arrays rank `rank` needs:
    its rows of A_low and its rows of A_low^T, both with global column ids."""
    b, e = plan.rows(rank)
    low_loc = low_csr[b:e].tocsr()
    low_loc.sort_indices()
    low_t = low_csr.T.tocsr()
    low_t.sort_indices()
    low_t_loc = low_t[b:e].tocsr()
    return low_loc, low_t_loc, (deg[b:e].copy() if deg is not None else None), b


def _padded_columns(mat, plan):
    """scipy CSR with global column ids -> (indptr, padded column ids sorted within each row, values)."""
    m = mat.tocsr()
    m.sort_indices()                               # the padded numbering is monotone in the global id: order is kept
    return m.indptr.astype(np.int32), plan.padded_ids(m.indices).astype(np.int32), m.data.astype(np.float32)


_P61 = (1 << 61) - 1


def _edge_hash(a, b, m, n):
    """A 61-bit hash per (a, b, multiplicity) edge (splitmix64 of the pair's index), as Python-int-safe uint64."""
    with np.errstate(over="ignore"):
        z = (a.astype(np.uint64) * np.uint64(n) + b.astype(np.uint64)) * np.uint64(4) + m.astype(np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z >> np.uint64(3)                              # < 2^61


def _sum_mod_p61(h):
    """sum(h) mod 2^61 - 1 without overflow: 61-bit values summed in two 31-bit halves."""
    lo = int((h & np.uint64(0x7FFFFFFF)).sum(dtype=np.uint64))          # < 2^31 * nnz: fine below 2^33 edges
    hi = int((h >> np.uint64(31)).sum(dtype=np.uint64))
    return (lo + (hi << 31)) % _P61


def pattern_form_of_rows(indptr, indices, vals, row_begin, n_global, group=None, max_multiplicity=4):
    """The pattern-only form (graph.implicit_form) decided from THIS rank's rows alone plus one tiny exchange.

    ``indptr / indices / vals``: the rank's row block of A_low, global column ids, columns ascending inside every row.
    Row-local part: every value an exact small multiple of the row's minimum s_i, no column listed twice.  Global part:
    the multiset {(i, j, m)} must equal {(j, i, m)} -- each rank hashes its edges both ways, the per-rank sums (mod
    2^61 - 1) are all-gathered (16 bytes per rank) and compared: equal multisets give equal sums, different ones
    differ except with probability ~2^-61.  Every rank returns the same decision.
    -> (indptr_P, indices_P, s) of the rank's rows, or None."""
    import torch.distributed as dist
    ip = np.asarray(indptr, dtype=np.int64)
    ix = np.asarray(indices, dtype=np.int64)
    v = np.asarray(vals, dtype=np.float32)
    n_loc = ip.size - 1
    counts = np.diff(ip)
    rows = np.repeat(np.arange(n_loc, dtype=np.int64), counts)
    ok = bool(ix.size == 0 or (v > 0).all())
    s = np.ones(n_loc, np.float32)
    mult = np.ones(ix.size, np.int64)
    if ok and ix.size:
        s = np.full(n_loc, np.inf, np.float32)
        np.minimum.at(s, rows, v)                                        # the row's smallest value
        s = np.where(np.isinf(s), np.float32(1), s).astype(np.float32)   # (empty rows)
        m = v / s[rows]
        mr = np.round(m)
        ok = bool(((mr >= 1) & (mr <= max_multiplicity) & (m == mr)).all())
        mult = mr.astype(np.int64)
        same_row = rows[1:] == rows[:-1]
        ok = ok and not bool((same_row & (ix[1:] <= ix[:-1])).any())      # sorted, coalesced rows
    h_fwd = h_bwd = 0
    if ok:
        g = rows + int(row_begin)
        h_fwd = _sum_mod_p61(_edge_hash(g, ix, mult, n_global))
        h_bwd = _sum_mod_p61(_edge_hash(ix, g, mult, n_global))
    mine = torch.tensor([int(ok), h_fwd, h_bwd], dtype=torch.int64)
    if group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        group = group if group is not None else dist.group.WORLD
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        every = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(dist.get_world_size(group))]
        dist.all_gather(every, mine.to(dev), group=group)
        every = torch.stack(every).cpu()
    else:
        every = mine[None]
    if not bool(every[:, 0].all()):
        return None
    if sum(int(t) for t in every[:, 1]) % _P61 != sum(int(t) for t in every[:, 2]) % _P61:
        return None                                       # the pattern is not symmetric
    if bool((mult == 1).all()):
        return ip.astype(np.int32), ix.astype(np.int32), s
    new_ip = np.zeros(n_loc + 1, np.int64)
    np.add.at(new_ip, rows + 1, mult)
    return np.cumsum(new_ip).astype(np.int32), np.repeat(ix, mult).astype(np.int32), s


def make_sharded_operators_from_rows(indptr, indices, vals, deg_rows, plan, rank, device, group=None, with_structure=False,
                                     t_rows=None, _form="decide"):
    """This rank's FilterOperators from ITS OWN rows of A_low -- (indptr, indices, vals) of the row block
    [plan.rows(rank)) with global column ids sorted inside every row, ``deg_rows`` the block of d = rowsum(I + A) -- so
    that no rank has to hold, sort or transpose the whole graph (a loader can read each rank's rows straight from a
    row-partitioned file).  Pattern-only form whenever A_low allows it (pattern_form_of_rows: a row-local test and a
    16-byte-per-rank exchange for the symmetry of the pattern); otherwise the explicit form, which also needs the
    rank's rows of A_low^T: ``t_rows = (indptr, indices, vals)`` (columns of A_low the rank does not own -- the caller
    has to provide them, make_sharded_operators does from the global matrix)."""
    import torch.distributed as dist
    group = group if group is not None else dist.group.WORLD
    b, e = plan.rows(rank)
    dev = torch.device(device)
    ip = np.asarray(indptr)
    if ip.size != e - b + 1:
        raise ValueError(f"make_sharded_operators_from_rows: {ip.size - 1} rows given, the plan gives rank {rank} {e - b}")

    def dev_t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    common = dict(row_offset=b, n_global=plan.n_global, group=group)
    form = _form                                       # (make_sharded_operators has decided already)
    if isinstance(form, str):
        form = None
        if tuning.HOST.implicit:
            form = pattern_form_of_rows(ip, indices, vals, b, plan.n_global, group)
    if form is not None:
        ipp, ixp, s = form
        pat = CsrGraph.from_csr(dev_t(ipp.astype(np.int32)), dev_t(plan.padded_ids(ixp).astype(np.int32)), None, plan.n_gathered)
        ops = FilterOperators(pat, dev_t(deg_rows) if with_structure else None, row_scale=dev_t(s), **common)
        ops.low_t_override = pat
    else:
        if t_rows is None:
            raise ValueError("make_sharded_operators_from_rows: A_low has no pattern-only form here (or ACM_IMPLICIT=0); the "
                             "explicit form needs the rank's rows of A_low^T (t_rows)")
        ops = FilterOperators(CsrGraph.from_csr(dev_t(ip.astype(np.int32)), dev_t(plan.padded_ids(indices).astype(np.int32)),
                                                dev_t(np.asarray(vals, np.float32)), plan.n_gathered),
                              dev_t(deg_rows) if with_structure else None, **common)
        tip, tix, tv = t_rows
        ops.low_t_override = CsrGraph.from_csr(dev_t(np.asarray(tip).astype(np.int32)),
                                               dev_t(plan.padded_ids(tix).astype(np.int32)),
                                               dev_t(np.asarray(tv, np.float32)), plan.n_gathered)
    ops.plan = plan
    return ops


def make_sharded_operators(low_csr, deg, device, group=None, with_structure=False, plan=None, row_cost=DEFAULT_ROW_COST,
                           relabel=False):
    """FilterOperators for this rank (or the unsharded ones when no process group is active).  ``plan``: a ShardPlan
    (default: work-balanced blocks of this graph, identical on every rank because it is a function of indptr only).
    ``relabel`` (single process): sort the node numbering by degree inside the operator (graph.relabel_by_degree)."""
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        low = CsrGraph.from_scipy(low_csr, device)
        d = torch.from_numpy(np.ascontiguousarray(deg)).to(device) if with_structure else None
        ops = as_implicit(FilterOperators(low, d))
        return relabel_by_degree(ops) if relabel else ops
    group = group if group is not None else dist.group.WORLD
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    low_csr = low_csr.tocsr()
    if plan is None:
        plan = shard_plan(low_csr.indptr, world, row_cost)
    if plan.world != world or plan.n_global != low_csr.shape[0]:
        raise ValueError(f"{plan} does not match {world} ranks / {low_csr.shape[0]} rows")
    # every rank cuts ITS rows out of the host matrix and builds from those alone (make_sharded_operators_from_rows: what a
    # loader with a row-partitioned file calls directly); only the explicit form also needs the rank's rows of A_low^T
    b, e = plan.rows(rank)
    loc = low_csr[b:e].tocsr()
    loc.sort_indices()
    form = None
    if tuning.HOST.implicit:
        form = pattern_form_of_rows(loc.indptr, loc.indices, loc.data, b, plan.n_global, group)
    t_rows = None
    if form is None:
        _, low_t_loc, _, _ = shard_filter_arrays(low_csr, deg, plan, rank)
        low_t_loc.sort_indices()
        t_rows = (low_t_loc.indptr, low_t_loc.indices, low_t_loc.data)
    return make_sharded_operators_from_rows(loc.indptr, loc.indices, loc.data, deg[b:e] if deg is not None else None,
                                            plan, rank, device, group, with_structure, t_rows, _form=form)


def local_rows(array, plan, rank):
    b, e = plan.rows(rank)
    return array[b:e]


def local_index(idx, plan, rank, n_global=None):
    """Global node indices -> indices into this rank's row block (only the owned ones).  ``plan``: a ShardPlan, or
    (legacy) the world size of an equal-rows plan over ``n_global`` nodes."""
    if not isinstance(plan, ShardPlan):
        plan = equal_rows_plan(n_global, int(plan))
    b, e = plan.rows(rank)
    idx = np.asarray(idx)
    own = idx[(idx >= b) & (idx < e)]
    return own - b
