"""Row-sharded ACM operators: one process per GPU, RCCL (torch.distributed "nccl")
over xGMI for the halo exchange.

Partition: contiguous, equal-sized row blocks (the node count is padded to a
multiple of the world size with isolated dummy nodes, see data.synthetic_dataset).
Rank p owns rows [p*n_loc, (p+1)*n_loc) of A_low, of A_low^T, of X / Z / H / out,
of struc_low and of the labels.  Per layer the only data-path collectives are

    forward : all-gather of the projected features [Z_L | Z_H] (and struc_low rows)
    backward: all-gather of the row-local gradients [G_L | G_H] (and D*G_S)
              + one all-reduce of the (tiny) replicated-parameter gradients

(functional.AcmConvFunction issues them through ``FilterOperators.group``).
The reference is single-process (no collective call sites), so
this module has no reference counterpart; its contract is "N-rank result ==
1-rank result", tested with world_size 2 on gloo (CPU) with the kernel launches
replaced by a test double, and by construction on RCCL.
"""
import os

import numpy as np
import torch

from .graph import CsrGraph, FilterOperators, as_implicit, implicit_form


def shard_bounds(n_global, world, rank):
    if n_global % world:
        raise ValueError(f"node count {n_global} is not a multiple of the world size {world}; pad the graph")
    n_loc = n_global // world
    return rank * n_loc, (rank + 1) * n_loc


def shard_filter_arrays(low_csr, deg, world, rank):
    """Host-side split of a global scipy CSR A_low (and d) into the arrays rank `rank` needs:
    its rows of A_low and its rows of A_low^T, both with global column ids."""
    n = low_csr.shape[0]
    b, e = shard_bounds(n, world, rank)
    low_loc = low_csr[b:e].tocsr()
    low_loc.sort_indices()
    low_t = low_csr.T.tocsr()
    low_t.sort_indices()
    low_t_loc = low_t[b:e].tocsr()
    return low_loc, low_t_loc, (deg[b:e].copy() if deg is not None else None), b


def make_sharded_operators(low_csr, deg, device, group=None, with_structure=False):
    """FilterOperators for this rank (or the unsharded ones when no process group is active)."""
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        low = CsrGraph.from_scipy(low_csr, device)
        d = torch.from_numpy(np.ascontiguousarray(deg)).to(device) if with_structure else None
        return as_implicit(FilterOperators(low, d))
    group = group if group is not None else dist.group.WORLD
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    form = None
    if os.environ.get("ACM_IMPLICIT", "1") != "0":
        # pattern-only form, detected on the global matrix (host): the rank's rows of P serve A_low (row-scaled)
        # and, P being symmetric, A_low^T on the pre-scaled all-gathered gradients -- one column-id stream, no
        # transposed slice
        g = low_csr.tocsr()
        g.sort_indices()
        form = implicit_form(torch.from_numpy(g.indptr.astype(np.int64)), torch.from_numpy(g.indices.astype(np.int64)),
                             torch.from_numpy(g.data.astype(np.float32)), g.shape[0], g.shape[1])
    if form is not None:
        ip, ix, s = (t.numpy() for t in form)
        b, e = shard_bounds(low_csr.shape[0], world, rank)
        ip_loc = (ip[b:e + 1] - ip[b]).astype(np.int32)
        ix_loc = ix[ip[b]:ip[e]]
        dev = torch.device(device)
        pat = CsrGraph.from_csr(torch.from_numpy(ip_loc).to(dev), torch.from_numpy(np.ascontiguousarray(ix_loc)).to(dev),
                                None, low_csr.shape[1])
        ops = FilterOperators(pat, torch.from_numpy(np.ascontiguousarray(deg[b:e])).to(dev) if with_structure else None,
                              row_offset=b, n_global=low_csr.shape[0], group=group,
                              row_scale=torch.from_numpy(np.ascontiguousarray(s[b:e])).to(dev))
        ops.low_t_override = pat
        return ops
    low_loc, low_t_loc, deg_loc, b = shard_filter_arrays(low_csr, deg, world, rank)
    ops = FilterOperators(CsrGraph.from_scipy(low_loc, device),
                          torch.from_numpy(np.ascontiguousarray(deg_loc)).to(device) if with_structure else None,
                          row_offset=b, n_global=low_csr.shape[0], group=group)
    ops.low_t_override = CsrGraph.from_scipy(low_t_loc, device)
    return ops


def local_rows(array, world, rank):
    b, e = shard_bounds(array.shape[0], world, rank)
    return array[b:e]


def local_index(idx, world, rank, n_global):
    """Global node indices -> indices into this rank's row block (only the owned ones)."""
    b, e = shard_bounds(n_global, world, rank)
    idx = np.asarray(idx)
    own = idx[(idx >= b) & (idx < e)]
    return own - b
