"""Deterministic dropout masks shared by the golden generator (which patches them into the
reference) and the GPU accuracy test (which patches them into this package): the same CPU
generator stream on both sides, so two implementations train on identical masks."""
import torch


class SeededDropout:
    """Drop-in for torch.nn.functional.dropout.  Mask k of epoch e comes from a CPU generator
    seeded with (seed, e, k); call .next_epoch() once per training step."""

    def __init__(self, seed, device="cpu"):
        self.seed, self.epoch, self.site, self.device = int(seed), 0, 0, device

    def next_epoch(self):
        self.epoch += 1
        self.site = 0

    def __call__(self, inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + self.epoch * 101 + self.site)
        self.site += 1
        keep = torch.bernoulli(torch.full(tuple(inp.shape), 1.0 - p), generator=g).to(inp.device)
        return inp * keep / (1.0 - p)


class PhiloxDropout:
    """Drop-in for torch.nn.functional.dropout that hands the REFERENCE the masks this library's kernels draw themselves
    (counter-based dropout: oracle/philox.py restates csrc/acm_common.h) -- so a reference run can be recorded that the
    fused small-graph step (acm_small_step: masks generated inside the kernels, never materialised) replays exactly.

    Two-layer acmgcn / acmgcnp models call F.dropout twice per training forward (ACM-Pytorch/models/models.py:116,160):
    site 0 = the input features (tag 0; the library keys the mask of a CSR feature matrix by the position of the entry in
    the row-major sorted nonzero list, column 0), site 1 = the hidden activations [n, 64] (tag 1, keyed by (row, column)).
    Step counter: 0 for the first optimizer step; call .next_epoch() before every training step."""

    def __init__(self, seed, features):
        from oracle.philox import dropout_factors
        self._factors = dropout_factors
        self.seed, self.step, self.site = int(seed), -1, 0
        nz = torch.nonzero(features)                    # row-major sorted, like the coalesced COO behind SparseFeatures.auto
        self._flat = nz[:, 0] * features.shape[1] + nz[:, 1]
        self._shape = tuple(features.shape)

    def next_epoch(self):
        self.step += 1
        self.site = 0

    def __call__(self, inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        site, self.site = self.site, self.site + 1
        if site == 0:
            assert tuple(inp.shape) == self._shape, "site 0 is the input feature matrix"
            f = self._factors(self.seed, self.step, 0, p, self._flat.numel(), 1)[:, 0]
            m = torch.zeros(inp.numel(), dtype=inp.dtype)
            m[self._flat] = torch.from_numpy(f).to(inp.dtype)
            return inp * m.view(inp.shape).to(inp.device)
        assert site == 1, "two dropout sites per forward (acmgcn / acmgcnp)"
        f = self._factors(self.seed, self.step, 1, p, inp.shape[0], inp.shape[1])
        return inp * torch.from_numpy(f).to(inp.dtype).to(inp.device)
