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
