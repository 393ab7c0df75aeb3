#!/usr/bin/env python3
"""acm_conv_agg_bwd with the carried gather, role by role (ACM_AGG_BWD_ROLES = 1 backward waves only, 2 gather waves only,
3 both): eager per-kernel times of the pipelined headline step."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from probe_pipeline import build  # noqa: E402
from acm_gnn_amd import data as D, functional as AF  # noqa: E402


def main():
    wl = D.bench_workload("twitch-gamer", seed=0, node_order="degree")
    step, model = build(None, wl, use_graph=False)
    for roles in ("3", "1", "2", "3"):
        os.environ["ACM_AGG_BWD_ROLES"] = roles
        for _ in range(5):
            step()
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        for _ in range(10):
            step()
        AF.set_kernel_timer(None)
        s = timer.summary()
        print(f"roles {roles}: " + "  ".join(f"{k} {tot / cnt * 1e3:.1f}" for k, (cnt, tot) in s.items() if "agg" in k), flush=True)


if __name__ == "__main__":
    main()
