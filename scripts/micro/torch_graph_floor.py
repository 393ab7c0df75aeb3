"""Per-node cost of small kernels inside a torch-captured hipGraph (compare with launch_floor.hip)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acm_gnn_amd import functional as AF

dev = torch.device("cuda:0")


def timed(name, fn, nodes, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    print(f"{name:50s} {a.elapsed_time(b) * 1000 / (reps * nodes):7.2f} us per node")


n = 168114
logits = torch.randn(n, 2, device=dev)
labels = torch.randint(0, 2, (n,), device=dev)
w = torch.ones(n, device=dev) / n
x = torch.zeros(1024, device=dev)
big = torch.zeros(n, 64, device=dev)


def many_nll():
    for _ in range(50):
        AF.nll_loss_and_grad(logits, labels, w)


def many_add():
    for _ in range(100):
        x.add_(1.0)


def many_add_big():
    for _ in range(100):
        big.add_(1.0)


def alt():
    for _ in range(50):
        big.add_(1.0)
        x.add_(1.0)


timed("torch x.add_(1) on 1024 floats", many_add, 100)
timed("torch big.add_(1) on 43 MB", many_add_big, 100)
timed("alternate big / small add_", alt, 100)
timed("acm nll (rows + final) on 168k x 2", many_nll, 100)
