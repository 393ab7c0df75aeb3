import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from acm_gnn_amd import functional as AF
dev = torch.device("cuda", 0)
n, k, nn = 169343, 128, 192
x = torch.randn(n, k, device=dev); w = torch.randn(k, nn, device=dev); z = torch.empty(n, nn, device=dev); dz = torch.randn(n, nn, device=dev); dw = torch.empty(k, nn, device=dev)
for _ in range(10):
    AF.gemm(x, w, out=z)
    AF.gemm(x, dz, trans_a=True, out=dw)
torch.cuda.synchronize()
