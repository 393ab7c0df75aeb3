import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import functional as AF
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from probe_bx3_parts import timeit  # noqa
dev = torch.device("cuda", 0)
n, k = 169343, 128
for nn in (192, 15):
    x = torch.randn(n, k, device=dev); dz = torch.randn(n, nn, device=dev); dw = torch.empty(k, nn, device=dev)
    for dbg in (0, 1, 2, 3, 4, 5, 7):
        os.environ["ACM_GEMM_BX3_DBG"] = str(dbg)
        print(json.dumps({"N": nn, "dbg(1=nofeed,2=nopark,4=noload)": dbg, "us": round(timeit(lambda: AF.gemm(x, dz, trans_a=True, out=dw)), 1)}), flush=True)
