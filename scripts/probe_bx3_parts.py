import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import functional as AF
dev = torch.device("cuda", 0)
def timeit(fn, reps=10, inner=10):
    """us per call: `inner` calls captured in one graph (no host time between the launches), `reps` replays"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
    g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * inner) * 1e6
if __name__ == '__main__':
  n, k = 169343, 128
  for nn in (192, 21):
      x = torch.randn(n, k, device=dev); w = torch.randn(k, nn, device=dev); z = torch.empty(n, nn, device=dev)
      for blocks in ("", "512", "128"):
          for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
              os.environ["ACM_GEMM_BX3_DBG"] = str(dbg)
              if blocks: os.environ["ACM_GEMM_BX3_BLOCKS"] = blocks
              else: os.environ.pop("ACM_GEMM_BX3_BLOCKS", None)
              print(json.dumps({"N": nn, "blocks": blocks or "auto", "dbg(1=nostore,2=nomfma,4=noload)": dbg, "us": round(timeit(lambda: AF.gemm(x, w, out=z)), 1)}), flush=True)

