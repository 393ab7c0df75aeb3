"""acm_conv_acmii_fwd from two builds of the same source: MFMA results in VGPRs (-amdgpu-mfma-vgpr-form, the shipped
library) and in AGPRs (lib/libacm_hip_acmii_agpr.so, built beside it by acm_gnn_amd/build.py).  Same arithmetic in the
same order, so the outputs must be BIT-IDENTICAL; a difference means a VALU instruction read an MFMA result before it
had landed -- the hazard round 3 met with inline-assembly reads, which the compiler's hazard recogniser does not see
(the reads are builtins now: acm_conv_acmii.hip: abs_add).  Each build runs in its own process."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, %(root)r)
from acm_gnn_amd import GraphConvolution, tuning
from oracle import acm_oracle as O
out_path, k = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(3)
n = 3000
a = sp.random(n, n, density=0.012, random_state=rng, format="csr")
a = ((a + a.T) > 0).astype(np.float64).tolil()
a[0, 1:1500] = 1
a[1:1500, 0] = 1                                    # a hub row: pieces + the fix-up kernel
a.setdiag(0)
low, high, un = O.filters_linkx(sp.csr_matrix(a))
dev = "cuda:0"
torch.manual_seed(1)
layer = GraphConvolution(7, 64, n, "acmgcnp", variant=True, structure_info=int(k == 4), attn_layernorm=True).to(dev)
x = torch.randn(n, 7, generator=torch.Generator().manual_seed(2)).to(dev)
from acm_gnn_amd import functional as AF
timer = AF.KernelTimer()
AF.set_kernel_timer(timer)
with torch.no_grad(), tuning.override(rewrites=3):       # the fp32-MFMA kernel (the default for k = 3 is the bf16 mask form)
    out = layer(x, low.to(dev), high.to(dev), un.to(dev) if k == 4 else None)
AF.set_kernel_timer(None)
assert any(key.startswith("conv_acmii_fwd") for key in timer.events), sorted(timer.events)
np.save(out_path, out.cpu().numpy())
"""


@pytest.mark.parametrize("k", [3, 4])
def test_acmii_forward_is_bit_identical_with_and_without_mfma_vgpr_form(k, tmp_path):
    from acm_gnn_amd import build as B
    assert os.path.exists(B.VARIANT_PATH), "python -m acm_gnn_amd.build builds lib/libacm_hip_acmii_agpr.so beside the library"
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT})
    outs = []
    for lib in (B.LIB_PATH, B.VARIANT_PATH):
        path = tmp_path / (os.path.basename(lib) + ".npy")
        env = dict(os.environ, ACM_HIP_LIBRARY=lib)
        subprocess.run([sys.executable, str(script), str(path), str(k)], env=env, check=True, cwd=ROOT, timeout=600)
        outs.append(np.load(path))
    assert np.isfinite(outs[0]).all() and float(np.abs(outs[0]).max()) > 0
    assert np.array_equal(outs[0], outs[1]), float(np.abs(outs[0] - outs[1]).max())
