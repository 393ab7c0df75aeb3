"""The C ABI without Python in the loop: plain-C programs link libacm_hip.so and check it against host loops --
tests/c_abi/abi_smoke.c: SpMM (explicit and pattern-only handles), GEMM, the Adam step, the loss;
tests/c_abi/abi_layer.c: the layer operator itself (acm_conv_fwd / acm_conv_bwd_local / acm_conv_bwd_spmm and the
aggregate-first pair), forward against a double-precision restatement of ACM-Geometric/layers.py:57-63,101-108 and
gradients against finite differences of it."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prog", ["abi_smoke", "abi_layer"])
def test_plain_c_client(prog, tmp_path):
    from acm_gnn_amd import _lib
    _lib.load()                                              # make sure the library is built
    lib_dir = os.path.join(ROOT, "acm_gnn_amd", "lib")
    exe = str(tmp_path / prog)
    subprocess.check_call(["gcc", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c_abi", prog + ".c"),
                           "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L", lib_dir, "-lacm_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert prog + " ok" in out.stdout
    print(out.stdout)
