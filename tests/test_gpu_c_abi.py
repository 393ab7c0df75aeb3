"""The C ABI without Python in the loop: a plain-C program (tests/c_abi/abi_smoke.c) links libacm_hip.so, builds a
graph, and checks SpMM (explicit and pattern-only handles), GEMM and the Adam step against host loops."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client(tmp_path):
    from acm_gnn_amd import _lib
    _lib.load()                                              # make sure the library is built
    lib_dir = os.path.join(ROOT, "acm_gnn_amd", "lib")
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L", lib_dir, "-lacm_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout
