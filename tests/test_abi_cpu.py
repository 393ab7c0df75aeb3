"""The C ABI: the shared library builds/loads here (no GPU needed), exports every function
include/acm_hip.h declares, and the ctypes mirrors of its structs have the C layout."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "acm_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acm_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from acm_gnn_amd import _lib
    lib = _lib.load()
    assert lib.acm_version() == _lib.ABI_VERSION
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in acm_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_calls_without_gpu_fail_loudly_not_silently():
    """On this CPU-only box a compute call must return an error code + message, never succeed."""
    from acm_gnn_amd import _lib
    lib = _lib.load()
    out = C.c_void_p()
    st = lib.acm_csr_create(0, 0, 0, None, None, None, 0, C.byref(out))
    assert st == 1 and b"indptr" in lib.acm_last_error()                # ACM_EINVAL
    nbytes = C.c_size_t()
    assert lib.acm_conv_bwd_local_workspace_bytes(100, 64, 5, C.byref(nbytes)) == 2   # ACM_ESHAPE
    assert lib.acm_conv_agg_bwd_workspace_bytes(100, 40, 64, C.byref(nbytes)) == 4    # ACM_EUNSUPPORTED
    assert lib.acm_gemm(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, 0, None, 0, None) == 1


STRUCTS = {"acm_csr_info_t": "CsrInfo", "acm_conv_fwd_t": "ConvFwd", "acm_conv_bwd_local_t": "ConvBwdLocal",
           "acm_conv_bwd_spmm_t": "ConvBwdSpmm", "acm_conv_agg_fwd_t": "ConvAggFwd", "acm_conv_agg_bwd_t": "ConvAggBwd",
           "acm_spmm_opts_t": "SpmmOpts", "acm_dropout_t": "Dropout", "acm_adam_tensor_t": "AdamTensor", "acm_adam_config_t": "AdamConfig"}


def test_ctypes_struct_layouts_match_the_c_header(tmp_path):
    """Compile a tiny C program against include/acm_hip.h that prints sizeof and the offset of
    every field, and compare with the ctypes mirrors in acm_gnn_amd/_lib.py."""
    from acm_gnn_amd import _lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "acm_hip.h"', "int main(void){"]
    for cname, pyname in STRUCTS.items():
        cls = getattr(_lib, pyname)
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, pyname in STRUCTS.items():
        cls = getattr(_lib, pyname)
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_graft_entry_build_hook_exists():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
