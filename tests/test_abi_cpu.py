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
           "acm_spmm_opts_t": "SpmmOpts", "acm_dropout_t": "Dropout", "acm_adam_tensor_t": "AdamTensor", "acm_adam_config_t": "AdamConfig",
           "acm_tuning_t": "Tuning", "acm_small_step_t": "SmallStep"}


def _struct(pyname):
    from acm_gnn_amd import _lib, tuning
    return tuning.Tuning if pyname == "Tuning" else getattr(_lib, pyname)


def test_ctypes_struct_layouts_match_the_c_header(tmp_path):
    """Compile a tiny C program against include/acm_hip.h that prints sizeof and the offset of
    every field, and compare with the ctypes mirrors in acm_gnn_amd/_lib.py."""
    from acm_gnn_amd import _lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "acm_hip.h"', "int main(void){"]
    for cname, pyname in STRUCTS.items():
        cls = _struct(pyname)
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
        cls = _struct(pyname)
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_tuning_record_is_the_one_dispatch_mechanism():
    """acm_tuning_t: layout, get / set / validation / reset through the real library; no acm_* launch path reads the
    environment (no getenv in any .hip translation unit; one read of ACM_TUNING at load time in acm_csr.cpp)."""
    import glob
    from acm_gnn_amd import _lib, tuning
    lib = _lib.load()
    t = tuning.Tuning()
    assert C.sizeof(t) == 4 * 16
    assert lib.acm_tuning_get(C.byref(t)) == 0
    loaded = {k: getattr(t, k) for k in tuning.KERNEL_KEYS}
    assert loaded == dict(chunk=0, wide_form=0, bwd_split=-1, rows16=7, agg_fused=1, gemm_forms=7), loaded
    t.rows16, t.chunk = 5, 256
    assert lib.acm_tuning_set(C.byref(t)) == 0
    u = tuning.Tuning()
    lib.acm_tuning_get(C.byref(u))
    assert (u.rows16, u.chunk) == (5, 256)
    t.chunk = 100                                         # not a power of two
    assert lib.acm_tuning_set(C.byref(t)) == 1 and b"chunk" in lib.acm_last_error()
    lib.acm_tuning_get(C.byref(u))
    assert (u.rows16, u.chunk) == (5, 256)                # a rejected record changes nothing
    assert lib.acm_tuning_set(None) == 0                  # NULL: back to the load-time record
    lib.acm_tuning_get(C.byref(u))
    assert {k: getattr(u, k) for k in tuning.KERNEL_KEYS} == loaded
    tuning.invalidate()
    for path in glob.glob(os.path.join(ROOT, "acm_gnn_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "acm_gnn_amd", "csrc", "*.h")):
        assert "getenv" not in open(path).read(), path
    host = open(os.path.join(ROOT, "acm_gnn_amd", "csrc", "acm_csr.cpp")).read()
    assert host.count("getenv(") == 1 and 'getenv("ACM_TUNING")' in host
    for path in glob.glob(os.path.join(ROOT, "acm_gnn_amd", "*.py")) + glob.glob(os.path.join(ROOT, "acm_gnn_amd", "*", "*.py")):
        if os.path.basename(path) in ("tuning.py", "_lib.py", "build.py"):      # ACM_TUNING; the library path; HIPCC
            continue
        assert "os.environ" not in open(path).read(), path


def test_acm_tuning_variable_is_read_once_at_load(tmp_path):
    """ACM_TUNING in a child's environment: the library's record and the host record both take it at load / import; later
    changes of the variable do nothing; unknown keys are reported (stderr by the library, ValueError by the host parser)."""
    code = (
        "import ctypes as C, os, sys\n"
        "from acm_gnn_amd import _lib, tuning\n"
        "_lib.load()\n"
        "os.environ['ACM_TUNING'] = 'rows16=0'\n"          # too late: both records were filled at load / import
        "k = tuning.kernel()\n"
        "print(k['rows16'], k['gemm_forms'], k['wide_form'], tuning.HOST.rewrites, tuning.HOST.pipeline, tuning.HOST.relabel)\n")
    env = dict(os.environ, ACM_TUNING="rows16=5,gemm_forms=1,rewrites=2,pipeline=0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    assert out.stdout.split() == ["5", "1", "0", "2", "0", "-1"], (out.stdout, out.stderr)
    env["ACM_TUNING"] = "rows16=9,nonsense=1"
    from acm_gnn_amd import _lib
    out = subprocess.run([sys.executable, "-c", f"import ctypes\nctypes.CDLL({_lib.library_path()!r})"], env=env,
                         capture_output=True, text=True)                   # the library alone: reports and ignores the items
    assert "bad value 'rows16=9'" in out.stderr and "unknown key 'nonsense'" in out.stderr, out.stderr
    # ... and the package does the same with it (one policy): reported, ignored, the valid items taken
    code2 = "import acm_gnn_amd\nfrom acm_gnn_amd import tuning\nprint(tuning.HOST.pipeline)\n"
    env2 = dict(env, ACM_TUNING="pipeline=77,nonsense=1,relabel=7", ACM_GATHER_DTYPE="bf16")
    out = subprocess.run([sys.executable, "-c", code2], env=env2, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["77"], (out.stdout, out.stderr)
    assert "unknown key 'nonsense'" in out.stderr and "relabel=7 outside" in out.stderr and "ACM_GATHER_DTYPE" in out.stderr
    from acm_gnn_amd import tuning
    with pytest.raises(ValueError):
        tuning.parse("nonsense=1")
    with pytest.raises(ValueError):
        tuning.parse("relabel=7")
    assert tuning.parse("rows16=5, rewrites=0") == ({"rows16": 5}, {"rewrites": 0})
    with tuning.override(rewrites=0, implicit=0):
        assert (tuning.HOST.rewrites, tuning.HOST.implicit) == (0, 0)
    assert (tuning.HOST.rewrites, tuning.HOST.implicit) == (15, 1)


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_graft_entry_build_hook_exists():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
