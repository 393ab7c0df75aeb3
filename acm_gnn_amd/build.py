"""Build libacm_hip.so in-tree with hipcc for gfx950 (MI355X).

The library is plain C ABI (include/acm_hip.h); no torch headers are involved,
so this is an ordinary ``hipcc -shared`` of three translation units.  The build
is skipped when the library is newer than every source.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libacm_hip.so")
SOURCES = ["acm_csr.cpp", "acm_gemm.hip", "acm_conv.hip", "acm_conv_agg.hip", "acm_loss.hip", "acm_optim.hip", "acm_dropout.hip", "acm_proj.hip", "acm_reduce.hip", "acm_linear.hip", "acm_conv_acmii.hip", "acm_conv_agg16.hip", "acm_gemm_rows.hip", "acm_gemm_bx3.hip", "acm_conv_local16.hip", "acm_conv_acmii_v.hip", "acm_small.hip", "acm_conv_aggw.hip"]
ARCH = "gfx950"
# Per-source compiler switches.  acm_conv_acmii.hip reads every MFMA result on the VALU right away (ReLU + sum per edge):
# with the results in AGPRs hipcc copies each of them through v_accvgpr_read (32 extra VALU instructions per 16 MFMAs and
# 28 more registers); the VGPR form of the MFMA writes them where the VALU can use them (150 -> 95 instructions per batch,
# 152 -> 96 registers).
EXTRA_FLAGS = {"acm_conv_acmii.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "acm_conv_agg16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "acm_conv_acmii_v.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _deps():
    out = [os.path.join(CSRC, s) for s in SOURCES]
    out += [os.path.join(CSRC, "acm_common.h"), os.path.join(CSRC, "acm_conv_device.h"), os.path.join(CSRC, "acm_stream_device.h"), os.path.join(CSRC, "acm_rows16_device.h"), os.path.join(CSRC, "acm_adam_device.h"), os.path.join(CSRC, "acm_bx3_device.h"), os.path.join(CSRC, "acm_reduce_device.h"),
            os.path.join(INCLUDE, "acm_hip.h")]
    return out


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libacm_hip.so. Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
             "-Wall", "-Wno-unused-function"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout.decode(errors="replace"))
    os.replace(tmp, LIB_PATH)
    _build_agpr_variant(hipcc, flags, objs, verbose)
    return LIB_PATH


# The same library with acm_conv_acmii.hip compiled WITHOUT -amdgpu-mfma-vgpr-form (MFMA results in AGPRs, copied out by
# v_accvgpr_read): tests/test_gpu_mfma_forms.py requires bit-identical conv_acmii_fwd output from both builds -- the
# arithmetic is the same, so any difference is a register hazard (round 3 saw one with inline assembly reading MFMA
# results; the reads are builtins now).  Test infrastructure: nothing loads it unless ACM_HIP_LIBRARY names it.
VARIANT_PATH = os.path.join(LIB_DIR, "libacm_hip_acmii_agpr.so")


def _build_agpr_variant(hipcc, flags, objs, verbose=False):
    src = "acm_conv_acmii.hip"
    obj = os.path.join(LIB_DIR, "acm_conv_acmii.agpr.o")
    cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src} (AGPR variant):\n{res.stdout.decode(errors='replace')}")
    std = os.path.join(LIB_DIR, "acm_conv_acmii.o")
    tmp = VARIANT_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + [obj if o == std else o for o in objs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("link failed (AGPR variant):\n" + res.stdout.decode(errors="replace"))
    os.replace(tmp, VARIANT_PATH)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
