"""The ONE dispatch mechanism (include/acm_hip.h: acm_tuning_t + the host-side record below).

Where two execution forms compute the same result (up to fp32 re-association), a *tuning switch* says which one runs.
All switches live in two small records, both filled ONCE from the environment variable

    ACM_TUNING="key=value,key=value,..."

-- the library's ``acm_tuning_t`` when ``libacm_hip.so`` is loaded (kernel-level forms; read and written here through
``acm_tuning_get`` / ``acm_tuning_set``), and the host record ``HOST`` when this module is imported (which algebraic
rewrite / operator form / loop structure the Python host layer picks).  Nothing else in the package reads the process
environment for dispatch, and no ``acm_*`` launch path calls ``getenv``.

Kernel-level keys (acm_tuning_t; see the header for the values):
    chunk, wide_form, bwd_split, rows16, agg_fused, gemm_forms
Host-level keys:
    rewrites   bit mask of the algebraic rewrites of a first layer: 1 = aggregate-first ``A (X W) = (A X) W`` (ACM,
               acmsgc), 2 = ACMII recompute-on-gather, 4 = with 2, the mask form of it on the bf16 matrix pipe (forward and
               weight gradients from V = masks^T inputs: acm_conv_acmii_v.hip), 8 = with 1, for wide dense inputs
               (16 < F_in <= 128) the projections and the head behind the gather as ONE kernel (acm_conv_aggw_fwd) instead of
               two products + acm_conv_head_fwd.  Default 15; 0 = the literal form (project, then gather 2F floats per
               edge) -- ``bench.py``'s ``literal_ms_per_step`` and the parity tests compare them
    implicit   1 = pattern-only operators where the filter allows (one 4-byte id stream, row scales in the epilogues);
               0 = always the explicit (id, value) form with an explicit transposed CSR.  Default 1
    relabel    in-operator degree relabelling: -1 = graphs of >= 32 768 nodes (default), 0 = never, 1 = always
    pipeline   the training loop's input pipeline (next step's first-layer gather inside this step's backward):
               0 = off, n > 0 = on for graphs of at least n rows.  Default 8192
    csr_features  wide, mostly-zero feature matrices handed over DENSE (bag-of-words, one-hot: the reference's loaders
               densify them) are projected from a CSR copy made once per tensor (graph.SparseFeatures.auto): 0 = never,
               n > 0 = for inputs of at least n columns with at most 1/16 of the entries nonzero.  Default 256
    small_step the fused six-launch training step / three-launch evaluation pass for small graphs (small.SmallPlan,
               acm_small_step): 0 = never, n > 0 = for graphs of at most n rows (the kernels take <= 16384).  Default 16384

One policy for a malformed ACM_TUNING in both readers: the item is REPORTED on stderr and IGNORED -- by the
library when it is loaded, by this module when it is imported (``parse(text)`` itself stays strict for programmatic use:
it raises ValueError).  Switches of earlier rounds that still sit in the environment (ACM_GATHER_DTYPE, ...) are reported once.

Tests and probes flip forms with ``override(...)`` (a context manager; restores both records on exit) -- or, for child
processes, by putting ACM_TUNING into the child's environment.
"""
import contextlib
import ctypes as C
import os
import threading

KERNEL_KEYS = ("chunk", "wide_form", "bwd_split", "rows16", "agg_fused", "gemm_forms")
HOST_DEFAULTS = {"rewrites": 15, "implicit": 1, "relabel": -1, "pipeline": 8192, "csr_features": 256, "small_step": 16384}
HOST_RANGES = {"rewrites": (0, 15), "implicit": (0, 1), "relabel": (-1, 1), "pipeline": (0, 1 << 31), "csr_features": (0, 1 << 31),
               "small_step": (0, 16384)}

REWRITE_AGG_FIRST = 1
REWRITE_ACMII_RECOMPUTE = 2
REWRITE_ACMII_MASK = 4
REWRITE_AGGW_FUSED = 8
ROWS16_EPI, ROWS16_BWD, ROWS16_LOCAL = 1, 2, 4
GEMM_ROWS, GEMM_BX3, GEMM_BX3_WIDE, GEMM_ROWS_ALWAYS = 1, 2, 4, 8


class Tuning(C.Structure):
    """ctypes image of acm_tuning_t."""
    _fields_ = [(k, C.c_int32) for k in KERNEL_KEYS] + [("reserved", C.c_int32 * 10)]


class _Host:
    __slots__ = tuple(HOST_DEFAULTS)

    def __init__(self, **kw):
        for k, v in HOST_DEFAULTS.items():
            setattr(self, k, int(kw.get(k, v)))

    def as_dict(self):
        return {k: getattr(self, k) for k in HOST_DEFAULTS}


def parse(text):
    """'key=value,...' -> (kernel dict, host dict); unknown keys and out-of-range host values raise ValueError."""
    kern, host = {}, {}
    for item in (text or "").split(","):
        item = item.strip()
        if not item:
            continue
        key, sep, val = item.partition("=")
        if not sep:
            raise ValueError(f"ACM_TUNING: '{item}' is not key=value")
        try:
            v = int(val, 0)
        except ValueError:
            raise ValueError(f"ACM_TUNING: '{item}': the value must be an integer") from None
        if key in KERNEL_KEYS:
            kern[key] = v
        elif key in HOST_DEFAULTS:
            lo, hi = HOST_RANGES[key]
            if not lo <= v <= hi:
                raise ValueError(f"ACM_TUNING: {key}={v} outside {lo}..{hi}")
            host[key] = v
        else:
            raise ValueError(f"ACM_TUNING: unknown key '{key}' (known: {', '.join(KERNEL_KEYS + tuple(HOST_DEFAULTS))})")
    return kern, host


def parse_lenient(text, report=None):
    """``parse`` item by item: a malformed / unknown / out-of-range item is reported (``report(message)``, default: one
    line on stderr) and ignored -- what the library does with the same variable when it is loaded."""
    import sys
    report = report or (lambda msg: sys.stderr.write(f"acm_gnn_amd: {msg} (ignored)\n"))
    kern, host = {}, {}
    for item in (text or "").split(","):
        if not item.strip():
            continue
        try:
            k, h = parse(item)
        except ValueError as exc:
            report(str(exc))
            continue
        kern.update(k)
        host.update(h)
    return kern, host


# switches of rounds 1-3 that no launch path reads any more: say so once instead of ignoring them silently
LEGACY_ENV = ("ACM_GATHER_DTYPE", "ACM_EVAL_AGG_CACHE", "ACM_CHUNK", "ACM_WIDE_FORM", "ACM_BWD_SPLIT", "ACM_AGG_FUSED", "ACM_PIPELINE",
              "ACM_IMPLICIT", "ACM_RELABEL", "ACM_REWRITES", "ACM_ROWS16", "ACM_GEMM_FORMS")


def _report_legacy(environ, report=None):
    import sys
    report = report or (lambda msg: sys.stderr.write(f"acm_gnn_amd: {msg}\n"))
    found = [k for k in LEGACY_ENV if k in environ]
    if found:
        report(f"{', '.join(found)}: environment switches of earlier rounds are no longer read; use ACM_TUNING=\"key=value,...\"")
    return found


_LOADED_HOST = parse_lenient(os.environ.get("ACM_TUNING", ""))[1]           # read ONCE, at import
_report_legacy(os.environ)
HOST = _Host(**_LOADED_HOST)
_lock = threading.RLock()
_kernel_cache = None
_kernel_touched = False


def kernel():
    """The library's current acm_tuning_t as a dict (cached; ``set_kernel`` / ``override`` refresh it)."""
    global _kernel_cache
    if _kernel_cache is None:
        from . import _lib
        t = Tuning()
        _lib.check(_lib.load().acm_tuning_get(C.byref(t)), "acm_tuning_get")
        _kernel_cache = {k: int(getattr(t, k)) for k in KERNEL_KEYS}
    return _kernel_cache


def set_kernel(**kw):
    """Change fields of the library's record (acm_tuning_set validates them)."""
    global _kernel_cache, _kernel_touched
    from . import _lib
    with _lock:
        _kernel_touched = True
        cur = dict(kernel())
        for k, v in kw.items():
            if k not in KERNEL_KEYS:
                raise KeyError(k)
            cur[k] = int(v)
        t = Tuning()
        for k, v in cur.items():
            setattr(t, k, v)
        _lib.check(_lib.load().acm_tuning_set(C.byref(t)), "acm_tuning_set")
        _kernel_cache = None


def apply(**kw):
    """Set host and / or kernel switches until further notice (``override`` is the scoped form)."""
    host_kw = {k: v for k, v in kw.items() if k in HOST_DEFAULTS}
    kern_kw = {k: v for k, v in kw.items() if k in KERNEL_KEYS}
    unknown = set(kw) - set(host_kw) - set(kern_kw)
    if unknown:
        raise KeyError(f"unknown tuning switch(es): {sorted(unknown)}")
    for k, v in host_kw.items():
        lo, hi = HOST_RANGES[k]
        if not lo <= int(v) <= hi:
            raise ValueError(f"{k}={v} outside {lo}..{hi}")
    with _lock:
        for k, v in host_kw.items():
            setattr(HOST, k, int(v))
        if kern_kw:
            set_kernel(**kern_kw)


def reset(kernel_too=True):
    """Back to the load-time records (defaults + ACM_TUNING); the library's only if this process ever changed it."""
    global _kernel_cache, _kernel_touched
    with _lock:
        for k, v in HOST_DEFAULTS.items():
            setattr(HOST, k, int(_LOADED_HOST.get(k, v)))
        if kernel_too and _kernel_touched:
            from . import _lib
            _lib.check(_lib.load().acm_tuning_set(None), "acm_tuning_set")
            _kernel_touched = False
        _kernel_cache = None


def invalidate():
    """Forget the cached copy of the library's record (a test that swaps the library calls this)."""
    global _kernel_cache
    _kernel_cache = None


def gemm_forms():
    return kernel()["gemm_forms"]


@contextlib.contextmanager
def override(**kw):
    """``with tuning.override(rewrites=0, rows16=0): ...`` -- host and kernel switches for the duration of the block
    (process-wide, like the records themselves: not for concurrent use from several threads)."""
    host_kw = {k: v for k, v in kw.items() if k in HOST_DEFAULTS}
    kern_kw = {k: v for k, v in kw.items() if k in KERNEL_KEYS}
    unknown = set(kw) - set(host_kw) - set(kern_kw)
    if unknown:
        raise KeyError(f"unknown tuning switch(es): {sorted(unknown)}")
    for k, v in host_kw.items():
        lo, hi = HOST_RANGES[k]
        if not lo <= int(v) <= hi:
            raise ValueError(f"{k}={v} outside {lo}..{hi}")
    with _lock:
        old_host = HOST.as_dict()
        old_kern = dict(kernel()) if kern_kw else None
        try:
            for k, v in host_kw.items():
                setattr(HOST, k, int(v))
            if kern_kw:
                set_kernel(**kern_kw)
            yield
        finally:
            for k, v in old_host.items():
                setattr(HOST, k, v)
            if old_kern is not None:
                set_kernel(**old_kern)


def env_for_child(**kw):
    """The ACM_TUNING string that gives a child process the current switches plus ``kw``."""
    cur = dict(HOST.as_dict())
    try:
        cur.update(kernel())
    except Exception:            # library not loadable here: host keys only
        pass
    cur.update({k: int(v) for k, v in kw.items()})
    return ",".join(f"{k}={v}" for k, v in cur.items())
