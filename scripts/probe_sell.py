#!/usr/bin/env python3
"""Narrow gather over per-wave id streams (scripts/micro/sell_gather.hip) against the CSR kernel of the library, on the
twitch-shaped graph: us per P = A X (32-byte and 16-byte rows) for id-prefetch depth D, rows-in-flight R and the number
of waves the streams are cut for.  Checks every variant against scipy first."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acm_gnn_amd import data as D, functional as AF  # noqa: E402
from acm_gnn_amd.graph import CsrGraph  # noqa: E402

DEV = torch.device("cuda:0")
SENT8, SENT4 = (1 << 27) - 1, (1 << 28) - 1


def build_streams(indptr, indices, n_waves, max_steps=8, pad_steps=16, sentinel=SENT8):
    """Sliced-ELL streams: items (row pieces of <= 32 max_steps neighbours) sorted by length, 4 per slice, slice s
    dealt to wave s % n_waves; returns (stream ids, wave_ptr, wave_step, desc, item_row)."""
    n = indptr.size - 1
    deg = np.diff(indptr).astype(np.int64)
    lmax = 32 * max_steps
    pieces = np.maximum(1, -(-deg // lmax))
    item_row = np.repeat(np.arange(n), pieces)
    first = np.cumsum(pieces) - pieces
    k = np.arange(item_row.size) - first[item_row]
    begin = indptr[item_row] + k * lmax
    end = np.minimum(begin + lmax, indptr[item_row + 1])
    length = end - begin
    order = np.argsort(-length, kind="stable")
    item_row, begin, length = item_row[order], begin[order], length[order]
    n_items = item_row.size
    n_slices = -(-n_items // 4)
    padn = n_slices * 4 - n_items
    lens4 = np.concatenate([length, np.zeros(padn, np.int64)]).reshape(n_slices, 4)
    steps = np.maximum(1, -(-lens4.max(1) // 32))
    wave = np.arange(n_slices) % n_waves
    rnd = np.arange(n_slices) // n_waves
    # per-wave order = round order; position of slice in the wave-major list
    key = np.lexsort((rnd, wave))
    slice_pos = np.empty(n_slices, np.int64)
    slice_pos[key] = np.arange(n_slices)
    steps_sorted = steps[key]
    step_base_sorted = np.cumsum(steps_sorted) - steps_sorted
    step_base = np.empty(n_slices, np.int64)
    step_base[key] = step_base_sorted
    total_steps = int(steps.sum())
    wave_ptr = np.zeros(n_waves + 1, np.int64)
    np.add.at(wave_ptr, wave + 1, 1)
    wave_ptr = np.cumsum(wave_ptr)
    wave_first = np.full(n_waves, total_steps, np.int64)
    has = wave_ptr[1:] > wave_ptr[:-1]
    wave_first[has] = step_base_sorted[wave_ptr[:-1][has]]
    desc = np.full((n_slices, 8), -1, np.int32)
    desc[slice_pos, 0] = steps
    outs = np.concatenate([np.arange(n_items), np.full(padn, -1)]).reshape(n_slices, 4)
    desc[slice_pos, 1:5] = outs
    stream = np.full((total_steps + pad_steps) * 128, sentinel, np.int32)
    # edges
    it = np.repeat(np.arange(n_items), length)
    kk = np.arange(int(length.sum())) - np.repeat(np.cumsum(length) - length, length)
    sl, g = it // 4, it % 4
    pos = (step_base[sl] + kk // 32) * 128 + g * 32 + kk % 32
    stream[pos] = indices[np.repeat(begin, length) + kk]
    return stream, wave_ptr.astype(np.int32), wave_first.astype(np.int32), desc, item_row, total_steps


def main():
    so = "/tmp/sell_gather.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", "-o", so,
                           os.path.join(ROOT, "scripts", "micro", "sell_gather.hip")])
    lib = C.CDLL(so)
    lib.sell_gather.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    adj, x_np, y_np, (tr, _, _), n = D.synthetic_dataset("twitch-gamer")
    order = os.environ.get("ORDER", "degree")
    if order == "degree":
        perm = D.degree_order(adj)
        adj, *_ = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), perm)
    low, deg = D.build_filters(adj)
    low = low.tocsr()
    low.sort_indices()
    nnz = low.nnz
    indptr, indices = low.indptr.astype(np.int64), low.indices.astype(np.int32)
    pat = low.copy()
    pat.data[:] = 1.0
    g = CsrGraph.from_csr(torch.from_numpy(low.indptr.astype(np.int32)).to(DEV), torch.from_numpy(indices).to(DEV), None, n)
    stream_h = torch.cuda.current_stream().cuda_stream
    print(f"n {n}  nnz {nnz}  order {order}")
    for width in (8, 4):
        x = torch.randn(n, width, device=DEV)
        ref = torch.from_numpy(pat @ x.cpu().numpy().astype(np.float64))
        y = torch.empty(n, width, device=DEV)

        def timeit(fn, reps=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        if width == 8 and os.environ.get("HUB4"):
            continue
        if width == 8 and os.environ.get("VARIANTS", "1") == "1":
            lib.sell_gather_v.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
            for n_waves in ((4096, 8192) if os.environ.get("HUB_VARIANTS") else ()):
                stream, wptr, wstep, desc, item_row, total_steps = build_streams(indptr, indices, n_waves, max_steps=8)
                d_stream, d_wptr, d_wstep, d_desc = (torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc))
                out = torch.zeros(item_row.size, width, device=DEV)
                line = [f"  variants, waves {n_waves}:"]
                for aux, hub, wpb in ((0, 0, 4), (1, 0, 4), (2, 0, 4), (3, 0, 4), (16, 0, 4), (17, 0, 4), (0, 0, 16),
                                      (0, 512, 16), (0, 1024, 16), (0, 2048, 16), (0, 4096, 16), (0, 4608, 16),
                                      (0, 1024, 8), (0, 2048, 8)):
                    def run():
                        st = lib.sell_gather_v(aux, hub, wpb, d_stream.data_ptr(), d_wptr.data_ptr(), d_wstep.data_ptr(),
                                               d_desc.data_ptr(), x.data_ptr(), n * width * 4, out.data_ptr(), n_waves, stream_h)
                        assert st == 0, st
                    out.zero_()
                    run()
                    torch.cuda.synchronize()
                    got = np.zeros((n, width))
                    np.add.at(got, item_row, out.cpu().double().numpy())
                    e = float(np.abs(got - ref.numpy()).max())
                    assert e < 1e-3, (aux, hub, wpb, e)
                    line.append(f"aux{aux}/hub{hub}/wpb{wpb} {timeit(run):6.1f}")
                print(" ".join(line), flush=True)
            lib.sell_gather_id.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint] + [C.c_void_p] * 4 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
            for n_waves in (4096, 8192):
                stream, wptr, wstep, desc, item_row, total_steps = build_streams(indptr, indices, n_waves, max_steps=8)
                d_stream, d_wptr, d_wstep, d_desc = (torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc))
                out = torch.zeros(item_row.size, width, device=DEV)
                line = [f"  id policy (0 plain, 1 nt, 2 sc1, 3 nt+sc1, 4 sc0+sc1), waves {n_waves}:"]
                for depth, idp in ((2, 0), (2, 1), (2, 2), (2, 3), (2, 4), (6, 0), (6, 1), (6, 2), (6, 3), (6, 4), (12, 1), (12, 2)):
                    def run():
                        st = lib.sell_gather_id(depth, idp, d_stream.data_ptr(), stream.size * 4, d_wptr.data_ptr(),
                                                d_wstep.data_ptr(), d_desc.data_ptr(), x.data_ptr(), n * width * 4,
                                                out.data_ptr(), n_waves, stream_h)
                        assert st == 0, st
                    out.zero_()
                    run()
                    torch.cuda.synchronize()
                    got = np.zeros((n, width))
                    np.add.at(got, item_row, out.cpu().double().numpy())
                    e = float(np.abs(got - ref.numpy()).max())
                    assert e < 1e-3, (depth, idp, e)
                    line.append(f"D{depth}/p{idp} {timeit(run):6.1f}")
                print(" ".join(line), flush=True)
            if os.environ.get("ONLY_VARIANTS"):
                return
        if width == 4 and os.environ.get("HUB4"):
            lib.sell_gather_q.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
            for n_waves in (4096, 8192):
                stream, wptr, wstep, desc, item_row, total_steps = build_streams(indptr, indices, n_waves, max_steps=8, sentinel=SENT4)
                d_stream, d_wptr, d_wstep, d_desc = (torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc))
                out = torch.zeros(item_row.size, width, device=DEV)
                line = [f"  16-byte rows, hub rows in LDS, waves {n_waves}:"]
                for hub, wpb in ((0, 4), (0, 16), (2048, 16), (4096, 16), (8190, 16), (2048, 8), (4095, 8)):
                    def run():
                        st = lib.sell_gather_q(hub, wpb, d_stream.data_ptr(), d_wptr.data_ptr(), d_wstep.data_ptr(),
                                               d_desc.data_ptr(), x.data_ptr(), n * width * 4, out.data_ptr(), n_waves, stream_h)
                        assert st == 0, st
                    out.zero_()
                    run()
                    torch.cuda.synchronize()
                    got = np.zeros((n, width))
                    np.add.at(got, item_row, out.cpu().double().numpy())
                    e = float(np.abs(got - ref.numpy()).max())
                    assert e < 1e-3, (hub, wpb, e)
                    line.append(f"hub{hub}/wpb{wpb} {timeit(run):6.1f}")
                print(" ".join(line), flush=True)
            return
        t_csr = timeit(lambda: AF.spmm(g, x, out=y))
        err = float((y.cpu().double() - ref).abs().max())
        print(f"width {width}: CSR kernel {t_csr:7.1f} us  (err {err:.1e})")
        for max_steps in (8, 4):
            for n_waves in (4096, 5120, 8192, 16384):
                stream, wptr, wstep, desc, item_row, total_steps = build_streams(
                    indptr, indices, n_waves, max_steps=max_steps, sentinel=SENT8 if width == 8 else SENT4)
                per_wave = np.zeros(n_waves, np.int64)
                sl_wave = np.repeat(np.arange(n_waves), np.diff(wptr))
                np.add.at(per_wave, sl_wave, desc[:, 0])
                d_stream, d_wptr, d_wstep, d_desc = (torch.from_numpy(a).to(DEV) for a in (stream, wptr, wstep, desc))
                out = torch.zeros(item_row.size, width, device=DEV)
                line = [f"  max_steps {max_steps} waves {n_waves:5d} slots/nnz {total_steps * 128 / nnz:.3f} "
                        f"steps/wave {per_wave.mean():.1f}..{per_wave.max()}:"]
                for depth, rows in ((1, 1), (2, 1), (4, 1), (2, 2), (4, 2), (6, 2), (8, 2), (4, 3)):
                    def run():
                        st = lib.sell_gather(width, depth, rows, d_stream.data_ptr(), d_wptr.data_ptr(), d_wstep.data_ptr(),
                                             d_desc.data_ptr(), x.data_ptr(), n * width * 4, out.data_ptr(), n_waves, stream_h)
                        assert st == 0, st
                    run()
                    torch.cuda.synchronize()
                    got = np.zeros((n, width))
                    np.add.at(got, item_row, out.cpu().double().numpy())
                    e = float(np.abs(got - ref.numpy()).max())
                    assert e < 1e-3, (depth, rows, e)
                    line.append(f"D{depth}R{rows} {timeit(run):6.1f}")
                print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
