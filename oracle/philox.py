"""TEST INFRASTRUCTURE (like everything under oracle/): a numpy restatement of the counter-based dropout mask the HIP
kernels draw (acm_gnn_amd/csrc/acm_common.h: acm_philox7 / acm_drop1 / acm_drop4; include/acm_hip.h: acm_dropout_t).

The reference draws its masks with torch's generator (F.dropout in ACM-Geometric/models.py:54,70); the library's masks
are a pure function of (seed, step, tag, row, column) so that forward and backward kernels can regenerate them.  The
checker regenerates the same masks here and replays them into the oracle's forward (oracle.gcn_forward(masks=...)).
Known-answer vectors of Philox4x32 (Random123) pin the generator in tests/test_dropout_cpu.py.

Imported only by tests/, bench.py's post-timing check and __graft_entry__.smoke()."""
import numpy as np


def philox7_words(seed, step, tag, rows, blocks):
    """Philox4x32-7 words of the dropout mask: counter (row, block | tag << 16, step lo, step hi), key = seed.
    rows / blocks are integer arrays of equal shape; returns uint32 [4, ...]."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    x0 = np.asarray(rows).astype(np.uint64) & mask
    x1 = (np.asarray(blocks).astype(np.uint64) | np.uint64((int(tag) << 16) & 0xFFFFFFFF)) & mask
    x2 = np.full(x0.shape, int(step) & 0xFFFFFFFF, np.uint64)
    x3 = np.full(x0.shape, (int(step) >> 32) & 0xFFFFFFFF, np.uint64)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(7):
        p0, p1 = np.uint64(M0) * x0, np.uint64(M1) * x2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        x0, x1, x2, x3 = hi1 ^ x1 ^ np.uint64(k0), lo1, hi0 ^ x3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([x0, x1, x2, x3]).astype(np.uint32)


def dropout_factors(seed, step, tag, p, n_rows, n_cols, row_offset=0, rows=None):
    """[n_rows, n_cols] float64 factors (0 or 1 / (1 - p), the latter rounded like the kernels' fp32 constant) of the
    mask with probability ``p`` at counter value ``step``: element (row, col) uses word (col >> 4) & 3 of the Philox call
    with block (col & 15) + 16 (col >> 6).  ``rows``: an explicit array of row numbers instead of 0 .. n_rows - 1."""
    if p <= 0:
        return np.ones((n_rows if rows is None else len(rows), n_cols))
    rr = (np.arange(n_rows) if rows is None else np.asarray(rows)) + int(row_offset)
    r, c = np.meshgrid(rr, np.arange(n_cols), indexing="ij")
    w = philox7_words(seed, step, tag, r, (c & 15) + 16 * (c >> 6))
    word = np.take_along_axis(w, ((c >> 4) & 3)[None], 0)[0]
    t = float(np.float32(p)) * 4294967296.0
    thresh = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(word >= thresh, np.float64(inv), 0.0)
