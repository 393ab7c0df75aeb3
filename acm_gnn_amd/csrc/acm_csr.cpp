// Graph handle: CSR row block + nnz-balanced work list, transpose, row slicing.
// One-off preprocessing (host side); the hot path never comes through here.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <new>
#include <queue>
#include <string>
#include <utility>
#include <vector>

#include "acm_common.h"

static thread_local char g_err[512] = "";

void acm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int acm_version(void) { return ACM_ABI_VERSION; }
extern "C" const char* acm_last_error(void) { return g_err; }

// ---------------------------------------------------------------- tuning record
namespace {
const acm_tuning_t kTuningDefaults = {0, 0, -1, 7, 1, 7, {0}};
struct TuningField { const char* name; int32_t acm_tuning_t::*field; };
const TuningField kTuningFields[] = {{"chunk", &acm_tuning_t::chunk},           {"wide_form", &acm_tuning_t::wide_form},
                                     {"bwd_split", &acm_tuning_t::bwd_split},   {"rows16", &acm_tuning_t::rows16},
                                     {"agg_fused", &acm_tuning_t::agg_fused},   {"gemm_forms", &acm_tuning_t::gemm_forms}};
// host-side keys of the same variable (acm_gnn_amd/tuning.py reads them; listed here so that they are not "unknown")
const char* const kHostKeys[] = {"rewrites", "implicit", "relabel", "pipeline", "csr_features", "small_step"};

const char* tuning_invalid(const acm_tuning_t& t) {
    if (t.chunk != 0 && (t.chunk < 8 || t.chunk > 4096 || (t.chunk & (t.chunk - 1)))) return "chunk";
    if (t.wide_form < 0 || t.wide_form > 3) return "wide_form";
    if (t.bwd_split < -1 || t.bwd_split > 1) return "bwd_split";
    if (t.rows16 < 0 || t.rows16 > 7) return "rows16";
    if (t.agg_fused < 0 || t.agg_fused > 1) return "agg_fused";
    if (t.gemm_forms < 0 || t.gemm_forms > 15) return "gemm_forms";
    return nullptr;
}

// defaults + ACM_TUNING; a malformed variable is reported on stderr (once, at load) and the offending item ignored
acm_tuning_t tuning_from_environment() {
    acm_tuning_t t = kTuningDefaults;
    const char* env = getenv("ACM_TUNING");
    if (!env) return t;
    std::string text(env);
    size_t pos = 0;
    while (pos < text.size()) {
        size_t end = text.find(',', pos);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        const std::string key = item.substr(0, eq), val = eq == std::string::npos ? "" : item.substr(eq + 1);
        bool known = false;
        for (const char* h : kHostKeys) known = known || key == h;
        for (const TuningField& f : kTuningFields) {
            if (key != f.name) continue;
            known = true;
            char* stop = nullptr;
            const long v = strtol(val.c_str(), &stop, 0);
            acm_tuning_t probe = t;
            probe.*(f.field) = (int32_t)v;
            if (val.empty() || *stop || tuning_invalid(probe)) fprintf(stderr, "libacm_hip: ACM_TUNING: bad value '%s' ignored\n", item.c_str());
            else t = probe;
        }
        if (!known) fprintf(stderr, "libacm_hip: ACM_TUNING: unknown key '%s' ignored\n", key.c_str());
    }
    return t;
}
const acm_tuning_t g_tuning_loaded = tuning_from_environment();      // read ONCE, when the library is loaded
acm_tuning_t g_tuning = g_tuning_loaded;
}  // namespace

const acm_tuning_t& acm_tuning() { return g_tuning; }

extern "C" int acm_tuning_get(acm_tuning_t* out) {
    ACM_REQUIRE(out, ACM_EINVAL, "acm_tuning_get: NULL argument");
    *out = g_tuning;
    return ACM_OK;
}

extern "C" int acm_tuning_set(const acm_tuning_t* in) {
    if (!in) {
        g_tuning = g_tuning_loaded;
        return ACM_OK;
    }
    const char* bad = tuning_invalid(*in);
    ACM_REQUIRE(!bad, ACM_EINVAL, "acm_tuning_set: field '%s' out of range", bad);
    g_tuning = *in;
    for (int32_t& r : g_tuning.reserved) r = 0;
    return ACM_OK;
}

namespace {

void free_streams(AcmStreams* t) {
    if (!t) return;
    for (void* q : {(void*)t->ids, (void*)t->waves, (void*)t->items, (void*)t->long_rows,
                    (void*)t->long_index, (void*)t->counters, (void*)t->slots})
        if (q) (void)hipFree(q);
    delete t;
}

void free_item_streams(AcmItemStreams* t) {
    if (!t) return;
    for (void* q : {(void*)t->ids, (void*)t->quads, (void*)t->waves})
        if (q) (void)hipFree(q);
    delete t;
}

void free_handle(acm_csr* a) {
    if (!a) return;
    if (a->indptr) (void)hipFree(a->indptr);
    if (a->indices) (void)hipFree(a->indices);
    if (a->vals) (void)hipFree(a->vals);
    if (a->long_index) (void)hipFree(a->long_index);
    if (a->src_pos) (void)hipFree(a->src_pos);
    if (a->items) (void)hipFree(a->items);
    if (a->long_rows) (void)hipFree(a->long_rows);
    free_streams(a->streams);
    free_item_streams(a->item_streams);
    delete a;
}

// Split rows into work items of at most `chunk` neighbours -- host side, from a host copy of indptr.
// A long row (more than `chunk` neighbours) becomes P = min(16, ceil(deg / chunk)) near-equal pieces with consecutive
// partial slots.  All pieces come FIRST in the list, packed into windows of ACM_WINDOW = 16 items so that the pieces of
// one row never straddle a window (a window that cannot take the next row is filled up with empty pieces of the row
// before, and so is the last one): a kernel whose workgroup takes one window per round (16 groups of 16 lanes) can
// combine the pieces of a row through LDS and finish the row itself instead of leaving partial sums to a second
// launch; every other kernel writes the pieces to their partial slots as before.  The whole rows follow in row order.
// Returns in n_multi the rows that take several windows.
void build_items(const std::vector<int32_t>& indptr, int chunk, std::vector<AcmItem>& items,
                 std::vector<AcmLongRow>& longs, int64_t& n_slots, int32_t& max_deg, int64_t& n_windows, int64_t& n_multi) {
    const int64_t n = (int64_t)indptr.size() - 1;
    items.clear();
    longs.clear();
    items.reserve(n + n / 8);
    n_slots = 0;
    max_deg = 0;
    n_multi = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int32_t b = indptr[r], e = indptr[r + 1];
        const int32_t deg = e - b;
        max_deg = std::max(max_deg, deg);
        if (deg <= chunk) continue;
        int32_t pieces = (deg + chunk - 1) / chunk;
        // A row that sixteen pieces of up to 4 x chunk neighbours do not cover takes SEVERAL whole windows (at most sixteen)
        // of pieces of about 2 x chunk: sixteen pieces in one window all run on the CU that window's workgroup lands on,
        // and a row of 21 k neighbours then keeps one texture path busy for ~20 us whatever the rest of the chip does (the
        // floor of a rank's gathers in an 8-rank plan: DESIGN.md section 7).  With the chunk sized to the operator
        // (nnz / 32768 < chunk <= nnz / 16384) the bar of 64 x chunk neighbours is 0.5 .. 1 x the mean CU's share of the
        // whole operator: no row of the twitch- or arXiv-year-shaped graphs reaches it on one GPU, the top rows of a rank
        // of an 8-rank plan do.  (A lower bar -- every row above 32 x chunk that is also above nnz / 128 -- left the rows of
        // 8-13 k neighbours in one window each and was slower per rank: 27-29 us against 22-27 us for the output-layer
        // backward gather.)  A window of such a row leaves its sum in its first piece's slot and a small second launch adds
        // the windows (AcmLongRow.windows > 1).
        int32_t windows = 0;
        if (pieces > ACM_WINDOW && (deg + ACM_WINDOW - 1) / ACM_WINDOW > 4 * chunk) {
            windows = (deg + 2 * ACM_WINDOW * chunk - 1) / (2 * ACM_WINDOW * chunk);
            if (windows > ACM_WINDOW) windows = ACM_WINDOW;
            pieces = windows * ACM_WINDOW;
        } else if (pieces > ACM_WINDOW) {
            pieces = ACM_WINDOW;
        }
        const int32_t room = ACM_WINDOW - (int32_t)(items.size() % ACM_WINDOW);
        if ((room < pieces || windows) && room < ACM_WINDOW) {    // does not fit: pad with empty pieces of the row before
            AcmLongRow& prev = longs.back();
            for (int32_t q = 0; q < room; ++q) items.push_back({prev.row, indptr[prev.row + 1], indptr[prev.row + 1], (int32_t)n_slots++});
            prev.slot_end = (int32_t)n_slots;
        }
        const int32_t per = (deg + pieces - 1) / pieces;
        AcmLongRow lr = {(int32_t)r, (int32_t)n_slots, 0, windows};
        for (int32_t q = 0; q < pieces; ++q) {
            const int32_t s = std::min(e, b + q * per);
            items.push_back({(int32_t)r, s, std::min(e, s + per), (int32_t)n_slots++});
        }
        lr.slot_end = (int32_t)n_slots;
        n_multi += windows > 1;
        longs.push_back(lr);
    }
    if (!longs.empty() && items.size() % ACM_WINDOW) {
        AcmLongRow& prev = longs.back();
        while (items.size() % ACM_WINDOW) items.push_back({prev.row, indptr[prev.row + 1], indptr[prev.row + 1], (int32_t)n_slots++});
        prev.slot_end = (int32_t)n_slots;
    }
    n_windows = (int64_t)items.size() / ACM_WINDOW;
    for (int64_t r = 0; r < n; ++r)
        if (indptr[r + 1] - indptr[r] <= chunk) items.push_back({(int32_t)r, indptr[r], indptr[r + 1], -1});
}

// Finish a handle whose indptr/indices/vals device arrays are already in place.
int finish_handle(acm_csr* a, const std::vector<int32_t>& h_indptr, int chunk) {
    // chunk <= 0: size the chunks to the work one 16-lane group gets when the chip is full (256 CUs x 8 waves
    // x 4 groups): long chunks mean fewer partial sums and a shorter fix-up pass, but one item must not
    // outlast the average group's whole share.  Measured optimum per graph (profiles/r01_o_chunk_sweep.txt):
    // 128 for squirrel/chameleon (<=0.4 M nnz), 256 for penn94/arxiv-year (2.5 M), 1024 for twitch-gamer (13.7 M).
    const int env_chunk = acm_tuning().chunk;
    int auto_chunk = ACM_MIN_CHUNK;
    while (auto_chunk < ACM_MAX_CHUNK && 2 * (int64_t)auto_chunk <= a->nnz / ACM_GROUPS_IN_FLIGHT) auto_chunk *= 2;
    a->chunk = chunk > 0 ? chunk : (env_chunk > 0 ? env_chunk : auto_chunk);
    std::vector<AcmItem> items;
    std::vector<AcmLongRow> longs;
    build_items(h_indptr, a->chunk, items, longs, a->n_slots, a->max_degree, a->n_windows, a->n_multi);
    a->n_items = (int64_t)items.size();
    a->n_long = (int64_t)longs.size();
    if (a->n_items) {
        ACM_CHECK_HIP(hipMalloc((void**)&a->items, items.size() * sizeof(AcmItem)));
        ACM_CHECK_HIP(hipMemcpy(a->items, items.data(), items.size() * sizeof(AcmItem),
                                hipMemcpyHostToDevice));
    }
    if (a->n_long) {
        ACM_CHECK_HIP(hipMalloc((void**)&a->long_rows, longs.size() * sizeof(AcmLongRow)));
        ACM_CHECK_HIP(hipMemcpy(a->long_rows, longs.data(), longs.size() * sizeof(AcmLongRow),
                                hipMemcpyHostToDevice));
        // row -> long-row record, for consumers that fold the fix-up of a long row into their own pass
        std::vector<int32_t> index((size_t)a->n_rows, -1);
        for (size_t i = 0; i < longs.size(); ++i) index[(size_t)longs[i].row] = (int32_t)i;
        ACM_CHECK_HIP(hipMalloc((void**)&a->long_index, index.size() * sizeof(int32_t)));
        ACM_CHECK_HIP(hipMemcpy(a->long_index, index.data(), index.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return ACM_OK;
}

acm_csr* new_handle(int64_t n_rows, int64_t n_cols, int64_t nnz) {
    acm_csr* a = new (std::nothrow) acm_csr();
    if (!a) return nullptr;
    memset(a, 0, sizeof(*a));
    a->n_rows = n_rows;
    a->n_cols = n_cols;
    a->nnz = nnz;
    (void)hipGetDevice(&a->device);
    return a;
}

int alloc_arrays(acm_csr* a, bool with_vals = true) {
    ACM_CHECK_HIP(hipMalloc((void**)&a->indptr, (size_t)(a->n_rows + 1) * sizeof(int32_t)));
    // keep the arrays non-null even for an empty graph so kernels may take their address
    const size_t m = (size_t)std::max<int64_t>(a->nnz, 1);
    ACM_CHECK_HIP(hipMalloc((void**)&a->indices, m * sizeof(int32_t)));
    if (with_vals) ACM_CHECK_HIP(hipMalloc((void**)&a->vals, m * sizeof(float)));   // NULL = pattern-only (implicit ones)
    return ACM_OK;
}

}  // namespace

extern "C" int acm_csr_create(int64_t n_rows, int64_t n_cols, int64_t nnz,
                              const int32_t* indptr_dev, const int32_t* indices_dev,
                              const float* vals_dev, int chunk, acm_csr_t** out) {
    ACM_REQUIRE(out, ACM_EINVAL, "acm_csr_create: out is NULL");
    *out = nullptr;
    ACM_REQUIRE(indptr_dev, ACM_EINVAL, "acm_csr_create: indptr is NULL");
    ACM_REQUIRE(nnz == 0 || indices_dev, ACM_EINVAL, "acm_csr_create: indices NULL with nnz > 0");
    ACM_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, ACM_ESHAPE, "acm_csr_create: negative size");
    ACM_REQUIRE(n_rows < INT32_MAX && n_cols < INT32_MAX && nnz < INT32_MAX, ACM_EUNSUPPORTED,
                "acm_csr_create: sizes must fit int32 (rows %lld cols %lld nnz %lld)",
                (long long)n_rows, (long long)n_cols, (long long)nnz);
    ACM_CHECK_HIP(hipDeviceSynchronize());   // the caller's producer kernels must have finished
    std::vector<int32_t> h_indptr((size_t)n_rows + 1);
    ACM_CHECK_HIP(hipMemcpy(h_indptr.data(), indptr_dev, h_indptr.size() * sizeof(int32_t),
                            hipMemcpyDeviceToHost));
    ACM_REQUIRE(h_indptr[0] == 0 && h_indptr[n_rows] == nnz, ACM_ESHAPE,
                "acm_csr_create: indptr[0]=%d indptr[n]=%d but nnz=%lld", h_indptr[0],
                h_indptr[n_rows], (long long)nnz);
    for (int64_t r = 0; r < n_rows; ++r)
        ACM_REQUIRE(h_indptr[r] <= h_indptr[r + 1], ACM_ESHAPE,
                    "acm_csr_create: indptr not monotone at row %lld", (long long)r);
    if (nnz) {
        std::vector<int32_t> h_idx((size_t)nnz);
        ACM_CHECK_HIP(hipMemcpy(h_idx.data(), indices_dev, (size_t)nnz * sizeof(int32_t),
                                hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < nnz; ++k)
            ACM_REQUIRE(h_idx[k] >= 0 && h_idx[k] < n_cols, ACM_ESHAPE,
                        "acm_csr_create: column id %d out of range at position %lld", h_idx[k],
                        (long long)k);
    }
    acm_csr* a = new_handle(n_rows, n_cols, nnz);
    ACM_REQUIRE(a, ACM_ENOMEM, "acm_csr_create: host allocation failed");
    int st = alloc_arrays(a, vals_dev != nullptr);
    if (st == ACM_OK) {
        hipError_t e = hipMemcpy(a->indptr, indptr_dev, (size_t)(n_rows + 1) * sizeof(int32_t),
                                 hipMemcpyDeviceToDevice);
        if (e == hipSuccess && nnz)
            e = hipMemcpy(a->indices, indices_dev, (size_t)nnz * sizeof(int32_t),
                          hipMemcpyDeviceToDevice);
        if (e == hipSuccess && nnz && vals_dev)
            e = hipMemcpy(a->vals, vals_dev, (size_t)nnz * sizeof(float), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            acm_set_error("acm_csr_create: device copy failed: %s", hipGetErrorString(e));
            st = ACM_EHIP;
        }
    }
    if (st == ACM_OK) st = finish_handle(a, h_indptr, chunk);
    if (st != ACM_OK) {
        free_handle(a);
        return st;
    }
    *out = a;
    return ACM_OK;
}

extern "C" int acm_csr_transpose(const acm_csr_t* a, int chunk, acm_csr_t** out) {
    ACM_REQUIRE(a && out, ACM_EINVAL, "acm_csr_transpose: NULL argument");
    *out = nullptr;
    ACM_CHECK_HIP(hipDeviceSynchronize());
    const int64_t n = a->n_rows, m = a->n_cols, nnz = a->nnz;
    std::vector<int32_t> ip((size_t)n + 1), ix((size_t)nnz);
    std::vector<float> v((size_t)nnz);
    ACM_CHECK_HIP(hipMemcpy(ip.data(), a->indptr, ip.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nnz) {
        ACM_CHECK_HIP(hipMemcpy(ix.data(), a->indices, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (a->vals) ACM_CHECK_HIP(hipMemcpy(v.data(), a->vals, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost));
    }
    // stable counting sort by column: rows of A^T come out sorted by original row id
    std::vector<int32_t> tp((size_t)m + 1, 0), tx((size_t)nnz), tpos((size_t)nnz);
    std::vector<float> tv((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) ++tp[(size_t)ix[k] + 1];
    for (int64_t c = 0; c < m; ++c) tp[c + 1] += tp[c];
    std::vector<int32_t> cur(tp.begin(), tp.end() - 1);
    for (int64_t r = 0; r < n; ++r)
        for (int32_t k = ip[r]; k < ip[r + 1]; ++k) {
            const int32_t pos = cur[ix[k]]++;
            tx[pos] = (int32_t)r;
            tv[pos] = v[k];
            tpos[pos] = k;
        }
    acm_csr* t = new_handle(m, n, nnz);
    ACM_REQUIRE(t, ACM_ENOMEM, "acm_csr_transpose: host allocation failed");
    int st = alloc_arrays(t, a->vals != nullptr);
    if (st == ACM_OK) {
        hipError_t e = hipMemcpy(t->indptr, tp.data(), tp.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz)
            e = hipMemcpy(t->indices, tx.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz && t->vals)
            e = hipMemcpy(t->vals, tv.data(), (size_t)nnz * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&t->src_pos, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
        if (e == hipSuccess && nnz)
            e = hipMemcpy(t->src_pos, tpos.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            acm_set_error("acm_csr_transpose: upload failed: %s", hipGetErrorString(e));
            st = ACM_EHIP;
        }
    }
    if (st == ACM_OK) st = finish_handle(t, tp, chunk);
    if (st != ACM_OK) {
        free_handle(t);
        return st;
    }
    *out = t;
    return ACM_OK;
}

extern "C" int acm_csr_slice_rows(const acm_csr_t* a, int64_t row_begin, int64_t row_end, int chunk,
                                  acm_csr_t** out) {
    ACM_REQUIRE(a && out, ACM_EINVAL, "acm_csr_slice_rows: NULL argument");
    *out = nullptr;
    ACM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= a->n_rows, ACM_ESHAPE,
                "acm_csr_slice_rows: bad range [%lld,%lld) of %lld rows", (long long)row_begin,
                (long long)row_end, (long long)a->n_rows);
    ACM_CHECK_HIP(hipDeviceSynchronize());
    const int64_t n = row_end - row_begin;
    std::vector<int32_t> ip((size_t)n + 1);
    ACM_CHECK_HIP(hipMemcpy(ip.data(), a->indptr + row_begin, ip.size() * sizeof(int32_t),
                            hipMemcpyDeviceToHost));
    const int32_t base = ip[0];
    for (auto& x : ip) x -= base;
    const int64_t nnz = ip[n];
    acm_csr* s = new_handle(n, a->n_cols, nnz);
    ACM_REQUIRE(s, ACM_ENOMEM, "acm_csr_slice_rows: host allocation failed");
    int st = alloc_arrays(s, a->vals != nullptr);
    if (st == ACM_OK) {
        hipError_t e = hipMemcpy(s->indptr, ip.data(), ip.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz)
            e = hipMemcpy(s->indices, a->indices + base, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice);
        if (e == hipSuccess && nnz && a->vals)
            e = hipMemcpy(s->vals, a->vals + base, (size_t)nnz * sizeof(float), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            acm_set_error("acm_csr_slice_rows: copy failed: %s", hipGetErrorString(e));
            st = ACM_EHIP;
        }
    }
    if (st == ACM_OK) st = finish_handle(s, ip, chunk);
    if (st != ACM_OK) {
        free_handle(s);
        return st;
    }
    *out = s;
    return ACM_OK;
}

extern "C" void acm_csr_destroy(acm_csr_t* a) { free_handle(a); }

extern "C" int acm_shard_plan(int64_t n_rows, const int64_t* indptr, int world, int64_t row_cost, int64_t* bounds) {
    ACM_REQUIRE(indptr && bounds, ACM_EINVAL, "acm_shard_plan: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && world >= 1 && row_cost >= 0, ACM_ESHAPE, "acm_shard_plan: n_rows %lld world %d row_cost %lld",
                (long long)n_rows, world, (long long)row_cost);
    ACM_REQUIRE(indptr[0] == 0, ACM_EINVAL, "acm_shard_plan: indptr[0] must be 0");
    // cost prefix c(r) = indptr[r] + row_cost * r, non-decreasing in r
    auto cost = [&](int64_t r) { return (__int128)indptr[r] + (__int128)row_cost * r; };
    const __int128 total = cost(n_rows);
    bounds[0] = 0;
    for (int p = 1; p < world; ++p) {
        const __int128 target = total * p / world;
        int64_t lo = bounds[p - 1], hi = n_rows;          // smallest r in [lo, n_rows] with c(r) >= target
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (cost(mid) >= target) hi = mid; else lo = mid + 1;
        }
        int64_t r = lo;
        if (r > bounds[p - 1] && target - cost(r - 1) < cost(r) - target) --r;      // the nearer boundary
        bounds[p] = r;
    }
    bounds[world] = n_rows;
    return ACM_OK;
}


// ------------------------------------------------------------------ per-wave id streams
namespace {

template <class T>
int upload(T** dev, const std::vector<T>& host, size_t min_elems = 1) {
    const size_t n = std::max(host.size(), min_elems);
    ACM_CHECK_HIP(hipMalloc((void**)dev, n * sizeof(T)));
    if (!host.empty()) ACM_CHECK_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return ACM_OK;
}

}  // namespace

extern "C" int acm_csr_build_streams(acm_csr_t* a, int n_waves, int lmax) {
    ACM_REQUIRE(a, ACM_EINVAL, "acm_csr_build_streams: NULL handle");
    if (a->streams) return ACM_OK;
    ACM_REQUIRE(a->vals == nullptr, ACM_EUNSUPPORTED, "acm_csr_build_streams: pattern-only operators only");
    ACM_REQUIRE(a->n_cols < ACM_STREAM_SENTINEL, ACM_EUNSUPPORTED, "acm_csr_build_streams: %lld columns", (long long)a->n_cols);
    if (lmax <= 0) lmax = 512;
    ACM_REQUIRE(lmax % 32 == 0 && lmax <= 4096, ACM_EINVAL, "acm_csr_build_streams: lmax %d must be a multiple of 32, at most 4096", lmax);
    ACM_CHECK_HIP(hipDeviceSynchronize());
    const int64_t n = a->n_rows, nnz = a->nnz;
    std::vector<int32_t> ip((size_t)n + 1), ix((size_t)nnz);
    ACM_CHECK_HIP(hipMemcpy(ip.data(), a->indptr, ip.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nnz) ACM_CHECK_HIP(hipMemcpy(ix.data(), a->indices, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    // work items: whole rows of at most lmax neighbours, near-equal pieces (whole steps) of the longer ones
    struct Item { int32_t row, begin, len, slot; };
    std::vector<Item> items;
    items.reserve((size_t)n + (size_t)(nnz / lmax) + 16);
    std::vector<AcmLongRow> longs;
    int64_t n_slots = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int32_t b = ip[r], deg = ip[r + 1] - ip[r];
        if (deg <= lmax) {
            items.push_back({(int32_t)r, b, deg, -1});
            continue;
        }
        const int32_t pieces = (deg + lmax - 1) / lmax;
        const int32_t per = (((deg + pieces - 1) / pieces) + 31) / 32 * 32;
        AcmLongRow lr = {(int32_t)r, (int32_t)n_slots, 0, 0};
        for (int32_t q = 0; q * per < deg; ++q) items.push_back({(int32_t)r, b + q * per, std::min(per, deg - q * per), (int32_t)n_slots++});
        lr.slot_end = (int32_t)n_slots;
        longs.push_back(lr);
    }
    // by length (row order with round-robin dealing was measured and is slower: profiles/r02_probe_roles_order.txt)
    std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.len > y.len; });
    const int64_t n_items = (int64_t)items.size(), n_slices = (n_items + 3) / 4;
    std::vector<int32_t> sl_steps((size_t)n_slices);
    int64_t total_steps = 0;
    for (int64_t s = 0; s < n_slices; ++s) {
        int32_t longest = 0;
        for (int64_t i = s * 4; i < std::min(n_items, s * 4 + 4); ++i) longest = std::max(longest, items[(size_t)i].len);
        sl_steps[(size_t)s] = std::max(1, (longest + 31) / 32);
        total_steps += sl_steps[(size_t)s];
    }
    ACM_REQUIRE((total_steps + ACM_STREAM_PAD_STEPS) * 512 < (int64_t)0xFFFFFFF0u, ACM_EUNSUPPORTED,
                "acm_csr_build_streams: %lld steps exceed 32-bit stream offsets", (long long)total_steps);
    if (n_waves <= 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device);
        n_waves = cus * 4 * 5;                                       // five waves per SIMD (the kernel's occupancy)
    }
    if ((int64_t)n_waves > n_slices) n_waves = (int)std::max<int64_t>(n_slices, 1);
    n_waves = (n_waves + 3) / 4 * 4;
    // longest slice first, each to the least loaded wave (cost = steps + the row-local stage of its four rows)
    // cost of a slice in quarter steps: 4 per wave step + 1 per slice (descriptor + the finish of its four rows)
    const int64_t epi_cost = 1;
    std::vector<int32_t> wave_of((size_t)n_slices);
    {
        typedef std::pair<int64_t, int32_t> Load;
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int32_t w = 0; w < n_waves; ++w) heap.push({0, w});
        for (int64_t s = 0; s < n_slices; ++s) {
            Load l = heap.top();
            heap.pop();
            wave_of[(size_t)s] = l.second;
            l.first += 4 * (int64_t)sl_steps[(size_t)s] + epi_cost;
            heap.push(l);
        }
    }
    // wave-major order.  Within a wave the slices are SHUFFLED (a fixed permutation seeded by the wave's index): long slices are gather-bound (16 steps per row-local stage, neighbours all over the
    // table), short ones are bound by the row-local stage's VALU work; longest-first makes every wave -- the whole chip --
    // memory-bound first and VALU-bound last, so the two never overlap
    std::vector<int32_t> count((size_t)n_waves + 1, 0);
    for (int64_t s = 0; s < n_slices; ++s) ++count[(size_t)wave_of[(size_t)s] + 1];
    for (int32_t w = 0; w < n_waves; ++w) count[(size_t)w + 1] += count[(size_t)w];
    std::vector<int32_t> pos_of((size_t)n_slices), cur(count.begin(), count.end() - 1);
    for (int64_t s = 0; s < n_slices; ++s) pos_of[(size_t)s] = cur[(size_t)wave_of[(size_t)s]]++;
    {
        {
            std::vector<int32_t> slice_at((size_t)n_slices);
            for (int64_t s = 0; s < n_slices; ++s) slice_at[(size_t)pos_of[(size_t)s]] = (int32_t)s;
            for (int32_t w = 0; w < n_waves; ++w) {
                const int32_t b = count[(size_t)w], m = count[(size_t)w + 1] - b;
                uint64_t state = 0x9E3779B97F4A7C15ull * (uint64_t)(w + 1);
                for (int32_t i = m - 1; i > 0; --i) {                // Fisher-Yates with a 64-bit LCG
                    state = state * 6364136223846793005ull + 1442695040888963407ull;
                    const int32_t j = (int32_t)((state >> 33) % (uint64_t)(i + 1));
                    std::swap(slice_at[(size_t)(b + i)], slice_at[(size_t)(b + j)]);
                }
            }
            for (int64_t q = 0; q < n_slices; ++q) pos_of[(size_t)slice_at[(size_t)q]] = (int32_t)q;
        }
    }
    std::vector<int32_t> h_steps((size_t)n_slices + 2, 1), h_items(((size_t)n_slices + 2) * 16, -1), step_base((size_t)n_slices + 1, 0);
    for (int64_t s = 0; s < n_slices; ++s) h_steps[(size_t)pos_of[(size_t)s]] = sl_steps[(size_t)s];
    for (int64_t q = 0; q < n_slices; ++q) step_base[(size_t)q + 1] = step_base[(size_t)q] + h_steps[(size_t)q];
    std::vector<int32_t> h_waves((size_t)n_waves * 4);
    for (int32_t w = 0; w < n_waves; ++w) {
        const int32_t b = count[(size_t)w], e = count[(size_t)w + 1];
        h_waves[(size_t)w * 4 + 0] = b;
        h_waves[(size_t)w * 4 + 1] = e;
        h_waves[(size_t)w * 4 + 2] = step_base[(size_t)b];
        h_waves[(size_t)w * 4 + 3] = step_base[(size_t)e] - step_base[(size_t)b];
    }
    std::vector<int32_t> h_ids(((size_t)total_steps + ACM_STREAM_PAD_STEPS) * 128, ACM_STREAM_SENTINEL);
    for (int64_t s = 0; s < n_slices; ++s) {
        const size_t q = (size_t)pos_of[(size_t)s];
        for (int g = 0; g < 4; ++g) {
            const int64_t i = s * 4 + g;
            if (i >= n_items) break;
            const Item& it = items[(size_t)i];
            h_items[q * 16 + 4 * g + 0] = it.row;
            h_items[q * 16 + 4 * g + 1] = it.slot;
            int32_t* dst = h_ids.data() + (size_t)step_base[q] * 128 + (size_t)g * 32;
            for (int32_t k = 0; k < it.len; ++k) dst[(size_t)(k >> 5) * 128 + (k & 31)] = ix[(size_t)it.begin + k];
        }
    }
    for (size_t q = 0; q < (size_t)n_slices + 2; ++q)
        for (int g = 0; g < 4; ++g) h_items[q * 16 + 4 * g + 2] = h_steps[q], h_items[q * 16 + 4 * g + 3] = 0;
    AcmStreams* t = new (std::nothrow) AcmStreams();
    ACM_REQUIRE(t, ACM_ENOMEM, "acm_csr_build_streams: host allocation failed");
    memset(t, 0, sizeof(*t));
    t->total_steps = total_steps;
    t->n_slices = n_slices;
    t->n_long = (int64_t)longs.size();
    t->n_slots = n_slots;
    t->n_waves = n_waves;
    t->lmax = lmax;
    int st = upload(&t->ids, h_ids);
    if (st == ACM_OK) st = upload(&t->waves, h_waves);
    if (st == ACM_OK) st = upload(&t->items, h_items);
    if (st == ACM_OK && !longs.empty()) {
        st = upload(&t->long_rows, longs);
        std::vector<int32_t> index((size_t)n, -1);
        for (size_t i = 0; i < longs.size(); ++i) index[(size_t)longs[i].row] = (int32_t)i;
        if (st == ACM_OK) st = upload(&t->long_index, index);
        if (st == ACM_OK && hipMalloc((void**)&t->counters, longs.size() * sizeof(int32_t)) != hipSuccess) st = ACM_EHIP;
        if (st == ACM_OK && hipMemset(t->counters, 0, longs.size() * sizeof(int32_t)) != hipSuccess) st = ACM_EHIP;
        if (st == ACM_OK && hipMalloc((void**)&t->slots, (size_t)n_slots * 8 * sizeof(float)) != hipSuccess) st = ACM_EHIP;
        if (st == ACM_EHIP) acm_set_error("acm_csr_build_streams: device allocation failed");
    }
    if (st != ACM_OK) {
        free_streams(t);
        return st;
    }
    a->streams = t;
    return ACM_OK;
}

extern "C" int acm_csr_build_item_streams(acm_csr_t* a, int n_waves) {
    ACM_REQUIRE(a, ACM_EINVAL, "acm_csr_build_item_streams: NULL handle");
    if (a->item_streams) return ACM_OK;
    ACM_REQUIRE(a->vals == nullptr, ACM_EUNSUPPORTED, "acm_csr_build_item_streams: pattern-only operators only");
    ACM_REQUIRE(a->n_items > 0 && a->nnz > 0, ACM_EUNSUPPORTED, "acm_csr_build_item_streams: empty operator");
    ACM_REQUIRE(a->n_cols < ((int64_t)1 << 31) - 1, ACM_EUNSUPPORTED, "acm_csr_build_item_streams: %lld columns", (long long)a->n_cols);
    ACM_CHECK_HIP(hipDeviceSynchronize());
    const int64_t n_items = a->n_items, nnz = a->nnz;
    std::vector<AcmItem> items((size_t)n_items);
    std::vector<AcmLongRow> longs((size_t)a->n_long);
    std::vector<int32_t> ix((size_t)nnz), lidx;
    ACM_CHECK_HIP(hipMemcpy(items.data(), a->items, items.size() * sizeof(AcmItem), hipMemcpyDeviceToHost));
    ACM_CHECK_HIP(hipMemcpy(ix.data(), a->indices, ix.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (a->n_long) {
        ACM_REQUIRE(a->long_index, ACM_EUNSUPPORTED, "acm_csr_build_item_streams: handle without a long-row index");
        lidx.resize((size_t)a->n_rows);
        ACM_CHECK_HIP(hipMemcpy(longs.data(), a->long_rows, longs.size() * sizeof(AcmLongRow), hipMemcpyDeviceToHost));
        ACM_CHECK_HIP(hipMemcpy(lidx.data(), a->long_index, lidx.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    const int64_t n_quads = (n_items + 3) / 4;
    std::vector<int32_t> qb((size_t)n_quads, 0);                      // batches of a quad
    int64_t total = 0;
    for (int64_t i = 0; i < n_items; ++i) {
        const int32_t nb = (items[(size_t)i].end - items[(size_t)i].begin + 31) / 32;
        qb[(size_t)(i / 4)] += nb;
        total += nb;
    }
    ACM_REQUIRE((total + ACM_ITEM_STREAM_PAD) * 128 < ((int64_t)1 << 32), ACM_EUNSUPPORTED,       // (the kernels keep BYTE offsets in 32 bits)
                "acm_csr_build_item_streams: %lld batches exceed 32-bit stream offsets", (long long)total);
    if (n_waves <= 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device);
        n_waves = cus * 8;                                           // two waves per SIMD (the kernels' occupancy)
    }
    if ((int64_t)n_waves > n_quads) n_waves = (int)n_quads;
    n_waves = (n_waves + 3) / 4 * 4;
    // longest quad first, each to the least loaded wave; cost in batches: its batches + the four item ends and the rows' epilogue
    const int64_t quad_cost = 10;
    std::vector<int32_t> order((size_t)n_quads), wave_of((size_t)n_quads);
    for (int64_t q = 0; q < n_quads; ++q) order[(size_t)q] = (int32_t)q;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return qb[(size_t)x] > qb[(size_t)y]; });
    {
        typedef std::pair<int64_t, int32_t> Load;
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int32_t w = 0; w < n_waves; ++w) heap.push({0, w});
        for (int64_t o = 0; o < n_quads; ++o) {
            Load l = heap.top();
            heap.pop();
            wave_of[(size_t)order[(size_t)o]] = l.second;
            l.first += qb[(size_t)order[(size_t)o]] + quad_cost;
            heap.push(l);
        }
    }
    std::vector<int32_t> count((size_t)n_waves + 1, 0);
    for (int64_t q = 0; q < n_quads; ++q) ++count[(size_t)wave_of[(size_t)q] + 1];
    for (int32_t w = 0; w < n_waves; ++w) count[(size_t)w + 1] += count[(size_t)w];
    std::vector<int32_t> cur(count.begin(), count.end() - 1), quad_at((size_t)n_quads);
    for (int64_t o = 0; o < n_quads; ++o) quad_at[(size_t)cur[(size_t)wave_of[(size_t)order[(size_t)o]]]++] = order[(size_t)o];
    const int32_t zero_row = (int32_t)a->n_cols;
    std::vector<int32_t> h_quads((size_t)n_quads * 16, 0), h_waves((size_t)n_waves * 4, 0),
        h_ids(((size_t)total + ACM_ITEM_STREAM_PAD) * 32, zero_row);
    int64_t batch = 0;
    for (int32_t w = 0; w < n_waves; ++w) {
        h_waves[(size_t)w * 4 + 0] = count[(size_t)w];
        h_waves[(size_t)w * 4 + 1] = count[(size_t)w + 1];
        h_waves[(size_t)w * 4 + 2] = (int32_t)batch;
        for (int32_t pos = count[(size_t)w]; pos < count[(size_t)w + 1]; ++pos) {
            const int64_t q = quad_at[(size_t)pos];
            for (int g = 0; g < 4; ++g) {
                int32_t* d = h_quads.data() + (size_t)pos * 16 + 4 * g;
                const int64_t i = q * 4 + g;
                d[1] = -1;
                if (i >= n_items) continue;
                const AcmItem& it = items[(size_t)i];
                const int32_t len = it.end - it.begin, nb = (len + 31) / 32;
                bool owner = it.slot < 0;
                if (!owner) owner = longs[(size_t)lidx[(size_t)it.row]].slot_begin == it.slot;
                d[0] = it.row, d[1] = it.slot, d[2] = nb, d[3] = 1 | (owner ? 2 : 0);
                for (int32_t k = 0; k < len; ++k) h_ids[(size_t)batch * 32 + (size_t)k] = ix[(size_t)it.begin + (size_t)k];
                batch += nb;
            }
        }
        h_waves[(size_t)w * 4 + 3] = (int32_t)batch - h_waves[(size_t)w * 4 + 2];
    }
    AcmItemStreams* t = new (std::nothrow) AcmItemStreams();
    ACM_REQUIRE(t, ACM_ENOMEM, "acm_csr_build_item_streams: host allocation failed");
    memset(t, 0, sizeof(*t));
    t->total_batches = total;
    t->n_quads = n_quads;
    t->n_waves = n_waves;
    int st = upload(&t->ids, h_ids);
    if (st == ACM_OK) st = upload(&t->quads, h_quads);
    if (st == ACM_OK) st = upload(&t->waves, h_waves);
    if (st != ACM_OK) {
        free_item_streams(t);
        return st;
    }
    a->item_streams = t;
    return ACM_OK;
}

extern "C" int acm_csr_info(const acm_csr_t* a, acm_csr_info_t* info) {
    ACM_REQUIRE(a && info, ACM_EINVAL, "acm_csr_info: NULL argument");
    info->n_rows = a->n_rows;
    info->n_cols = a->n_cols;
    info->nnz = a->nnz;
    info->n_items = a->n_items;
    info->n_long_rows = a->n_long;
    info->n_partial_slots = a->n_slots;
    info->chunk = a->chunk;
    info->max_degree = a->max_degree;
    info->indptr = a->indptr;
    info->indices = a->indices;
    info->vals = a->vals;
    info->src_pos = a->src_pos;
    info->stream_steps = a->streams ? a->streams->total_steps : 0;
    info->stream_slices = a->streams ? a->streams->n_slices : 0;
    info->stream_waves = a->streams ? a->streams->n_waves : 0;
    info->stream_long_rows = a->streams ? (int32_t)a->streams->n_long : 0;
    info->item_stream_waves = a->item_streams ? a->item_streams->n_waves : 0;
    info->reserved = 0;
    info->item_stream_batches = a->item_streams ? a->item_streams->total_batches : 0;
    return ACM_OK;
}

extern "C" int acm_spmm_workspace_bytes(const acm_csr_t* a, int width, size_t* bytes) {
    ACM_REQUIRE(a && bytes, ACM_EINVAL, "acm_spmm_workspace_bytes: NULL argument");
    ACM_REQUIRE(width > 0, ACM_ESHAPE, "acm_spmm_workspace_bytes: width must be positive");
    *bytes = (size_t)a->n_slots * (size_t)width * sizeof(float);
    return ACM_OK;
}
