/*
 * acm_hip.h -- C ABI of libacm_hip.so, the MI355X (gfx950) implementation of the
 * ACM graph-convolution hot path.
 *
 * The reference (SitaoLuan/ACM-GNN) is pure Python; its "FFI" for this path is
 * the set of torch ATen calls issued by GraphConvolution.forward and by the
 * autograd graph they record.  Every entry point below names the reference
 * call sites it replaces (paths relative to the reference root; G =
 * ACM-Geometric/layers.py, P = ACM-Pytorch/models/layers.py).
 *
 * Conventions
 *   - plain C: opaque handles, POD structs of device pointers and sizes, no
 *     torch/C++ types, no exceptions across the boundary.
 *   - every function returns ACM_OK (0) or an acm_status_t; the message for
 *     the last failure on the calling thread is acm_last_error().
 *   - all device buffers (inputs, outputs, saved tensors, workspaces) are owned
 *     by the caller; the library allocates device memory only inside
 *     acm_csr_create / acm_csr_transpose / acm_csr_slice_rows.
 *   - all launches are asynchronous on the hipStream_t passed in (as void*);
 *     no internal synchronisation => hipGraph-capturable.  The only process-wide state is the
 *     thread-local error string and the tuning record below (acm_tuning_t): no entry point reads
 *     the environment.
 *   - fp32 everywhere (the reference computes in torch.FloatTensor, G:19-28),
 *     int32 indices, row-major dense matrices with explicit leading dimensions.
 */
#ifndef ACM_HIP_H
#define ACM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACM_ABI_VERSION 28

typedef enum {
    ACM_OK = 0,
    ACM_EINVAL = 1,        /* null pointer / bad flag                          */
    ACM_ESHAPE = 2,        /* inconsistent sizes                               */
    ACM_EHIP = 3,          /* a HIP runtime call failed (text in last_error)   */
    ACM_EUNSUPPORTED = 4,  /* valid request outside the implemented envelope   */
    ACM_ENOMEM = 5,        /* caller workspace too small                       */
} acm_status_t;

typedef struct acm_csr acm_csr_t;   /* opaque: CSR row block + balanced work list */
typedef void* acm_stream_t;         /* hipStream_t                                */

int acm_version(void);
const char* acm_last_error(void);

/* ---------------------------------------------------------------- tuning --
 * The ONE dispatch mechanism of the library: where two execution forms compute the same result (up to fp32
 * re-association; the parity tests compare them with each other and with the oracle), this record says which one
 * the entry points launch.  It is filled ONCE, when the library is loaded, from the environment variable
 *     ACM_TUNING="key=value,key=value,..."        (keys = the field names below; unknown keys are an error at load)
 * and changed afterwards only through acm_tuning_set -- no acm_* launch path calls getenv.  Process-wide, not
 * synchronised: set it before launching, not while other threads launch.  Every value is validated by
 * acm_tuning_set (ACM_EINVAL names the field).  No reference counterpart (the reference has one execution form:
 * the ATen calls of ACM-Geometric/layers.py:78-116). */
typedef struct {
    int32_t chunk;        /* work-item length of acm_csr_create / _transpose / _slice_rows when their `chunk` argument
                             is <= 0: 0 = sized to the graph (128..1024, see acm_csr_create), else a power of two 8..4096 */
    int32_t wide_form;    /* gathers of rows wider than 8 floats: 0 = chosen per call (table size, width, operand type),
                             1 = one dword per lane (spmm_wide_kernel), 2 = four neighbours per dwordx4 instruction
                             (spmm_vec_kernel), 3 = two neighbours per instruction (spmm_pair_kernel) where it exists */
    int32_t bwd_split;    /* acm_conv_bwd_spmm (K4): -1 = chosen per call (one channel per pass for tables beyond the L2 at
                             mean degree >= 32), 0 = all channels in one pass, 1 = one channel per pass */
    int32_t rows16;       /* bit mask of the sixteen-rows-per-wave row-local kernels: 1 = forward stage (agg_epi16_kernel),
                             2 = aggregate-first backward (agg_bwd16_kernel, also the carrier of proj_* / next_agg),
                             4 = literal K3 (bwd_local16_kernel).  Default 7; a cleared bit falls back to the
                             four-rows-per-wave kernels */
    int32_t agg_fused;    /* acm_conv_agg_fwd: 1 = gather + row-local stage in one kernel where the shape allows (default),
                             0 = always two stages (acm_spmm_ex, then the row-local kernel) */
    int32_t gemm_forms;   /* bit mask for tall products: 1 = row-panel fp32 kernels (acm_gemm_rows.hip), 2 = split-bf16
                             projections for K <= 128 (acm_gemm_bx3.hip), 4 = split-bf16 TN form for K > 128 from
                             16 384 rows, 8 = row-panel kernels for EVERY shape they cover (tests).  Default 7;
                             0 = the 64x64 tile kernel only */
    int32_t reserved[10]; /* zero */
} acm_tuning_t;
int acm_tuning_get(acm_tuning_t* out);
int acm_tuning_set(const acm_tuning_t* in);   /* NULL: back to the load-time record (defaults + ACM_TUNING) */

/* ------------------------------------------------------------------ graph --
 * acm_csr_create: adopt (copy) a CSR row block living in device memory and
 * build the nnz-balanced work list the kernels walk (rows longer than `chunk`
 * neighbours are split into several work items whose partial sums are combined
 * deterministically in a second phase).  Rows are local (0..n_rows), column
 * ids index the gathered matrix (0..n_cols) -- for a row-sharded graph n_cols
 * is the global node count.
 * Replaces: the sparse-COO tensors built at ACM-Geometric/train.py:75-81 /
 * ACM-Pytorch/utils.py:619-629 and the per-call COO coalesce inside
 * torch.spmm (G:87-103).  `chunk` <= 0 sizes the chunks to the graph: a power of two in 128..1024
 * chosen so that one work item stays below the share of a 16-lane group when the chip is full
 * (nnz / 8192); acm_tuning_t.chunk overrides that choice.  The same rule
 * applies to acm_csr_transpose and acm_csr_slice_rows (computed from the new handle's own nnz).
 *
 * vals_dev == NULL makes a PATTERN-ONLY operator: every stored entry counts as 1 and the kernels read no
 * value stream.  This is the form the filterbank wants: A_low = D^-1 (A + I) has one value per row, so
 *     A_low   G = D^-1 (P G)            P = pattern of (A + I), symmetric for the reference's graphs
 *     A_low^T G = P (D^-1 G)
 * -- one 4-byte column-id stream shared by the forward and the backward products (and small enough to
 * stay in the 256 MB Infinity Cache between them) instead of two 8-byte (id, value) streams; the row
 * scales ride the epilogues (row_scale / g_scale / self_scale fields below).  A column may repeat
 * inside a row (a raw self-loop in A makes the diagonal of A + I count twice).
 */
int acm_csr_create(int64_t n_rows, int64_t n_cols, int64_t nnz,
                   const int32_t* indptr_dev, const int32_t* indices_dev,
                   const float* vals_dev, int chunk, acm_csr_t** out);
/* Transposed operator (n_cols x n_rows) for the backward SpMM; what autograd's
 * SparseAddmmBackward computes implicitly for loss.backward()
 * (ACM-Geometric/train.py:135). Deterministic (stable counting sort). */
int acm_csr_transpose(const acm_csr_t* a, int chunk, acm_csr_t** out);
/* Row block [row_begin,row_end) of an existing operator (multi-GPU row shard). */
int acm_csr_slice_rows(const acm_csr_t* a, int64_t row_begin, int64_t row_end,
                       int chunk, acm_csr_t** out);
void acm_csr_destroy(acm_csr_t* a);

/* acm_shard_plan: the row partition of a multi-GPU run ("contiguous, nnz-balanced row
 * blocks"; the reference is single-device -- ACM-Geometric/layers.py:10-11, models.py:20-21 -- and uses several
 * GPUs only as replicas, sh/run_all_settings.sh:2-24, so this has no reference counterpart).  Host code, no
 * device access: `indptr_host` is the CSR row-pointer array of the GLOBAL operator in host memory
 * (n_rows + 1 entries).  Writes bounds[0..world]: rank p owns rows [bounds[p], bounds[p+1]).  The cuts
 * equalise the prefix sum of  nnz(row) + row_cost  (row_cost >= 0: the per-row work of the row-local
 * kernels expressed in gathered edges; 0 = pure nnz balance); each cut is the row boundary nearest to
 * p / world of the total, found by binary search, and cuts never cross (a row heavier than one share gets a
 * block of its own).  Deterministic: every rank computes the same plan from the same indptr. */
int acm_shard_plan(int64_t n_rows, const int64_t* indptr_host, int world, int64_t row_cost,
                   int64_t* bounds_host);

typedef struct {
    int64_t n_rows, n_cols, nnz;
    int64_t n_items;          /* work items (one wave / lane-group each)  */
    int64_t n_long_rows;      /* rows split over several items            */
    int64_t n_partial_slots;  /* partial-sum slots those rows need        */
    int32_t chunk;
    int32_t max_degree;
    const int32_t* indptr;    /* device pointers owned by the handle      */
    const int32_t* indices;
    const float* vals;
    const int32_t* src_pos;   /* handles made by acm_csr_transpose: position of each entry in the source
                                 operator's value array (vals_T[k] = vals[src_pos[k]]); NULL otherwise */
    int64_t stream_steps;     /* acm_csr_build_streams: wave steps (128 id slots each), 0 = not built   */
    int64_t stream_slices;    /* slices (four work items each)                                          */
    int32_t stream_waves;     /* waves the streams are cut for = 4 x the blocks of the streamed kernel  */
    int32_t stream_long_rows; /* rows cut into pieces (combined by the last piece to arrive)            */
    int32_t item_stream_waves; /* acm_csr_build_item_streams: waves, 0 = not built (ABI 24)             */
    int32_t reserved;
    int64_t item_stream_batches; /* 32-neighbour batches over all waves                                 */
} acm_csr_info_t;
int acm_csr_info(const acm_csr_t* a, acm_csr_info_t* info);

/* acm_csr_build_streams: lay the column ids of a PATTERN-ONLY operator out in the order the waves of a streamed
 * gather consume them (the gather waves of acm_conv_agg_bwd_t.next_agg) -- a sliced-ELL copy of the id
 * stream: four rows of similar length per wave ("slice"), 32 neighbours per row and wave step, 128 ids per step padded
 * with an out-of-range sentinel (a buffer load answers it with zeros without touching memory), the slices dealt
 * longest-first to `n_waves` waves, each wave's slices contiguous.  A wave then walks ONE linear id stream with its
 * loads issued ahead, takes its slice descriptors through the scalar unit and runs no per-lane bounds logic.  Rows longer
 * than `lmax` neighbours are cut into pieces; the piece that arrives last (a device-scope arrival counter per row) adds
 * the partial sums in slot order and finishes the row, so results do not depend on the arrival order.
 * One-off host-side preprocessing like acm_csr_create (synchronises the device; must not be called while a stream is
 * capturing); idempotent.  n_waves <= 0: five waves per SIMD of the current device;
 * lmax <= 0: 512.  The handle owns the arrival counters (zero between launches: the last piece of a row resets its counter) and partial
 * slots, so launches that use the streams of one handle must be stream-ordered.  Costs (total steps x 512 B) of device memory, ~1.2 x the id array.
 * No reference counterpart (the reference hands torch.spmm a COO tensor, ACM-Geometric/layers.py:87-103). */
int acm_csr_build_streams(acm_csr_t* a, int n_waves, int lmax);

/* acm_csr_build_item_streams (ABI 24): per-wave BATCH streams over the handle's own work items, for kernels that take one
 * item at a time with 32 neighbours per wave step (acm_conv_acmii_v_fwd / _bwd).  The item list (rows, and the pieces of the
 * long rows with their partial slots -- the same items every other kernel walks) is cut into quads of four consecutive items;
 * quads are dealt longest-first to the least loaded of `n_waves` persistent waves; a wave's quads, and the column ids of
 * their batches, are contiguous, idle slots padded with n_cols (the index of an all-zero row the caller appends to its
 * table).  A wave then walks ONE linear id stream, fetches ahead across row boundaries, reads its parameters once, and
 * the waves' loads are balanced when the handle is built instead of by the dispatcher.  Same contract as
 * acm_csr_build_streams (one-off, synchronises, idempotent, pattern-only); n_waves <= 0: two waves per SIMD.
 * acm_csr_info_t.item_stream_waves says whether (and for how many waves) they exist.  No reference counterpart. */
int acm_csr_build_item_streams(acm_csr_t* a, int n_waves);

/* Bytes of caller-provided workspace the SpMM-type entry points need for an
 * operator when `width` fp32 columns are accumulated per row. */
int acm_spmm_workspace_bytes(const acm_csr_t* a, int width, size_t* bytes);

/* ----------------------------------------------------------------- GEMM --
 * C[M,N] = op(A)[M,K] * op(B)[K,N]  (fp32 in / fp32 accumulate on the f32 MFMA
 * pipe), optional ReLU epilogue.  transX = 0: X is stored row-major as written;
 * 1: stored transposed (so op(X) = X^T).
 * Replaces torch.mm(input, self.weight_*) G:87-89,96-104 / P:162-194 (one call
 * with the three weights concatenated), and the MmBackward GEMMs X^T*dZ and
 * dZ*W^T.  Workspace: acm_gemm_workspace_bytes (split-K partial slabs).
 */
int acm_gemm_workspace_bytes(int transA, int transB, int64_t M, int64_t N, int64_t K,
                             size_t* bytes);
int acm_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K,
             const float* A, int64_t lda, const float* B, int64_t ldb,
             float* C, int64_t ldc, int relu, void* workspace, size_t workspace_bytes,
             acm_stream_t stream);

/* Same product with the columns of C delivered as consecutive blocks of `c_col_block` columns, block j starting
 * at C + j * c_block_stride (rows of a block keep the pitch ldc >= c_col_block):
 *     C_j[m, n'] = (op(A) op(B))[m, j * c_col_block + n'].
 * dWcat = X^T [dZ_L | dZ_H | dZ_I] lands as three contiguous F_in x F matrices -- the layout of
 * weight_low / weight_high / weight_mlp (layers.py:19-21), so autograd adopts them without a copy each.
 * c_col_block = 0 is acm_gemm. */
int acm_gemm_blocks(int transA, int transB, int64_t M, int64_t N, int64_t K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, int64_t c_col_block, int64_t c_block_stride, int relu,
                    void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* -------------------------------------------------------- counter-based dropout --
 * The reference applies F.dropout to the input and to relu(layer-1 output) (ACM-Geometric/models.py:54,70;
 * ACM-Pytorch/models/models.py:149,164).  torch materialises a mask tensor per call; here the keep/drop
 * decision of element (row, col) is a pure function
 *     keep = Philox4x32-7( key = seed, counter = (global row, block(col) | tag << 16, step) ).word(col) >= p * 2^32
 *     block(col) = (col & 15) + 16 * (col >> 6),   word(col) = (col >> 4) & 3
 * so the forward and the backward kernels regenerate it in registers instead of writing / reading an
 * [n, F] mask (and nothing inside a captured step touches torch's generator).  `step` is a device counter
 * the training loop advances once per optimizer step (acm_adam_config_t.also_advance does it for free), so a
 * replayed hipGraph draws fresh masks; `tag` separates the masks drawn within one step.  Surviving elements
 * are scaled by 1 / (1 - p) like torch.  p = 0 disables.
 */
typedef struct {
    float    p;
    int32_t  tag;
    uint64_t seed;
    const int64_t* step;       /* device */
    int64_t  row_offset;       /* global index of local row 0 (a row shard draws the single-process mask) */
    int64_t  step_offset;      /* added to *step: 1 draws the masks of the NEXT optimizer step (input pipelining)  */
} acm_dropout_t;

/* dst[r, c] = src[r, c] * keep(r, c) / (1 - p) for c < n_cols, 0 for n_cols <= c < dst_cols (row padding for the
 * 16-byte gathers of the aggregate-first path).  src may equal dst when dst_cols == n_cols. */
int acm_dropout(int64_t n_rows, int64_t n_cols, const float* src, int64_t ld_src,
                float* dst, int64_t ld_dst, int64_t dst_cols, const acm_dropout_t* d, acm_stream_t stream);

/* The same product with the counter-based dropout (acm_dropout_t) applied to the STORED matrix A while
 * its tiles are staged -- A = the node-feature matrix X, whichever side of the product it is on:
 *     transA = 0:  C = drop(A) B        Z  = dropout(X) W        (ACM-Geometric/models.py:54 + layers.py:86-88,101-103)
 *     transA = 1:  C = drop(A)^T B      dW = dropout(X)^T dZ     (the MmBackward of the same)
 * so the dropped copy of X is never written (173 MB written and read on the arXiv-year-shaped graph).  Carried by the
 * row-panel kernels only (A much taller than wide: >= 4096 rows, 16..4096 columns; N <= 192; for transA = 1: >= 8192 rows
 * and at most 128 columns of A): ACM_EUNSUPPORTED otherwise -- the caller then applies acm_dropout itself.
 * a_drop NULL or p = 0: acm_gemm_blocks.  Workspace: acm_gemm_workspace_bytes. (ABI 20) */
int acm_gemm_drop(int transA, int transB, int64_t M, int64_t N, int64_t K,
                  const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, int64_t c_col_block, int64_t c_block_stride, int relu,
                  const acm_dropout_t* a_drop, void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* The three projections of an ACM layer with the weight matrices read IN PLACE (ABI 21):
 *     [Z_L 0 | Z_H 0 | Z_I] = relu?( drop?(X) [W_L 0 | W_H 0 | W_I] )        ACM-Geometric/layers.py:86-88,101-103 (+ models.py:54)
 * -- torch.mm(input, self.weight_*) x 3 as ONE launch without a packed copy of the weights (no torch.cat / fill launches in
 * the step).  w_*: [K, f], row pitch ld_w.  The first two channels sit in blocks of f_block >= f columns (zero columns
 * between: the 16-byte rows the narrow gathers fetch, see acm_conv_fwd), the product has 2 f_block + f columns; columns
 * [0, split_col) go to C, the rest to C2 (split_col = 0: all to C) -- the two tables of a narrow layer.  x_drop: the
 * caller's input dropout drawn while X is loaded (NULL / p = 0: none).  Runs on the split-bf16 row-panel kernel (fp32
 * accuracy, acm_gemm_bx3.hip): n_rows >= 8192, 32 <= K <= 128 with K % 4 == 0 and 16-byte aligned rows of X, at most 192
 * product columns; ACM_EUNSUPPORTED otherwise (the caller packs the weights and calls acm_gemm / acm_gemm_split). */
int acm_proj3(int64_t n_rows, int64_t K, const float* X, int64_t ldx,
              const float* w_low, const float* w_high, const float* w_mlp, int64_t ld_w, int64_t f, int64_t f_block,
              float* C, int64_t ldc, int64_t split_col, float* C2, int64_t ldc2, int relu,
              const acm_dropout_t* x_drop, acm_stream_t stream);

/* ------------------------------------------------ deferred final reductions --
 * Every backward kernel that produces parameter gradients (acm_conv_bwd_local, acm_proj_bwd, acm_conv_agg_bwd) and
 * acm_nll_loss end in the same second phase: per-block partial sums in the call's workspace, summed by one block per
 * output element in a fixed order.  On the GPU each of those second phases is a launch of its own (~5 us in a
 * replayed graph), and nothing reads their results before the optimizer.  A caller that owns the whole step may
 * therefore hand each of those calls an acm_reduce_list_t: the call then runs its main kernel, APPENDS the
 * description of its second phase to the list instead of launching it, and acm_reduce_flush runs all of them in one
 * launch (per 32 segments).  Contract while a call's segments are pending: its workspace must stay allocated and
 * untouched, and its reduced outputs (the loss, d_att_vec / d_ln_* / d_att_mix, dW, d_params) are undefined until
 * the flush.  The summation order is the one of the immediate form (thread t of 256 adds blocks t, t + 256, ...,
 * then a binary tree), so deferred and immediate results are bit-identical.
 * `defer` == NULL everywhere means "reduce now" (the only behaviour before ABI 14).
 * Replaces: nothing in the reference (autograd reduces inside each op); this is launch-count plumbing for the
 * captured training step of ACM-Geometric/train.py:119-137.
 */
typedef struct {
    const float* partial;      /* [nblk rows] x row_stride floats                                              */
    int32_t nblk, row_stride;
    int32_t q0, len;           /* this segment sums columns q0 .. q0 + len of `partial`                        */
    float*  dst;               /* element e -> dst[(e / inner) * outer_stride + blk(e % inner)]                */
    int32_t inner;             /* columns per destination row (len for a flat vector)                          */
    int32_t col_block;         /* 0: blk(q) = q;  else blk(q) = (q / col_block) * block_stride + q % col_block */
    int64_t outer_stride, block_stride;
    int32_t elem_stride;       /* 0: partial[b][q] at b * row_stride + q (row-major slabs, row_stride >= q0 + len);
                                * > 0: slabs stored in GROUPS of row_stride elements, partial[q / row_stride][b][q % row_stride]
                                * at (q / row_stride) * elem_stride + b * row_stride + q % row_stride, elem_stride >= nblk *
                                * row_stride.  With row_stride == 32 (one 128-byte line per block and group) a block of the
                                * second phase sums a whole group, reading one line per producer block instead of one float
                                * out of each of 32 lines -- same order of additions per element, bit-identical results  */
    int32_t reserved;
} acm_reduce_seg_t;

typedef struct {
    int32_t n, cap;            /* segments in use / allocated                                                   */
    acm_reduce_seg_t* segs;    /* host memory owned by the caller                                               */
} acm_reduce_list_t;

/* Run every pending segment of `list` on `stream` (ceil(n / 32) launches) and reset list->n to 0. */
int acm_reduce_flush(acm_reduce_list_t* list, acm_stream_t stream);

/* Same product delivered as two matrices: columns [0, split_col) to C (pitch ldc), the rest to C2 (pitch ldc2).
 * The projection of a narrow layer writes the gathered block [Z_L | Z_H] as its own compact 2F-float rows (the table
 * the fused SpMM gathers from: half the cache footprint of [Z_L | Z_H | Z_I | pad] rows) and Z_I next to it. */
int acm_gemm_split(int transA, int transB, int64_t M, int64_t N, int64_t K,
                   const float* A, int64_t lda, const float* B, int64_t ldb,
                   float* C, int64_t ldc, int64_t split_col, float* C2, int64_t ldc2, int relu,
                   void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* The projection of a narrow layer (f_out <= 8 columns per weight, e.g. the output layer) and its backward, as
 * streaming kernels over X instead of skinny GEMMs, reading the layer's three weight matrices where they are
 * (each f_in x f_out, pitch ldw: no packed [W_L | W_H | W_I] copy):
 *     acm_proj_fwd:  [Z_lh | Z_i] = relu?(X [W_L | W_H | W_I]),  Z_lh = the gathered block [Z_L | Z_H] (n x 2 f_out,
 *                    compact: the table the fused SpMM walks), Z_i (n x f_out)       -- torch.mm x3, layers.py:87-89
 *     acm_proj_bwd:  dX = dZ [W_L | W_H | W_I]^T  and  dW = X^T dZ (as column blocks like acm_gemm_blocks) in ONE
 *                    pass over X; n_out = 3 f_out in {3, 6, 9, 12, 15}                -- MmBackward of the same
 * Deterministic (fixed reduction order).  Workspace of the backward: acm_proj_bwd_workspace_bytes. */
int acm_proj_fwd(int64_t n_rows, int64_t f_in, int f_out, const float* X, int64_t ldx,
                 const float* w_low, const float* w_high, const float* w_mlp, int64_t ldw, int relu,
                 float* Z_lh, int64_t ld_lh, float* Z_i, int64_t ld_i, acm_stream_t stream);
/* The same with Z_H written at column h_col >= f_out of Z_lh instead of column f_out: for f_out in {3, 5, 6, 7} the
 * narrow gather wants the two gathered channels as blocks of 4 / 8 columns, [Z_L pad | Z_H pad], so that a neighbour's
 * row is a few aligned 16-byte fetches instead of 2 f_out scalar ones (the pad columns are never read into a result). */
int acm_proj_fwd_at(int64_t n_rows, int64_t f_in, int f_out, const float* X, int64_t ldx,
                    const float* w_low, const float* w_high, const float* w_mlp, int64_t ldw, int relu,
                    float* Z_lh, int64_t ld_lh, int64_t h_col, float* Z_i, int64_t ld_i, acm_stream_t stream);
int acm_proj_bwd_workspace_bytes(int64_t n_rows, int64_t f_in, int n_out, size_t* bytes);
int acm_proj_bwd(int64_t n_rows, int64_t f_in, int n_out, const float* X, int64_t ldx,
                 const float* dZ, int64_t lddz, const float* w_low, const float* w_high, const float* w_mlp, int64_t ldw,
                 float* dX, int64_t lddx, float* dW, int64_t lddw, int64_t dw_col_block, int64_t dw_block_stride,
                 void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream);

/* ------------------------------------------------ ACM-GCN++ residual branch --
 * xX = F.dropout(F.relu(self.mlpX(x)))  with mlpX = one nn.Linear  (ACM-Geometric/models.py:26-27,55-56,73;
 * ACM-Pytorch/models/models.py:50-53,150-152,165).
 *   acm_linear_fwd    Y = dropout(relu?(X W^T + b)) as ONE GEMM with bias / ReLU / counter-based dropout in its
 *                     epilogue; W in nn.Linear's layout [f_out, f_in] (pitch ldw); drop NULL or p = 0: none.
 *                     Workspace: acm_gemm_workspace_bytes(0, 1, n_rows, f_out, f_in).
 *   acm_bias_act      the same epilogue in place on a product that came out of acm_spmm_v (CSR features).
 *   acm_bias_act_bwd  G = dY * keep_scale * [Y > 0] (keep_scale = 1 / (1 - p); both masks are read off the forward's
 *                     output), d_bias = column sums of G (deferrable second phase); f <= 256.  The weight gradient is
 *                     then dW = G^T X (acm_gemm with transA, or acm_spmm_v on the transposed feature handle). */
int acm_linear_fwd(int64_t n_rows, int64_t f_in, int64_t f_out, const float* X, int64_t ldx,
                   const float* W, int64_t ldw, const float* bias, int relu, const acm_dropout_t* drop,
                   float* Y, int64_t ldy, void* workspace, size_t workspace_bytes, acm_stream_t stream);
int acm_bias_act(int64_t n_rows, int f, float* Y, int64_t ldy, const float* bias, int relu,
                 const acm_dropout_t* drop, acm_stream_t stream);
int acm_bias_act_bwd_workspace_bytes(int64_t n_rows, int f, size_t* bytes);
int acm_bias_act_bwd(int64_t n_rows, int f, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                     float keep_scale, int relu, float* G, int64_t ldg, float* d_bias,
                     void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream);
/*   acm_linear_bwd    (ABI 24) the whole backward of that Linear on a NARROW dense input that takes no gradient (f_in <= 16:
 *                     the raw features of twitch-gamer, 7 columns): dW = G^T X ([f_out, f_in], pitch lddw) and d_bias =
 *                     column sums of G in ONE pass over (Y, dY, X), G as acm_bias_act_bwd forms it but never stored --
 *                     replaces acm_bias_act_bwd + acm_gemm(transA) and the [n_rows, f_out] matrix between them.
 *                     Deterministic; both sums honour `defer`.  f_out <= 256. */
/*   acm_linear_fwd_add        Y = add + dropout(relu(X W^T + b)): the ACM-GCN++ hidden activations fea1 + xX (models.py:73) in
 *                             the Linear's own launch (Y may alias add); same shapes as acm_linear_bwd.
 *   acm_linear_bwd_recompute  its backward: Y no longer shows the masks, so they are formed again -- the pre-activation by
 *                             the forward's own fmaf chain (same bits, same sign), the dropout factors from the counter
 *                             (`drop` as in the forward, same step): the pass reads dY and X only.  Workspace:
 *                             acm_linear_bwd_workspace_bytes. */
int acm_linear_bwd_workspace_bytes(int64_t n_rows, int f_in, int f_out, size_t* bytes);
int acm_linear_bwd(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* Y, int64_t ldy,
                   const float* dY, int64_t lddy, float keep_scale, int relu, float* dW, int64_t lddw, float* d_bias,
                   void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream);
int acm_linear_fwd_add(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                       const float* bias, int relu, const acm_dropout_t* drop, const float* add, int64_t ld_add,
                       float* Y, int64_t ldy, acm_stream_t stream);
int acm_linear_bwd_recompute(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                             const float* bias, int relu, const acm_dropout_t* drop, const float* dY, int64_t lddy,
                             float* dW, int64_t lddw, float* d_bias, void* workspace, size_t workspace_bytes,
                             acm_reduce_list_t* defer, acm_stream_t stream);

/* ----------------------------------------------------------------- SpMM --
 * Y[r, 0:width] = sum_j A[r,j] * G[j, 0:width]   (plain CSR x dense; used for
 * k-hop ACM-SGC chains -- ACM-Pytorch/utils.py:631-637 -- and by tests).
 */
/* Narrow operands (width <= 8): when the rows of G are 16-byte aligned (8-byte for width <= 2) and ldg is at least the
 * next of 2 / 4 / 8 above `width`, a row is fetched as that whole block -- G must cover n_cols x ldg floats. */
int acm_spmm(const acm_csr_t* a, const float* G, int64_t ldg, int width,
             float* Y, int64_t ldy, void* workspace, size_t workspace_bytes,
             acm_stream_t stream);

/* Same product with the operator's values replaced by `vals` (device, nnz floats in the handle's
 * CSR order; NULL = the handle's own) and an optional ReLU on the result.  This is the
 * sparse-feature projection Z = X_csr [W_L|W_H|W_I] of the wide bag-of-words / one-hot inputs
 * (the reference multiplies the dense N x F_in matrix: ACM-Pytorch/utils.py:298,373-383,
 * ACM-Geometric/dataset.py:135-143), with X's *structure* in the handle and its values -- which
 * change every step under input dropout (models.py:54) -- passed per call; dWcat = X^T dZ is the
 * same call on acm_csr_transpose(X) with vals permuted by src_pos.
 */
int acm_spmm_v(const acm_csr_t* a, const float* vals, const float* G, int64_t ldg, int width,
               float* Y, int64_t ldy, int relu, void* workspace, size_t workspace_bytes,
               acm_stream_t stream);

/* The general form:  Y[r] = relu?( row_scale[r] * (A(vals) G)[r] - sub_scale[r] * SUB[r] ), every option
 * nullable (NULL scale = 1, NULL sub = no subtraction, NULL vals = the handle's own / implicit ones).
 * g_bf16: G points to a bf16 matrix (acm_cast_bf16), ldg in elements, 8 < width <= 64 even.
 * Uses: A_low X = D^-1 (P X) on a pattern-only handle (row_scale = 1/d); the structure-parameter gradient of
 * the aggregate-first form, d struc_low = A_low^T (D G_S) - G_S. */
typedef struct {
    const float* vals;
    const float* row_scale;
    const float* sub; int64_t ld_sub;
    const float* sub_scale;
    int32_t relu;
    int32_t g_bf16;
} acm_spmm_opts_t;

int acm_spmm_ex(const acm_csr_t* a, const void* G, int64_t ldg, int width, float* Y, int64_t ldy,
                const acm_spmm_opts_t* opts, void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* fp32 -> bf16 (round to nearest even) copy of a [n_rows, n_cols] matrix; dst leading dimension in elements. */
int acm_cast_bf16(int64_t n_rows, int64_t n_cols, const float* src, int64_t ld_src,
                  uint16_t* dst, int64_t ld_dst, acm_stream_t stream);

/* ---------------------------------------------------- fused ACM layer (K2) --
 * One pass over A_low computes every graph channel and the adaptive mixing:
 *
 *   P_L = A_low * ZL            P_H = A_low * ZH           [P_S = A_low * S]
 *   pre_L = P_L                 pre_H = ZH_self - P_H      [pre_S = deg * P_S - S_self]
 *   H_c  = relu(pre_c)  (variant/acmsgc: identity on L,H -- the ReLU was applied by the GEMM)
 *   H_I  = relu(ZI_self) (acmsgc: identity)
 *   att  = softmax( sigmoid([LN_c](H_c) . v_c)_c  *  M / k )          k = 3 | 4
 *   out  = scale * sum_c att_c * H_c
 *
 * Replaces G:86-116 / P:162-232 after the three torch.mm: 2-3 torch.spmm, 3-4
 * relu, 3-4 LayerNorm, 3-4 skinny mm, cat, sigmoid, mm, softmax, mul/add.
 * `A_high * Z = Z - A_low * Z` and `A * S = D (A_low S) - S` are exact
 * identities of the reference's filters (train.py:77-78).
 */
typedef struct {
    int32_t f_out;            /* F: columns per channel                                 */
    int32_t n_channels;       /* k = 3, or 4 with the structure channel                 */
    int32_t relu_after;       /* 1: ReLU on pre_L/pre_H (ACM);  0: ACMII / acmsgc       */
    int32_t relu_mlp;         /* 1: ReLU on the identity channel; 0: acmsgc             */
    int32_t layernorm;        /* 1: LayerNorm feeds the attention logits (G:59,67)      */
    float   scale;            /* 3 (G:92,108,116) or 1 with the structure channel (G:113) */
    int64_t row_offset;       /* global index of local row 0 (self rows of gathered src) */

    /* gathered sources, indexed by column id of the operator */
    const float* g_low;   int64_t ld_g_low;
    const float* g_high;  int64_t ld_g_high;
    const float* g_struc; int64_t ld_g_struc;     /* struc_low parameter, or NULL */
    /* self sources, indexed by local row */
    const float* s_high;  int64_t ld_s_high;
    const float* s_mlp;   int64_t ld_s_mlp;
    const float* s_struc; int64_t ld_s_struc;     /* or NULL */
    const float* deg;                             /* d_i = rowsum(I + A), local rows; or NULL */

    /* attention parameters */
    const float* att_vec[4];      /* v_low, v_high, v_mlp, v_struc : F floats each       */
    const float* ln_weight[4];    /* LayerNorm gamma per channel (NULL if !layernorm)    */
    const float* ln_bias[4];
    const float* att_mix;         /* k x k row-major (self.att_vec)                      */

    /* outputs */
    float* out;  int64_t ld_out;  /* [n_rows, F]                                         */
    float* pre;  int64_t ld_pre;  /* [n_rows, (k-1)*F] pre-activations, saved for backward */
    float* att;                   /* [n_rows, 4] mixing weights (self.att_low/...)        */
    /* optional fused post-op of the caller's inter-layer glue (models.py:70  dropout(relu(fea1))):
     *   out <- (post_relu ? max(out, 0) : out) * (post_scale ? post_scale[row][col] : 1)
     * post_scale is the dropout keep-mask / (1 - p); NULL = none. */
    const float* post_scale; int64_t ld_post_scale;
    int32_t post_relu;
    /* 0: g_low / g_high / g_struc are fp32 (default).  1: they point to bf16 (uint16_t) matrices produced by
     * acm_cast_bf16, leading dimensions in elements; products are accumulated in fp32.  Halves the gathered
     * bytes of the wide (F > 8) path at ~3 decimal digits of the operand (BASELINE config 3). */
    int32_t gather_bf16;
    /* optional per-row multiplier of every gathered sum (pattern-only a_low: 1/d_i); NULL = 1 */
    const float* row_scale;
    /* in-register dropout of the (post-ReLU) output, in addition to post_scale; p = 0: off */
    acm_dropout_t post_drop;
} acm_conv_fwd_t;

int acm_conv_fwd(const acm_csr_t* a_low, const acm_conv_fwd_t* p,
                 void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* The same epilogue WITHOUT a gather (ABI 25): the caller has formed the filtered channels itself -- the aggregate-first
 * form for wide inputs computes A_low (X W) as (A_low X) W (G:101-104 with the products reordered) -- and hands over
 *   pre_L = g_low[row],   pre_H = s_high[row] - g_high[row],   Z_I = s_mlp[row]
 * (three channels of 64 fp32 columns, rows 16-byte aligned, no row_scale / deg).  Everything behind that -- ReLU,
 * LayerNorm, attention head, mixing, post-op, the outputs `out`, `pre`, `att` -- is acm_conv_fwd's, bit for bit. */
int acm_conv_head_fwd(int64_t n_rows, const acm_conv_fwd_t* p, acm_stream_t stream);

/* The wide aggregate-first layer behind its gather (ABI 26): from P = A_low Xd (`agg`) and the dropped input Xd (`xs`), both
 * [n_rows, f_pad] fp32 with f_pad % 4 == 0, f_in <= f_pad <= 128 and 16-byte aligned rows whose pad columns are zero, and the
 * three [f_in, 64] weight matrices (row pitch ld_w), ONE row-local kernel computes
 *   pre_L = P W_L,   pre_H = (Xd - P) W_H,   Z_I = Xd W_I        (G:101-104 with A_low (X W) = (A_low X) W; fp32 numbers
 *                                                                  split into three bf16 each, six products kept: fp32 accuracy)
 * and acm_conv_fwd's epilogue on them.  Of `p` it reads the head (relu_after / relu_mlp / layernorm / scale, att_vec, ln_*,
 * att_mix), the post-op and the outputs `out`, `pre` = [pre_L | pre_H] and `att`; Z_I goes to `zi` (the s_mlp of the layer's
 * acm_conv_bwd_local).  f_out = 64, three channels; ACM_EUNSUPPORTED otherwise.  Replaces two acm_gemm calls and
 * acm_conv_head_fwd (and 2 x 512 bytes per row written and read back between them). */
int acm_conv_aggw_fwd(int64_t n_rows, int64_t f_in, int64_t f_pad, const float* agg, int64_t ld_agg, const float* xs,
                      int64_t ld_xs, const float* w_low, const float* w_high, const float* w_mlp, int64_t ld_w,
                      float* zi, int64_t ld_zi, const acm_conv_fwd_t* p, acm_stream_t stream);

/* ------------------------------------------- backward, row-local part (K3) --
 * From grad_out and the saved pre-activations recompute the attention head and
 * produce (a) the per-row gradients G_c = dL/d pre_c that feed the transposed
 * SpMM / the weight GEMM and (b) the reduced gradients of every attention /
 * LayerNorm parameter.  Replaces the autograd replay of G:57-75,92,106-116.
 */
typedef struct {
    int32_t f_out, n_channels, relu_after, relu_mlp, layernorm;
    float   scale;
    const float* grad_out; int64_t ld_grad_out;   /* [n_rows, F]          */
    const float* pre;      int64_t ld_pre;        /* saved by acm_conv_fwd */
    const float* s_mlp;    int64_t ld_s_mlp;      /* Z_I rows (local)      */
    const float* deg;                              /* as in forward         */
    const float* att_vec[4];
    const float* ln_weight[4];
    const float* ln_bias[4];
    const float* att_mix;
    /* outputs */
    float* g_low;   int64_t ld_g_low;     /* dL/dpre_L                         [n_rows,F] */
    float* g_high;  int64_t ld_g_high;    /* dL/dpre_H                                    */
    float* g_mlp;   int64_t ld_g_mlp;     /* dL/dZ_I  (relu mask applied)                 */
    float* g_struc; int64_t ld_g_struc;   /* deg_i * dL/dpre_S (pre-scaled for A_low^T)   */
    float* d_att_vec[4];                  /* F floats each (written, not accumulated)     */
    float* d_ln_weight[4];
    float* d_ln_bias[4];
    float* d_att_mix;                     /* k x k                                         */
    /* post-op of the forward (see acm_conv_fwd_t): grad_out is multiplied by post_scale and by
     * [out_before_post > 0] (recomputed) on load */
    const float* post_scale; int64_t ld_post_scale;
    int32_t post_relu;
    /* optional per-row multiplier of the g_low / g_high outputs (pattern-only backward: the operand of
     * A_low^T G = P (D^-1 G) is written pre-scaled by 1/d_i); NULL = 1.  `deg` NULL = 1 likewise
     * (g_struc = dL/dpre_S unscaled, which is what P needs: A_low^T (D G_S) = P G_S). */
    const float* g_scale;
    acm_dropout_t post_drop;              /* the forward's post_drop (same seed / step / tag) */
    acm_reduce_list_t* defer;             /* NULL: reduce the parameter gradients now; else append (see above) */
} acm_conv_bwd_local_t;

int acm_conv_bwd_local_workspace_bytes(int64_t n_rows, int f_out, int n_channels, size_t* bytes);
int acm_conv_bwd_local(int64_t n_rows, const acm_conv_bwd_local_t* p,
                       void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* The backward of the wide aggregate-first layer (acm_conv_aggw_fwd) in ONE kernel (ABI 27): acm_conv_bwd_local's row-local
 * head backward and the three weight gradients
 *   dW_L = P^T G_L,   dW_H = (Xd - P)^T G_H,   dW_I = Xd^T G_I          (the MmBackward of G:101-103 in aggregate-first order)
 * without [G_L | G_H | G_I] ever reaching memory: a workgroup computes G for 128 rows, leaves it in LDS as split-bf16 matrix
 * operands and contracts it with the rows of `agg` = P and `xs` = Xd loaded transposed ([n_rows, f_pad] as in the forward).
 * Of `p` it reads grad_out / pre / s_mlp, the head, the post-op (post_relu, post_drop; post_scale and g_scale must be NULL) and
 * writes d_att_vec / d_ln_* / d_att_mix; g_low / g_high / g_mlp are not touched.  d_w_*: [f_in, 64] at row pitch ld_dw.
 * All reduced outputs follow p->defer like acm_conv_bwd_local's.  f_out = 64, three channels; ACM_EUNSUPPORTED otherwise.
 * Replaces acm_conv_bwd_local + two transposed acm_gemm_blocks calls. */
int acm_conv_aggw_bwd_workspace_bytes(int64_t n_rows, int64_t f_pad, size_t* bytes);
int acm_conv_aggw_bwd(int64_t n_rows, int64_t f_in, int64_t f_pad, const float* agg, int64_t ld_agg, const float* xs,
                      int64_t ld_xs, const acm_conv_bwd_local_t* p, float* d_w_low, float* d_w_high, float* d_w_mlp,
                      int64_t ld_dw, void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* ------------------------------------- backward, transposed SpMM part (K4) --
 *   dZ_L = mask_L * (A_low^T G_L)
 *   dZ_H = mask_H * (G_H_self - A_low^T G_H)
 *   dS   = A_low^T (D G_S) - G_S_self                      (structure channel)
 * mask_c = [Z_c > 0] for ACMII (ReLU sits before the filter), 1 otherwise.
 * `a_low_t` is the transposed operator (acm_csr_transpose).  Replaces
 * SparseAddmmBackward x2-3 (autograd of G:87-88,96-103,111).
 */
typedef struct {
    int32_t f_out;
    int64_t row_offset;
    const float* g_low;   int64_t ld_g_low;    /* gathered, indexed by column id */
    const float* g_high;  int64_t ld_g_high;
    const float* g_struc; int64_t ld_g_struc;  /* deg-scaled, or NULL            */
    const float* s_high;  int64_t ld_s_high;   /* G_H rows of the local nodes    */
    const float* s_struc; int64_t ld_s_struc;  /* deg-scaled G_S, local rows     */
    const float* inv_deg;                       /* 1/d_i local rows (with s_struc) */
    const float* mask_low;  int64_t ld_mask_low;   /* Z_L (post-ReLU) or NULL    */
    const float* mask_high; int64_t ld_mask_high;
    float* dz_low;   int64_t ld_dz_low;
    float* dz_high;  int64_t ld_dz_high;
    float* d_struc;  int64_t ld_d_struc;       /* grad of struc_low rows, or NULL */
    /* optional per-row multiplier of the self term s_high (pattern-only backward: s_high holds D^-1 G_H, so
     * self_scale = d_i restores G_H); NULL = 1.  inv_deg NULL = 1. */
    const float* self_scale;
    /* gather_bf16 != 0 (ABI 20): g_low / g_high / g_struc point to bf16 tables (acm_cast_bf16; leading dimensions in
     * elements, even), 8 < f_out <= 64 even; the self terms s_high / s_struc stay fp32, sums accumulate in fp32.  Half the
     * gathered bytes of the fabric-bound wide transposed products (BASELINE config 3's tolerance; fp32 is the default). */
    int32_t gather_bf16;
} acm_conv_bwd_spmm_t;

int acm_conv_bwd_spmm(const acm_csr_t* a_low_t, const acm_conv_bwd_spmm_t* p,
                      void* workspace, size_t workspace_bytes, acm_stream_t stream);


/* ------------------------------ aggregate-first form of the layer (K2a / K3a) --
 * For the ACM (non-variant) and acmsgc models the filter and the projection
 * commute: A_low (X W) = (A_low X) W.  When F_in < F_out (twitch-gamer: 7 vs 64)
 * gathering X rows moves F_in floats per edge instead of 2 F_out:
 *
 *   P     = A_low X                       (one narrow gather, shared by both graph channels)
 *   pre_L = P W_L      pre_H = (X - P) W_H      Z_I = X W_I     (in registers, per row)
 *   ... then the same ReLU / LayerNorm / mixing head as acm_conv_fwd.
 *
 * Same reference call sites as acm_gemm + acm_conv_fwd (G:87-116);
 * equal to them up to fp32 re-association.  f_in <= 16, F <= 64, and only when
 * the layer input needs no gradient (first layer): the backward then needs no
 * transposed SpMM for the low/high channels, dW_L = P^T G_L, dW_H = (X - P)^T G_H,
 * dW_I = X^T G_I are row-local reductions (acm_conv_agg_bwd).
 * With the structure channel (n_channels = 4) the parameter S is still gathered
 * F-wide (PS = A_low S, pre_S = deg * PS - S_self) -- 72 floats per edge instead of
 * 192 -- and its gradient needs one F-wide transposed product (acm_spmm_ex).
 */
typedef struct {
    int32_t f_in, f_pad;       /* f_pad = row length (floats) of xg / xs / agg: 4, 8 or 16, >= f_in; padding is zero */
    int32_t f_out, relu_after, relu_mlp, layernorm;
    float   scale;
    const float* xg; int64_t ld_xg;    /* X rows indexed by column id (16-byte aligned rows)  */
    const float* xs; int64_t ld_xs;    /* X rows of the local nodes                            */
    const float* w_low; const float* w_high; const float* w_mlp; int64_t ld_w;   /* [f_in, F] each */
    const float* att_vec[4];
    const float* ln_weight[4];
    const float* ln_bias[4];
    const float* att_mix;              /* 3 x 3 */
    float* out; int64_t ld_out;        /* [n_rows, F]                                          */
    float* agg; int64_t ld_agg;        /* [n_rows, f_pad]  P = A_low X, saved for backward     */
    float* att;                        /* [n_rows, 4]                                          */
    const float* post_scale; int64_t ld_post_scale;   /* fused post-op, as in acm_conv_fwd_t   */
    int32_t post_relu;
    /* structure channel (n_channels = 4; att_mix is then 4 x 4, scale 1) */
    int32_t n_channels;                /* 3 or 4                                               */
    const void*  sg; int64_t ld_sg;    /* struc_low rows indexed by column id: fp32, or bf16 (acm_cast_bf16) if sg_bf16 */
    int32_t sg_bf16;
    const float* ss; int64_t ld_ss;    /* struc_low rows of the local nodes (fp32)             */
    const float* deg;                  /* d_i = rowsum(I + A), local rows                      */
    float* ps; int64_t ld_ps;          /* [n_rows, F]  A_low * S, saved for backward           */
    const float* row_scale;            /* optional per-row multiplier of both gathers (pattern-only a_low: 1/d_i) */
    acm_dropout_t post_drop;           /* in-register dropout of the output (see acm_conv_fwd_t) */
    /* optional output [n_rows, 4 * n_channels] (16-byte aligned rows): the row's head statistics
     * mean_c | rstd_c | sigmoid_c | alpha_c as the forward computed them.  Handed to acm_conv_agg_bwd they save it
     * the recomputation (three 16-lane reductions per channel and row: ~12 % of that kernel); bit-identical results. */
    float* head_stats; int64_t ld_head_stats;
    /* Optional: the NEXT layer's narrow projection (layers.py:87-89 of the following GraphConvolution), fused into this
     * layer's epilogue while the finished row (after the post-op) is still in registers:
     *   next_zlh[row] = [out_row W_L' | out_row W_H'],  next_zi[row] = out_row W_I'   (ReLU'd when next_relu is set)
     * -- what acm_proj_fwd would compute from `out` in a launch of its own.  next_f = F' <= 2 (0 = off);
     * the three weights are f_out x F' with leading dimension next_ld_w.                                             */
    const float* next_w_low; const float* next_w_high; const float* next_w_mlp; int64_t next_ld_w;
    int32_t next_f, next_relu;
    float* next_zlh; int64_t ld_next_zlh;
    float* next_zi;  int64_t ld_next_zi;
    /* agg_given != 0: `agg` already holds A_low * xg -- an earlier call of this operator on the same input wrote it
     * (e.g. every evaluation pass over a static feature matrix after the first; a training loop's input pipeline) -- so
     * that gather is skipped (the structure channel's A_low * S still runs) and xg is not read.                      */
    int32_t agg_given;
    int32_t reserved0;                 /* zero (ABI <= 21: use_streams, the streamed form of the fused forward -- measured
                                          slower than the CSR walk, DESIGN.md section 4, and removed in ABI 22)      */
    /* With agg_given: the row-local stage also stores the rows of `agg` and `xs` it read to agg_copy / xs_copy (NULL:
     * off; rows of f_pad floats) -- the operands of this layer's backward when `agg` / `xs` themselves are the buffers
     * an input pipeline refills before that backward runs (acm_conv_agg_bwd_t.next_agg).                             */
    float* agg_copy; int64_t ld_agg_copy;
    float* xs_copy;  int64_t ld_xs_copy;
    /* With agg_copy / xs_copy (ABI 22): the row-local stage REFILLS `xs` in place for the next training step,
     *     xs[row, :f_in] <- next_x[row, :f_in] * mask(next_drop),   xs[row, f_in:f_pad] <- 0
     * -- what acm_dropout(next_x -> xs, f_pad columns, next_drop) would do in a launch of its own (a lane overwrites exactly
     * the elements it has just read and copied to xs_copy).  next_x = NULL: off.  next_drop.step_offset = 1 draws the mask of
     * the step after this one.  The input pipeline's hand-over (acm_conv_agg_bwd_t.next_agg gathers from the refilled xs)
     * without its dropout launch.                                                                                     */
    const float* next_x; int64_t ld_next_x;
    acm_dropout_t next_drop;
} acm_conv_agg_fwd_t;

int acm_conv_agg_fwd(const acm_csr_t* a_low, const acm_conv_agg_fwd_t* p,
                     void* workspace, size_t workspace_bytes, acm_stream_t stream);

typedef struct {
    int32_t f_in, f_pad, f_out, relu_after, relu_mlp, layernorm;
    float   scale;
    const float* grad_out; int64_t ld_grad_out;
    const float* agg; int64_t ld_agg;
    const float* xs;  int64_t ld_xs;
    const float* w_low; const float* w_high; const float* w_mlp; int64_t ld_w;
    const float* att_vec[4];
    const float* ln_weight[4];
    const float* ln_bias[4];
    const float* att_mix;
    /* output: one flat vector (k = n_channels)
     *   [ dW_low : f_in x F ][ dW_high ][ dW_mlp ][ d att_vec : k x F ][ d ln_weight : k x F ][ d ln_bias : k x F ][ d att_mix : k x k ] */
    float* d_params;
    const float* post_scale; int64_t ld_post_scale;   /* post-op of the forward                */
    int32_t post_relu;
    int32_t n_channels;                /* 3 or 4                                               */
    const float* ps; int64_t ld_ps;    /* A_low * S saved by the forward                       */
    const float* ss; int64_t ld_ss;    /* struc_low rows (local)                               */
    const float* deg;
    float* g_struc; int64_t ld_g_struc;   /* out: g_struc_scale_i * dL/dpre_S  (the operand of the A_low^T product) */
    const float* g_struc_scale;           /* deg for an explicit A_low^T; NULL (= 1) for the pattern-only form  */
    acm_dropout_t post_drop;
    acm_reduce_list_t* defer;             /* NULL: reduce d_params now; else append to the list                 */
    const float* head_stats; int64_t ld_head_stats;   /* the forward's head_stats, or NULL: recompute            */
    const float* out; int64_t ld_out;     /* the forward's `out`, or NULL.  With post_relu set and no post_scale the fused
                                           * post-op is undone by reading it -- out != 0 <=> the ReLU passed AND the dropout
                                           * kept the element (relu'(0) = 0 as in torch) -- instead of recomputing the mixed
                                           * row and regenerating the Philox mask (an eighth of the kernel's VALU work)  */
    /* The NEXT step's input aggregation, carried by this launch (NULL next_agg: off).  This kernel is bound by its vector
     * instructions and leaves the memory system idle; the narrow gather  next_agg = next_row_scale * (A_low next_xg)  of the
     * next training step's first layer is bound by memory latency and depends on nothing this step computes (the input
     * features are constant, the dropout mask a function of the step counter: acm_dropout_t.step_offset = 1).  With
     * next_agg set, every workgroup is eight waves: four run the sixteen-rows-per-wave backward (acm_conv_agg16.hip),
     * four walk next_a's id streams (acm_csr_build_streams; the grid becomes stream_waves / 4 workgroups, one per CU);
     * the next forward then runs with acm_conv_agg_fwd_t.agg_given.  Needs: the shapes of the sixteen-rows-per-wave
     * backward (f_out = 64, head_stats, `out` behind a fused ReLU or no post-op; acm_tuning_t.rows16 bit 2),
     * ld_next_xg = f_pad, a pattern-only next_a with streams built for a multiple of four waves <= 1024 (and
     * <= n_rows / 4); ACM_EUNSUPPORTED otherwise.  next_agg / next_xg must not alias agg / xs.  May be combined
     * with proj_* below (one kernel then carries all three: bench.py's dominant launch).                              */
    const acm_csr_t* next_a;
    const float* next_xg; int64_t ld_next_xg;
    const float* next_row_scale;
    float* next_agg; int64_t ld_next_agg;
    /* Optional (ABI 20): the backward of the FOLLOWING layer's narrow projection  Z' = out [W_L' | W_H' | W_I']
     * (ACM-Geometric/layers.py:87-89 of the next GraphConvolution; what acm_proj_bwd computes in a launch of its own)
     * inside this kernel.  With proj_dz set, grad_out is NOT read: the gradient of this layer's output is formed per row,
     *     grad_out[row, :] = proj_dz[row, :] [W_L' | W_H' | W_I']^T        proj_dz = dL/dZ', [n_rows, 3 proj_f],
     * and the following layer's weight gradients
     *     proj_d_w[c][col][q] = sum_rows out[row, col] proj_dz[row, c proj_f + q]     (three f_out x proj_f blocks)
     * are reduced with d_params (same `defer`).  That removes the [n_rows, f_out] gradient from memory altogether
     * (43 MB written and read back on the twitch-shaped graph) and one launch.  Needs `out` (= the following layer's
     * input), the shapes of the sixteen-rows-per-wave backward (f_out = 64, head_stats) and proj_f <= 2; with or
     * without next_agg.  ACM_EUNSUPPORTED otherwise -- the caller then runs acm_proj_bwd itself and calls again with
     * grad_out. */
    const float* proj_dz; int64_t ld_proj_dz;
    const float* proj_w_low; const float* proj_w_high; const float* proj_w_mlp; int64_t proj_ld_w;
    int32_t proj_f;
    float* proj_d_w;
} acm_conv_agg_bwd_t;

int acm_conv_agg_bwd_workspace_bytes(int64_t n_rows, int f_in, int f_out, size_t* bytes);
int acm_conv_agg_bwd(int64_t n_rows, const acm_conv_agg_bwd_t* p,
                     void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* ------------------------------- ACMII first layer: recompute on gather (K1 + K2) --
 * ACMII (ACM-Geometric/layers.py:94-99; the default of ACM-Geometric, parse.py:57) applies the ReLU between the
 * projection and the filter, H_L = A_low relu(X W_L), H_H = (I - A_low) relu(X W_H), so the aggregate-first rewrite
 * does not apply and acm_gemm + acm_conv_fwd gather 2 F projected floats per edge.  When the layer input is narrow
 * (f_in <= 8, f_out = 64: the first layer on twitch-gamer) this entry point gathers the neighbour's INPUT row
 * (f_pad = 8 floats, a table that fits the L2) and recomputes relu(x_j [W_L | W_H]) per edge on the matrix pipe
 * (v_mfma_f32_16x16x4_f32, 16 neighbours per tile): same outputs as the two calls it replaces -- out, pre = [pre_L |
 * pre_H] and att as acm_conv_fwd, zlh = relu(X [W_L | W_H]) and zi = relu(X W_I) as the GEMM (K4's masks / self rows,
 * K3's s_mlp) -- equal to them up to fp32 re-association.  Values of an explicit operator must be
 * non-negative (relu(a z) = a relu(z)).  Workspace: acm_conv_acmii_fwd_workspace_bytes (partial sums of the long rows'
 * pieces, added in slot order by a second launch; never zero bytes).  The backward is acm_conv_bwd_local /
 * acm_conv_bwd_spmm / acm_gemm as for the literal form. */
typedef struct {
    int32_t f_in, f_pad, f_out;        /* f_pad = 8, f_out = 64                                                */
    int32_t layernorm;
    float   scale;
    const float* xg; int64_t ld_xg;    /* input rows indexed by column id, f_pad long, zero padded, 8-byte aligned */
    const float* xs; int64_t ld_xs;    /* input rows of the local nodes                                          */
    const float* w_low; const float* w_high; const float* w_mlp; int64_t ld_w;   /* [f_in, f_out] each          */
    const float* att_vec[4];
    const float* ln_weight[4];
    const float* ln_bias[4];
    const float* att_mix;              /* 3 x 3 */
    float* out; int64_t ld_out;        /* [n_rows, f_out]                                                        */
    float* pre; int64_t ld_pre;        /* [n_rows, 2 f_out]                                                      */
    float* att;                        /* [n_rows, 4]                                                            */
    float* zlh; int64_t ld_zlh;        /* [n_rows, 2 f_out]  relu(X [W_L | W_H])                                 */
    float* zi;  int64_t ld_zi;         /* [n_rows, f_out]    relu(X W_I)                                         */
    const float* post_scale; int64_t ld_post_scale;   /* fused post-op, as in acm_conv_fwd_t                    */
    int32_t post_relu;
    const float* row_scale;            /* pattern-only a_low: 1 / d_i                                            */
    acm_dropout_t post_drop;
    /* structure channel (n_channels = 4; att_mix 4 x 4, scale 1, pre is [n_rows, 3 f_out]): the caller gathers the
     * parameter first, ps = A_low S (acm_spmm_ex), as for acm_conv_agg_fwd; H_S = relu(deg * ps - ss) */
    int32_t n_channels;                /* 3 or 4                                                                 */
    const float* ps; int64_t ld_ps;    /* [n_rows, f_out]  A_low * S                                             */
    const float* ss; int64_t ld_ss;    /* struc_low rows of the local nodes                                      */
    const float* deg;                  /* d_i = rowsum(I + A)                                                    */
} acm_conv_acmii_fwd_t;

int acm_conv_acmii_fwd_workspace_bytes(const acm_csr_t* a_low, size_t* bytes);
int acm_conv_acmii_fwd(const acm_csr_t* a_low, const acm_conv_acmii_fwd_t* p,
                       void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* ------------------------ ACMII first layer on the bf16 matrix pipe: the mask form (ABI 24) --
 * relu(x_j W[:, c]) = m_j[c] * (x_j W[:, c]) with the mask m_j[c] = [x_j W[:, c] > 0], so
 *     sum_j a_ij relu(x_j W[:, c]) = sum_f W[f, c] V_i[c, f],    V_i[c, f] = sum_j a_ij m_j[c] x_j[f]:
 * V_i is a matrix product over the neighbour index with one 0 / 1 operand (exact in bf16) and one fp32 operand that is
 * exactly the sum of three bf16 numbers -- v_mfma_f32_16x16x32_bf16 at fp32 accuracy, sixteen times the rate of the fp32
 * matrix pipe acm_conv_acmii_fwd uses.  The same V gives the layer's weight gradients without any transposed product
 * (the layer input takes no gradient):  dW_L[f, c] = sum_i dH_L[i, c] rs_i V^L_i[c, f],
 *     dW_H[f, c] = sum_i dH_H[i, c] (m^H_i[c] x_i[f] - rs_i V^H_i[c, f])      (rs_i = 1 / d_i, pattern-only operator).
 * Replaces, for ACM-Geometric/layers.py:94-99 with f_in <= 8 and f_out = 64: torch.mm + F.relu + torch.spmm (x2) of the
 * forward, and SpmmBackward (x2) + ReluBackward + MmBackward of weight_low / weight_high.
 *
 *   acm_acmii_table        one 64-byte row per node, [x hi | x mid | x lo / 2 (8 bf16 each) | 16 mask bytes], plus an all-zero
 *                          row n_rows.  Rebuilt whenever x (input dropout) or the weights change: every training step.
 *                          x: [n_rows, ld_x >= 8], zero padded beyond f_in, 8-byte aligned rows.
 *   acm_conv_acmii_v_fwd   acm_conv_acmii_fwd's outputs from the table (p->xg is not read; p->row_scale is required;
 *                          p->zlh may be NULL when the operator has no long rows -- only its second half, the rows' own
 *                          relu(x_i W_H), is written).  Pattern-only operators whose column ids index the table (n_cols + 1
 *                          rows); a row block of a row-sharded operator qualifies (xs = its own rows, table over all columns).
 *                          Workspace: acm_conv_acmii_fwd_workspace_bytes.
 *   acm_conv_acmii_v_bwd   d_w_low, d_w_high ([f_in, 64], pitch ld_dw) from g_low = dH_L, g_high = dH_H ([n_rows, 64]) over
 *                          the SAME (forward) operator and table, and d_w_mlp = X^T dZ_I on the way (row-local: the launch
 *                          has every row in hand); deterministic; the final sums honour `defer`.
 * ACM_EUNSUPPORTED for operators with explicit values and f_in > 8: the caller keeps
 * acm_conv_acmii_fwd / acm_conv_bwd_spmm / acm_gemm. */
typedef struct {
    int32_t f_in;
    const void* table;                          /* acm_acmii_table's output for this step's x and weights           */
    const float* g_low;  int64_t ld_g_low;      /* dH_L [n_rows, 64]                                                 */
    const float* g_high; int64_t ld_g_high;     /* dH_H [n_rows, 64]                                                 */
    const float* g_mlp;  int64_t ld_g_mlp;      /* dZ_I [n_rows, 64] = dH_I masked by the identity channel's ReLU    */
    const float* x; int64_t ld_x;               /* the operator's OWN rows of the table's source ([n_rows, >= 8], zero padded) */
    int64_t self_offset;                        /* table row of the operator's row r = r + self_offset (0 unless the operator is a
                                                   row block of a larger one: rank * longest block of a row-sharded run)   */
    const float* row_scale;                     /* 1 / d_i                                                           */
    float* d_w_low; float* d_w_high; float* d_w_mlp; int64_t ld_dw;   /* [f_in, 64] each; d_w_mlp = X^T dZ_I        */
    acm_reduce_list_t* defer;                   /* NULL: reduce now; else append the two second phases              */
} acm_conv_acmii_bwd_t;

int acm_acmii_table_bytes(int64_t n_rows, size_t* bytes);
int acm_acmii_table(int64_t n_rows, int f_in, const float* x, int64_t ld_x, const float* w_low, const float* w_high,
                    int64_t ld_w, void* table, size_t table_bytes, acm_stream_t stream);
int acm_conv_acmii_v_fwd(const acm_csr_t* a_low, const acm_conv_acmii_fwd_t* p, const void* table,
                         void* workspace, size_t workspace_bytes, acm_stream_t stream);
int acm_conv_acmii_v_bwd_workspace_bytes(const acm_csr_t* a_low, size_t* bytes);
int acm_conv_acmii_v_bwd(const acm_csr_t* a_low, const acm_conv_acmii_bwd_t* p,
                         void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* ------------------------------------------------ fused step tail (loss) --
 *   loss   = sum_i w_i * (logsumexp(z_i) - z_i[y_i])
 *   dz_i   = w_i * (softmax(z_i) - onehot(y_i))
 * in one pass over the logits.  With w_i = 1/|train| on the training rows and 0 elsewhere this is
 * F.log_softmax(out, 1) + nn.NLLLoss()(out[train_idx], label[train_idx]) and its autograd
 * (ACM-Geometric/train.py:133-135, ACM-Pytorch/utils.py:566-571) without the index / index_put
 * kernels.  n_classes <= 64.  labels: int64.  Deterministic (fixed reduction tree).
 * Workspace: acm_nll_loss_workspace_bytes.
 */
int acm_nll_loss_workspace_bytes(int64_t n_rows, size_t* bytes);
int acm_nll_loss(int64_t n_rows, int n_classes, const float* logits, int64_t ld_logits,
                 const int64_t* labels, const float* row_weight,
                 float* loss, float* dlogits, int64_t ld_dlogits,
                 void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream);

/* ------------------------------------------------ evaluation metrics (ABI 28) --
 * The per-epoch evaluation of the reference's loops -- eval-mode logits -> accuracy on each index set and the NLL on the
 * validation set (ACM-Geometric/train.py:138-140 + data_utils.py:153-168: eval_acc over train / valid / test;
 * ACM-Pytorch/train.py:112-139: log_softmax, criterion(output[idx_val], labels[idx_val]), accuracy on idx_val / idx_test) --
 * as ONE launch over the logits instead of argmax / compare / log_softmax / gather / index / mean kernels:
 *     out[s]      = sum_i weights[s][i] * [argmax_c z_i[c] == y_i]          s = 0 .. n_sets - 1
 *     out[n_sets] = sum_i weights[loss_set][i] * (logsumexp(z_i) - z_i[y_i])
 * weights[s] = 1 / |set s| on the set's rows and 0 elsewhere (rows outside every set may carry the label -1 the
 * reference uses for "unlabeled": a zero-weight row is never looked up).  argmax takes the FIRST maximum, like torch.
 * Deterministic: per-block partial sums in `workspace` (fixed tree), the block that arrives last adds them in block
 * order; the arrival counter (the last 4 bytes of the workspace, zero before the first call) resets itself.
 * n_classes <= 64, n_sets <= 8.  Workspace: acm_eval_metrics_workspace_bytes (zero-initialised once by the caller).
 */
int acm_eval_metrics_workspace_bytes(int64_t n_rows, int n_sets, size_t* bytes);
int acm_eval_metrics(int64_t n_rows, int n_classes, const float* logits, int64_t ld_logits, const int64_t* labels,
                     const float* weights, int64_t ld_weights, int n_sets, int loss_set, float* out,
                     void* workspace, size_t workspace_bytes, acm_stream_t stream);

/* ---------------------------------------- output layer + loss + K3, fused --
 * For a narrow OUTPUT layer (f_out = n_classes <= 8, three channels, no post-op) whose result goes straight into the
 * masked NLL, the three row-local passes that follow its gather -- the head (acm_conv_fwd's row phase), acm_nll_loss
 * and acm_conv_bwd_local -- read and write the same few bytes per row: acm_conv_fwd_tail runs the gather and then ONE
 * row kernel that does all three (two launches less per step).  `fwd`, `loss` and `bwd` are exactly the arguments the
 * three separate calls would get, with bwd->grad_out == loss->dlogits and bwd->pre == fwd->pre; every output of the
 * three calls is produced (logits, pre, att, loss, dlogits, G tables, parameter-gradient sums).  The second phases
 * (loss sum, parameter-gradient sums) honour bwd->defer.  Returns ACM_EUNSUPPORTED when the layer does not qualify;
 * the caller then makes the three calls.
 * Replaces: ACM-Geometric/layers.py:57-63,86-116 (output layer) + train.py:133-135 in one pass.
 */
typedef struct {
    int32_t n_classes;
    const int64_t* labels;
    const float*   row_weight;
    float* loss;
    float* dlogits; int64_t ld_dlogits;
} acm_loss_t;

int acm_conv_fwd_tail_workspace_bytes(int64_t n_rows, int f_out, int n_channels, size_t* bytes);
int acm_conv_fwd_tail(const acm_csr_t* a_low, const acm_conv_fwd_t* fwd, const acm_loss_t* loss,
                      const acm_conv_bwd_local_t* bwd, void* workspace, size_t workspace_bytes,
                      void* tail_workspace, size_t tail_workspace_bytes, acm_stream_t stream);

/* ------------------------------------------------ fused optimizer update --
 * Adam / AdamW over a list of fp32 parameter tensors in one launch per 40 tensors: the update of
 * torch.optim.Adam / AdamW that closes the reference's training step (ACM-Geometric/train.py:113-119,137;
 * ACM-Pytorch/train.py:70-84, utils.py:572), same formulas in the same order:
 *     step += 1
 *     decoupled (AdamW): p *= 1 - lr * wd          else (Adam): g += wd * p
 *     m += (1 - beta1) * (g - m);   v = beta2 * v + (1 - beta2) * g * g
 *     p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `tensors` is a HOST array (its contents are copied into the kernel arguments, so a captured launch
 * replays without touching host memory); every pointer inside is device memory, `step` is one fp32
 * scalar per tensor (torch's capturable layout), incremented on the device.  amsgrad / maximize are not
 * provided (the reference never enables them).
 */
typedef struct {
    float*       param;
    const float* grad;
    float*       exp_avg;
    float*       exp_avg_sq;
    float*       step;
    int64_t      numel;
} acm_adam_tensor_t;

typedef struct {
    double  lr, beta1, beta2, eps, weight_decay;
    int32_t decoupled;                 /* 1 = AdamW, 0 = Adam (L2 added to the gradient) */
    int64_t* also_advance;             /* optional device counter incremented by 1 with the step counters
                                          (the acm_dropout_t.step of the model being trained) */
    int32_t* arrive;                   /* optional device int32, zero before the first call and left zero: with it the
                                          counters are advanced by the last block of the update launch itself instead
                                          of a second launch (calls sharing one `arrive` must be stream-ordered) */
    acm_reduce_list_t* pending;        /* optional (ABI 23): the step's deferred second phases.  The call flushes them --
                                          inside the update launch itself where it can: the reducing blocks lead the grid,
                                          a sum that is an element of a tensor's `grad` is stored AND applied at once (same
                                          formulas, same values as flush-then-update: bit-identical), the other tensors are
                                          updated by the blocks behind.  Needs `arrive`, <= 40 tensors, <= 28 segments and
                                          every `grad` written by the segments entirely or not at all; otherwise the call
                                          runs acm_reduce_flush first and then the plain update.  The list is empty afterwards */
} acm_adam_config_t;

int acm_adam_step(int32_t n_tensors, const acm_adam_tensor_t* tensors, const acm_adam_config_t* cfg,
                  acm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Small graphs (round 5, ABI 25): the WHOLE training step of the two-layer ACM model in six launches.
 *
 * Replaces, for graphs whose every table fits the caches (Cora / Chameleon / Squirrel: BASELINE configs 1-3;
 * ACM-Pytorch/train.py:95-139 runs them for thousands of epochs x 10 splits), the ~250 ATen launches of one step of
 * ACM-Pytorch/models/models.py:100-166 + utils.py:547-574 (model forward, log_softmax + nll_loss, backward, Adam) --
 * and the 17-18 launches the general path of this library needs -- by one launch per DEPENDENCY LEVEL of the step:
 *
 *   1  Z1 = drop(X) [W_L | W_H | W_I]                                  (CSR features; rows of W gathered by feature id)
 *   2  layer 1: gather [Z_L | Z_H | S] over the operator, head, mix, ReLU + dropout -> H; Z2 = H [W2_L | W2_H | W2_I]
 *   3  layer 2: gather [Z2_L | Z2_H | S2], head -> logits; masked NLL; the layer's own row-local backward
 *   4  layer 2: gather [G_L | G_H | G_S] (transposed operator = the same pattern) -> dZ2, dS2 (+ update of S2);
 *      dH = dZ2 W2^T; dW2 partial sums; layer 1's row-local backward
 *   5  layer 1: gather [G_L | G_H | G_S] -> dZ1, dS (+ update of S)
 *   6  dW1 = drop(X)^T dZ1 by feature row + update of W1; every other parameter-gradient sum + its update; the loss;
 *      the step counters
 *
 * Rows are never split over launches: a row longer than the handle's chunk is cut into pieces whose partial sums meet
 * in a slot buffer, and the piece that arrives last finishes the row (fixed order of additions: deterministic, no float
 * atomics).  Hidden width 64, at most 8 classes, pattern-only operator (acm_csr_create with vals = NULL) + row scale.
 * With `update` the Adam / AdamW update of every parameter is applied where its gradient becomes final (same formulas
 * as acm_adam_step); without it the gradients are written to t[..].grad and nothing is updated.
 * `train = 0`: launches 1-3 only, forward (evaluation pass: logits, att).  Dense features: convert once (CSR copy).
 */
#define ACM_SMALL_ROLES 17
enum {
    ACM_SR_W_LOW = 0, ACM_SR_W_HIGH, ACM_SR_W_MLP,                         /* [f_in, F] row-major                  */
    ACM_SR_V_LOW, ACM_SR_V_HIGH, ACM_SR_V_MLP, ACM_SR_V_STRUC,              /* att_vec_*: F floats                  */
    ACM_SR_LNW_LOW, ACM_SR_LNW_HIGH, ACM_SR_LNW_MLP, ACM_SR_LNW_STRUC,      /* LayerNorm gamma                      */
    ACM_SR_LNB_LOW, ACM_SR_LNB_HIGH, ACM_SR_LNB_MLP, ACM_SR_LNB_STRUC,      /* LayerNorm beta                       */
    ACM_SR_MIX,                                                             /* att_vec: k x k row-major             */
    ACM_SR_STRUC                                                            /* struc_low: [n, F]                    */
};

typedef struct {
    int32_t n_classes;          /* C <= 8: width of layer 2 (layer 1 is f_in -> 64)                                  */
    int32_t n_channels;         /* k = 3, or 4 with the structure channel                                            */
    int32_t relu_before;        /* 1: ACMII (`variant`): ReLU between projection and filter; 0: ACM, ReLU after      */
    int32_t layernorm;          /* 1: LayerNorm feeds the attention logits (ACM-Geometric dialect)                   */
    float   scale;              /* 3, or 1 with the structure channel                                                */
    int32_t train;              /* 1: the step; 0: forward only (launches 1-3)                                       */
    int32_t update;             /* 1: apply Adam / AdamW in place; 0: write gradients to t[..].grad                   */
    int32_t reserved0;
    /* t[layer][role]: param NULL = the role is absent / takes no gradient.  exp_avg / exp_avg_sq / step as in
     * acm_adam_tensor_t (needed with `update`); grad: optional output (required without `update`).  numel is ignored. */
    acm_adam_tensor_t t[2][ACM_SMALL_ROLES];
    /* first-layer input: CSR features (the handles passed to the call) with this call's values                          */
    const float* x_vals;        /* nnz(X) values in the order of the x handle                                        */
    const int32_t* xt_src_pos;  /* nnz(X): position in x_vals of every entry of the TRANSPOSED handle                */
    const float* xt_vals;       /* optional: x_vals in the transposed handle's order (x_vals[xt_src_pos[k]]), made once     */
    int32_t f_in;
    int32_t reserved1;
    acm_dropout_t drop_in;      /* on x_vals: element (position, 0)                                                  */
    acm_dropout_t drop_hidden;  /* on the hidden activations: element (row, column)                                  */
    const float* row_scale;     /* 1 / d_i                                                                           */
    const int64_t* labels;      /* [n]                                                                               */
    const float* row_weight;    /* [n]: 1 / |train| on training rows, else 0                                         */
    float* loss;                /* device scalar                                                                     */
    float* logits;              /* [n, C] contiguous                                                                 */
    float* att1; float* att2;   /* [n, 4] mixing weights of the two layers                                           */
    double  lr, beta1, beta2, eps, weight_decay;
    int32_t decoupled;
    int64_t* also_advance;      /* optional device counter incremented with the step counters (dropout step)         */
    int32_t* arrive;            /* reserved (not read: the last launch needs no arrival barrier)                     */
    void*   workspace;          /* acm_small_step_workspace_bytes; ZERO-FILLED by the caller once, kept between calls */
    size_t  workspace_bytes;
} acm_small_step_t;

int acm_small_step_workspace_bytes(const acm_csr_t* a_low, const acm_csr_t* x, const acm_csr_t* x_t, size_t* bytes);
/* x / x_t: CSR handles of the features and of their transpose (pattern; values come from p->x_vals; x_t may be NULL with
 * train = 0).  Errors: ACM_EUNSUPPORTED outside the envelope (explicit values, > 8 classes, > 16384 rows, ...). */
int acm_small_step(const acm_csr_t* a_low, const acm_csr_t* x, const acm_csr_t* x_t, const acm_small_step_t* p,
                   acm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ACM_HIP_H */
