/*
 * ogl_hip.h — C-ABI of libogl_hip.so: the MI355X (gfx950) streaming-GraphSAGE update path.
 *
 * The reference (MassimoPerini/online-gnn-learning, "R/" = its repo root) has no FFI: its hot path
 * is Python calling DGL / ATen.  Each entry point below replaces one of those call sites; the
 * reference-side binding a maintainer would add is the ctypes stub in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns an int status: 0 = OGL_OK, negative = OGL_E*; nothing throws;
 *  - all data pointers are CALLER-OWNED DEVICE buffers sized by the caller; the library never
 *    allocates on the hot path (the only allocations are inside ogl_graph_create);
 *  - matrices are row-major float32 with an explicit leading dimension `ld*` in ELEMENTS
 *    (ld >= logical width; pad columns are don't-care on input and left untouched on output);
 *  - vertex ids are int64 at the boundary (DGL / torch.LongTensor convention), block-local
 *    indices are int32, "no neighbour" is -1;
 *  - the last argument is the hipStream_t (passed as void*) the work is enqueued on; calls are
 *    asynchronous and re-entrant per stream; no call synchronises the device;
 *  - workspaces are explicit: query the size, allocate it yourself, pass it in.
 */
#ifndef OGL_HIP_H
#define OGL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGL_VERSION 100

enum {
  OGL_OK = 0,
  OGL_EINVAL = -1,     /* bad argument (null pointer, negative size, misaligned ld, id out of range) */
  OGL_ENOMEM = -2,     /* hipMalloc failed inside ogl_graph_create */
  OGL_EHIP = -3,       /* a HIP runtime call / kernel launch failed; see ogl_last_hip_error() */
  OGL_EWORKSPACE = -4  /* workspace pointer null or smaller than the queried size */
};

enum { OGL_REDUCE_MEAN = 0, OGL_REDUCE_MAX = 1, OGL_REDUCE_SUM = 2 };

/* Arithmetic of the dense projections (process-wide switch, default OGL_GEMM_F32):
 *   OGL_GEMM_F32    v_mfma_f32_32x32x2_f32 — bit-for-bit an fp32 fma chain;
 *   OGL_GEMM_BF16X6 every fp32 operand split exactly into 3 bf16 terms, the 6 leading cross products on
 *                   v_mfma_f32_32x32x16_bf16 with fp32 accumulation — fp32-GEMM accuracy (dropped terms
 *                   <= 2^-23 |a*b|), same tolerances in the tests; 1.2-1.4x faster where an operand is
 *                   reduction-contiguous, slower for the weight-gradient layout;
 *   OGL_GEMM_AUTO   BF16X6 for forward / input-gradient GEMMs, F32 for weight-gradient GEMMs. */
enum { OGL_GEMM_F32 = 0, OGL_GEMM_BF16X6 = 1, OGL_GEMM_AUTO = 2, OGL_GEMM_QUERY = -1 };
int ogl_set_gemm_mode(int mode); /* OGL_OK / OGL_EINVAL; OGL_GEMM_QUERY changes nothing and returns the current mode */

typedef struct ogl_graph ogl_graph_t;
typedef void* ogl_stream_t; /* hipStream_t */

int ogl_version(void);
/* Hash of the sources (csrc/ and include/ plus compiler flags) this binary was built from; build.py compares it with the tree
 * and rebuilds on mismatch ("unstamped" when compiled without the recipe). */
const char* ogl_source_hash(void);
const char* ogl_status_string(int status);
int ogl_last_hip_error(void); /* hipError_t of the most recent OGL_EHIP on this thread */
/* Diagnostics (tests and same-process A/B runs; never on a hot path): pin which of two forms of a kernel runs — every form returns the
 * same bits.  `previous` (nullable) receives the old setting.  OGL_EINVAL for an unknown knob or a value outside its range.
 *   OGL_KNOB_X3_TILE        tile of the plain / EXT one-split row-major image products — 0: 256 x 128, 1: 128 x 128, 2: 192 x 128; -1 returns
 *                           to the automatic choice.  Every tile computes every output element with the same sequence of MFMAs.
 *   OGL_KNOB_X3_STAGGER     1 / 0 = the producer / consumer image GEMM with STAGGERED multiplier waves (waves 4-7 run the last column
 *                           block of every step but a tile's last one behind the next step's opening barrier, on fragments kept in
 *                           registers: the matrix pipe has work while its SIMD partner's fragments arrive) / with every wave opening a
 *                           step on its fragment loads; 3 = staggered + a static issue priority for waves 4-7; -1 = OGL_X3_STAGGER
 *                           (default 3).  Every accumulator sees its reduction steps in the same order either way.
 *   OGL_KNOB_BLOCK_MIN_LDS  0 = the per-id minima of ogl_build_block_batched (n_ids > 0) through global atomics even where its LDS form applies.
 *   OGL_KNOB_REDUCE_HALF    1 / 0 = the max aggregator WITHOUT argmax over rows of <= 128 floats (the inference passes over a narrow
 *                           projection table) reads two neighbour rows per wave-instruction (half a wave per row) / one.
 *   OGL_KNOB_SEG_ROWS       1 / 0 = ogl_reduce_bwd_seg_apply on blocks of at most 32 768 edges as ONE launch (a block per 8 sources) / as
 *                           the tiled launch + its fix-up launch (sums in the same list order: the same bits where a source's edges
 *                           sit in one tile). */
enum { OGL_KNOB_X3_TILE = 0, OGL_KNOB_X3_STAGGER = 1, OGL_KNOB_BLOCK_MIN_LDS = 2, OGL_KNOB_REDUCE_HALF = 3, OGL_KNOB_SEG_ROWS = 4 };
int ogl_debug_set(int knob, int value, int* previous);

/* ------------------------------------------------------------------------------------------
 * Snapshot adjacency (replaces the DGLGraph the sampler reads:
 * R/train/graph/dynamic_graph_vertex.py:85,132-141 `graph.subgraph(evolving_vertices)`;
 * R/train/graph/dynamic_graph_edge.py:190-218 `add_nodes` + 2x `add_edges`).
 * One time-ordered CSR of the FINAL graph; a snapshot is a (n_present, cut) pair.
 *   indptr  [n+1] int64, indices [nnz] int32 : in-neighbours of v = indices[indptr[v]..indptr[v+1])
 *   keys    [nnz] int32, ascending inside each list (NULL = use `indices`):
 *           vertex stream: key = neighbour id  (ids are arrival-ordered), cut = n_present;
 *           edge stream  : key = row of the time-sorted edge table,       cut = t*edges_per_snapshot.
 * The handle borrows the three arrays (they must outlive it) and owns an int32[n] degree array.
 * ---------------------------------------------------------------------------------------- */
int ogl_graph_create(const int64_t* indptr, const int32_t* indices, const int32_t* keys,
                     int64_t n, int64_t nnz, ogl_graph_t** out);
/* deg_t[v] = #{e in adj(v): keys[e] < cut} for v < n_present, else 0.  This IS evolve(). */
int ogl_graph_set_snapshot(ogl_graph_t* g, int64_t n_present, int64_t cut, ogl_stream_t stream);
int ogl_graph_degrees(const ogl_graph_t* g, const int32_t** deg_out); /* device pointer, int32[n] */
int ogl_graph_copy_degrees(const ogl_graph_t* g, int32_t* out /* device int32[n] */, ogl_stream_t stream);
int ogl_graph_destroy(ogl_graph_t* g);

/* ------------------------------------------------------------------------------------------
 * One layer of dgl.sampling.MultiLayerNeighborSampler([S,S], replace=True)
 * (R/train/graphsage/pytorch/model.py:44,128,174,224,280,312): for every dst, `fanout` uniform
 * draws with replacement from its snapshot in-neighbours; -1 everywhere when the in-degree is 0.
 * Draw j of dst d = word (j&3) of Philox4x32-10(counter = {j>>2 | layer<<16, d_lo, d_hi, ctr_lo},
 * key = {seed_lo, seed_hi ^ ctr_hi}); offset = (draw * deg) >> 32.  picks: int64 [n_dst, fanout].
 * ---------------------------------------------------------------------------------------- */
int ogl_sample_layer(const ogl_graph_t* g, const int64_t* dst, int64_t n_dst, int fanout,
                     uint64_t seed, uint64_t ctr, int layer, int64_t* picks, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * dgl.to_block relabelling (inside NodeDataLoader, same call sites): src nodes = dst nodes first
 * (local id i = dst[i]), then new ids in first-appearance order of row-major `picks`.
 *   src_ids   int64 [>= n_dst*(1+fanout)]  out (first *n_src_out entries valid)
 *   n_src_out int64 [1]                    out (device)
 *   local_idx int32 [n_dst, fanout]        out (-1 where picks == -1)
 * ---------------------------------------------------------------------------------------- */
int64_t ogl_block_workspace_bytes(int64_t n_dst, int fanout);
int ogl_build_block(const int64_t* dst, int64_t n_dst, const int64_t* picks, int fanout,
                    int64_t* src_ids, int64_t* n_src_out, int32_t* local_idx,
                    void* workspace, int64_t workspace_bytes, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * graph.ndata['feat'][input_nodes] / graph.ndata['target'][seeds]
 * (R/train/graphsage/pytorch/model.py:54,88,91,182-183,232-233).  ld, ldo multiples of 4 floats
 * and 16-byte aligned bases select the 16-B/lane path; anything else falls back to dword copies.
 * ids outside [0, n_rows) never touch memory: they yield a zero row / label -1.
 * ---------------------------------------------------------------------------------------- */
int ogl_gather_rows(const float* table, int64_t ld, int64_t n_rows, const int64_t* ids, int64_t n,
                    int d, float* out, int64_t ldo, ogl_stream_t stream);
/* Zero fill of `bytes` bytes (any alignment): what the atomic-scatter backward kernels need in front of them (ogl_reduce_bwd,
 * ogl_out_layer_bwd_inputs) when no loss launch cleared their target on the side.  A kernel launch (capturable). */
int ogl_fill_zero(void* ptr, int64_t bytes, ogl_stream_t stream);
/* Measurement only (bench.py `hbm_copy_measured`; SURVEY.md section 8(d): "re-measure with a stream-copy microbench on the box"): dst[0 ..
 * bytes) = src[0 .. bytes) as one float4 per thread (the fastest form on this part: tools/micro/stream_copy.hip).  16-byte aligned, bytes a
 * multiple of 16. */
int ogl_stream_copy(const void* src, void* dst, int64_t bytes, ogl_stream_t stream);
int ogl_gather_i64(const int64_t* table, int64_t n_rows, const int64_t* ids, int64_t n,
                   int64_t* out, ogl_stream_t stream);

/* feat_drop of a SAGEConv layer (R/train/graphsage/pytorch/graphsage_dgl.py:41 `feat_drop=dropout` -> nn.Dropout on the
 * layer input; `--dropout`, R/train/__main__.py:36,124), optionally fused with the row gather of the layer-0 input:
 *   out[i, j] = keep(i, j) ? src[row(i), j] / (1 - p) : 0     for i < M, j < N  (rows nullable; ids outside [0, nrows) -> 0)
 *   keep(i, j) = word (j & 3) of Philox4x32-10(counter = {j >> 2, i_lo, i_hi, ctr_lo}, key = {seed_lo, seed_hi ^ ctr_hi})
 *                >= floor(p * 2^32);   0 <= p < 1.
 * The mask depends on (i, j, seed, ctr) only, so the backward pass is the same call on the output gradient (rows = NULL). */
int ogl_dropout_rows(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t M, int N, double p,
                     uint64_t seed, uint64_t ctr, float* out, int64_t ldo, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fixed-fanout neighbour reduction = the aggregator (DGL copy_src -> max on the live 'pool'
 * layer; mailbox.mean/.sum(axis=1) in R/train/graphsage/pytorch/aggregator_dgl.py:158,165,185).
 *   out[d,:] = op_j src[row(d,j), 0:d]   with row(d,j) = idx32[d,j] (block-local) or idx64[d,j]
 *   (global row of a table, e.g. sampler picks used directly against a cached projection);
 *   exactly one of idx32 / idx64 is non-NULL.  A dst whose slot 0 is -1 yields zeros.
 *   mean = float32 sum in slot order 0..S-1 then division by S; max keeps the first slot on ties.
 *   argmax (nullable, int32 [n_dst, d]): row index of the winning source (-1 if none), max only.
 * Backward (mean/sum: every slot; max: the argmax row only) accumulates into a PRE-ZEROED dsrc
 * with float atomics.  Row indices outside [0, n_src) are skipped, never dereferenced.
 * relu_out (nullable, max only) = the forward output: when the reduced rows were ReLU outputs the winner's
 * value is out[d,c], so (out > 0) is its ReLU mask and is applied here instead of in the projection's backward.
 * On the 16-B path the pad columns of `out` up to round_up(d,4) are overwritten.
 * ---------------------------------------------------------------------------------------- */
int ogl_reduce_fwd(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32,
                   const int64_t* idx64, int64_t n_dst, int fanout, int d, int op, float* out,
                   int64_t ldo, int32_t* argmax, ogl_stream_t stream);
/* ogl_reduce_fwd(OGL_REDUCE_MAX) that ALSO writes the bf16x3 image of `out` (ogl_x3_image_bytes(n_dst, d) bytes; byte for byte
 * what ogl_x3_split(out) builds): the pooled rows' consumer (fc_neigh, R/train/graphsage/pytorch/aggregator_dgl.py:206) then
 * runs on the image kernel without a split pass.  Vectorised path only (16-byte-aligned operands, ld multiples of 4).  `out` may be
 * NULL (pass ldo as for a real one): the image only — a consumer that reads nothing else (the cached inference layers). */
int ogl_reduce_fwd_img(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32, const int64_t* idx64, int64_t n_dst,
                       int fanout, int d, float* out, int64_t ldo, int32_t* argmax, void* image, ogl_stream_t stream);
/* ... and for the MEAN (mailbox.mean(axis=1) of the in-repo 'meanpool' / 'mean' layers, aggregator_dgl.py:156-159,181-185): the slot-order
 * sum divided by the fanout, + the bf16x3 image of the result (no argmax). */
int ogl_reduce_fwd_mean_img(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32, const int64_t* idx64, int64_t n_dst,
                            int fanout, int d, float* out, int64_t ldo, void* image, ogl_stream_t stream);
/* ... over rows of a resident TABLE through a block's local indices: out[d] = mean_j table[rows[idx32[d, j]]] (slot order; an index
 * outside [0, n_rows) or a row id outside [0, n_table) counts as no neighbour) + the bf16x3 image of out — the first layer of the
 * in-repo 'mean' mode (R/train/graphsage/pytorch/aggregator_dgl.py:156-159) reading graph.ndata['feat'] where it lies
 * (R/train/graphsage/pytorch/model.py:88 gathers feat[input_nodes] first).  n_table < 2^31; operands as for ogl_reduce_fwd_img. */
int ogl_reduce_fwd_rows_mean_img(const float* table, int64_t ldt, int64_t n_table, const int32_t* idx32, const int64_t* rows,
                                 int64_t n_rows, int64_t n_dst, int fanout, int d, float* out, int64_t ldo, void* image,
                                 ogl_stream_t stream);
int ogl_reduce_bwd(const float* dout, int64_t ldo, const int32_t* idx32, const int32_t* argmax,
                   const float* relu_out, int64_t ldr, int64_t n_dst, int fanout, int d, int op,
                   int64_t n_src, float* dsrc, int64_t lds, ogl_stream_t stream);
/* The same backward for mean / sum WITHOUT float atomics and without a zero fill: the edges are sorted by source once per block
 * (a plan that needs the indices only — it can run beside the forward pass) and the backward is a load-balanced segmented gather
 * (csrc/reduce_seg.hip), reproducible from run to run:
 *     dsrc[s, :] = (1 / divisor) * sum_{(dst, j): idx[dst, j] = s} dout[dst, :]      (in edge order; divisor = fanout for the mean)
 *   ogl_reduce_bwd_seg_plan    counts, scan, placement and ranking of the edges by source into `workspace`
 *                              (ogl_reduce_bwd_seg_workspace_bytes; 16-byte aligned; ids outside [0, n_src) are skipped);
 *   ogl_reduce_bwd_seg_apply   the gather, from a planned workspace: rows optionally multiplied by [mask[s, :] > 0] (the ReLU in
 *                              front of a pooling mean, aggregator_dgl.py:181-185), written as fp32 (`out`, nullable) and / or as
 *                              the row-major bf16x3 image of [n_src, d] (`image`, ogl_x3_image_bytes(n_src, d), nullable): what
 *                              ogl_linear_bwd_weight_x3k reads as its dy operand (interleave -1) — the first layer's pooled-row
 *                              gradient is then never materialised in fp32.  `add` (nullable; needs `out`): rows s < n_add of `out`
 *                              get add[s, :] on top — the head rows' own gradient of a SAGE layer (h[:n_dst] feeds fc_self, every
 *                              row feeds the mean) joins inside the launch instead of through an add launch behind it; the image
 *                              (dsrc's OTHER consumer reads the sum too) carries it as well.  d a multiple of 4, <= 1024; 16-byte
 *                              aligned rows. */
int64_t ogl_reduce_bwd_seg_workspace_bytes(int64_t n_dst, int fanout, int d, int64_t n_src);
int ogl_reduce_bwd_seg_plan(const int32_t* idx, int64_t n_dst, int fanout, int64_t n_src, int group_lists, void* workspace,
                            int64_t workspace_bytes, ogl_stream_t stream);
int ogl_reduce_bwd_seg_apply(const float* dout, int64_t ldd, const int32_t* idx, int64_t n_dst, int fanout, int d, int op, int64_t n_src,
                             const float* mask, int64_t ldm, const float* add, int64_t lda, int64_t n_add, float* out, int64_t ldo,
                             void* image, void* workspace, int64_t workspace_bytes, ogl_stream_t stream);
/* ... and as the TRANSPOSED, group-major bf16x3 image of dsrc (sources dealt round-robin over G = ceil(n_src / 32) groups of 32; the
 * layout ogl_pool_bwd_x3 writes: ogl_x3_image_bytes(d, 32 G) bytes): the dy operand of ogl_linear_bwd_weight_x3k with interleave = G,
 * i.e. the 'meanpool' first layer's fc_pool weight gradient on the same 256 x 128 product as the 'pool' mode's
 * (R/train/graphsage/pytorch/aggregator_dgl.py:178-186).
 *   ogl_reduce_bwd_seg_plan(..., group_lists = 1, ...)   the plan + its lists copied GROUP-MAJOR into the same workspace (two more launches,
 *                                    gradient-free like the plan: beside the forward pass);
 *   ogl_reduce_bwd_seg_apply_t       one block per source group, a thread per two columns, the group's entries read as one block-uniform
 *                                    run, sums in the plan's list order, no partial rows and no fix-up launch.  d <= 640; rows 8-byte
 *                                    aligned (even leading dimensions); mask nullable. */
int ogl_reduce_bwd_seg_apply_t(const float* dout, int64_t ldd, int64_t n_dst, int fanout, int d, int op, int64_t n_src, const float* mask,
                               int64_t ldm, void* image, const void* workspace, int64_t workspace_bytes, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense projections (torch.nn.Linear inside SAGEConv: fc_pool / fc_self / fc_neigh), fp32 MFMA.
 *   fwd:  y[M,N] = act( x[rows?] . w^T  (+ x2[rows2?] . w2^T)  + bias )
 *         x [*,K] (ldx), w [N,K] (ldw); the optional second pair gives fc_self(h)+fc_neigh(n) and
 *         concat->Linear in one pass; x_rows/x2_rows (nullable int64[M]) gather rows on the fly
 *         from a table of x_nrows rows (out-of-range rows read as zeros).
 *   relu_bwd:   out = dy (.) [y > 0]   — the mask of a fused-ReLU projection, applied once before its backward GEMMs
 *   bwd_input:  dx[M,K] = dy . w
 *   bwd_weight: dw[N,K] = dy^T . x[rows?], db[N] = column sums of dy (nullable);
 *         deterministic split over M through `workspace` (no atomics).
 * ---------------------------------------------------------------------------------------- */
int ogl_linear_fwd(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows,
                   int64_t M, int K, const float* w, int64_t ldw, int N, const float* bias,
                   const float* x2, int64_t ldx2, const int64_t* x2_rows, int64_t x2_nrows, int K2,
                   const float* w2, int64_t ldw2,
                   int relu, float* y, int64_t ldy, ogl_stream_t stream);
/* ogl_linear_fwd for a layer whose two projections each carry a bias — fc_self(h) + fc_neigh(neigh) with two nn.Linear(bias=True)
 * (DGL SAGEConv('pool'), R/train/graphsage/pytorch/graphsage_dgl.py:3): y = act(x . w^T + x2 . w2^T + (bias + bias2)); the sum of
 * the two biases is formed first (the rounding of the `bias + bias2` launch this replaces).  bias and bias2 are both required. */
int ogl_linear_fwd_dual_bias(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K, const float* w,
                             int64_t ldw, int N, const float* bias, const float* bias2, const float* x2, int64_t ldx2,
                             const int64_t* x2_rows, int64_t x2_nrows, int K2, const float* w2, int64_t ldw2, int relu, float* y,
                             int64_t ldy, ogl_stream_t stream);
/* ogl_linear_fwd with a per-ROW addend read from a table: y[i, :] = act(x[row(i)] . w^T + bias + add[add_rows[i], :])
 * (add_rows nullable = row i; ids outside [0, add_nrows) add nothing).  The inference layers read their self term
 * fc_self(x) + biases from a per-vertex table computed once per pass (R/inference_optimized.py:169,258 `h0proj`). */
int ogl_linear_fwd_addrows(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K,
                           const float* w, int64_t ldw, int N, const float* bias, const float* add, int64_t ld_add,
                           const int64_t* add_rows, int64_t add_nrows, int relu, float* y, int64_t ldy,
                           ogl_stream_t stream);
int ogl_relu_bwd(const float* dy, int64_t ldy, const float* y, int64_t ldyy, int64_t M, int N,
                 float* out, int64_t ldo, ogl_stream_t stream);
int ogl_linear_bwd_input(const float* dy, int64_t ldy, int64_t M, int N, const float* w, int64_t ldw,
                         int K, float* dx, int64_t lddx, ogl_stream_t stream);
int64_t ogl_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K);
int ogl_linear_bwd_weight(const float* dy, int64_t ldy,
                          const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows,
                          int64_t M, int N, int K, float* dw, int64_t lddw, float* db,
                          void* workspace, int64_t workspace_bytes, ogl_stream_t stream);

/* Weight gradient from TRANSPOSED operands: dw[N,K] = dyT[N,M] . xT[K,M]^T, db[N] = row sums of dyT (nullable).
 * Same result as ogl_linear_bwd_weight; both operands are reduction-contiguous, so it runs on the forward GEMM's
 * fast path (split-bf16 arithmetic in BF16X6 / AUTO mode).  ogl_transpose produces the operands:
 * dst[j, i] = src[row(i), j] for i < M, j < N (rows nullable = gather from a table of nrows rows). */
int64_t ogl_linear_bwd_weight_t_workspace_bytes(int64_t M, int N, int K);
int ogl_linear_bwd_weight_t(const float* dyT, int64_t lddyT, const float* xT, int64_t ldxT, int64_t M, int N,
                            int K, float* dw, int64_t lddw, float* db, void* workspace,
                            int64_t workspace_bytes, ogl_stream_t stream);
int ogl_transpose(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t M, int N,
                  float* dst, int64_t ldt, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same projections on PRE-SPLIT operands ("bf16x3 images", linear_x3.hip): the exact 3-term bf16 split of the
 * x6 arithmetic is done once per matrix by ogl_x3_split / ogl_x3_split_t instead of by every GEMM block, and the
 * GEMM stages its tiles global -> LDS by LDS-DMA.  Same six product terms as OGL_GEMM_BF16X6 (fp32 accuracy; the
 * summation order inside a term differs, so the two agree to fp32 rounding noise, not bit for bit).
 * Used for the large layer-0 products: the static feature table (R/train/graphsage/pytorch/model.py:62,
 * `graph.ndata['feat'][input_nodes]`) is split once per dataset, weights once per call.
 *
 * Image of an fp32 matrix [R, K]: (R + 1) rows (row R is all zero) of ceil(K/32) groups of 192 bytes; a group holds
 * 32 consecutive reduction elements as three 64-byte bf16 planes (hi, mid, lo term), pad elements are zero.
 *   ogl_x3_split    image row r = split(src[row(r), 0:K])           (rows nullable; ids outside [0, nrows) -> zero row).
 *                   append != 0 gives the image ONE extra reduction element at k = K (image reduction length K + 1:
 *                   size it with ogl_x3_image_bytes(R, K + 1)): append = 1 -> 1.0 in every row INCLUDING the zero row
 *                   (activations side), append = 2 -> append_vec[r] (weights side: the bias).  The product of two
 *                   such images is x . w^T + bias: the bias is added by the matrix pipe, ogl_linear_fwd_x3 has no
 *                   bias argument.
 *   ogl_x3_split_t  image row n = split(src[row(0:M), n]) for n < N: the image of the TRANSPOSE (reduction over the M
 *                   source rows); ones_row != 0 appends image row N = 1.0 for m < M (bias-gradient operand).  The image
 *                   then has N + 1 (+ zero) rows: size it with ogl_x3_image_bytes(N + 1, M).  Transposed images are
 *                   stored GROUP-MAJOR — [group][image row][192 bytes] instead of [image row][group][192 bytes] — so
 *                   that the rows of one reduction step are contiguous for their producers and for the GEMM's
 *                   stage loads; they are consumed only by ogl_linear_bwd_weight_x3.  interleave = G > 0 deals
 *                   the reduction index round-robin over G groups of 32 (the layout ogl_pool_bwd_x3 produces): the
 *                   image's reduction length becomes 32 * G (>= M required) and index m holds source row
 *                   (m % 32) * G + m / 32, zeros where that is >= M; the ones row covers every index < 32 * G.
 *   ogl_linear_fwd_x3         y[M,N] = act(x_img[row(i), :] . w_img^T), K = the images' reduction length (both built
 *                             with the same append choice); x image [x_img_rows, K] (x_img_rows = the R it was built
 *                             with: the zero row sits there) with optional gather by x_rows (ids outside
 *                             [0, x_nrows), x_nrows <= x_img_rows, read the zero row); w image [N, K].  y's pad
 *                             columns up to the next multiple of 4 may be overwritten.
 *   ogl_linear_bwd_weight_x3  dw[N,K], db[N] (nullable) from the images of dy^T ([N rows, M]) and [x | 1]^T
 *                             ([K + 1 rows, M], ones_row = 1)
 * ---------------------------------------------------------------------------------------- */
/* Backward of relu -> max-pool over sampled neighbours, fused down to the weight-gradient operand (pool_bwd_x3.hip):
 * the bf16x3 image of dP^T, where dP[s, f] = sum over {i : argmax[i, f] == s} of dout[i, f] * [relu_out[i, f] > 0]
 * (relu_out nullable: no mask) is what ogl_reduce_bwd(OGL_REDUCE_MAX) would scatter — without materialising dP and
 * without global float atomics.  idx32 = the block's [n_dst, fanout] local indices (argmax values are drawn from it);
 * fanout <= 63, d <= 640.
 * The image has d rows over a reduction of length 32 * G, G = ceil(n_src / 32), and its reduction index is DEALT
 * round-robin: index m stands for source row (m % 32) * G + m / 32 (sources >= n_src contribute zeros), so that
 * frequently sampled sources — numbered first by block builders — spread over all 32-deep groups.  Build the other
 * operand of the weight-gradient product with the same dealing: ogl_x3_split_t(..., interleave = G).
 * image: ogl_x3_image_bytes(d, 32 * G) bytes. */
int64_t ogl_pool_bwd_x3_workspace_bytes(int64_t n_dst, int fanout, int d, int64_t n_src);
int ogl_pool_bwd_x3(const float* dout, int64_t ldo, const int32_t* argmax, const float* relu_out, int64_t ldr,
                    const int32_t* idx32, int64_t n_dst, int fanout, int d, int64_t n_src, void* image,
                    void* workspace, int64_t workspace_bytes, ogl_stream_t stream);
/* The same backward in two calls.  Everything ogl_pool_bwd_x3 derives from argmax / relu_out / idx32 — which source groups a
 * destination touches, the order and the segment offsets of its columns — is known when the FORWARD aggregation has run:
 * ogl_pool_bwd_x3_plan writes it into `workspace` (same size as above) and can be enqueued right after the forward pass, on another
 * stream, beside the forward products; ogl_pool_bwd_x3_apply (the backward's critical path: autograd of the max-pool in
 * R/train/graphsage/pytorch/aggregator_dgl.py:171 feeding fc_pool's weight gradient, :199-206) builds the image from `dout` and that
 * workspace.  plan + apply on one stream = ogl_pool_bwd_x3 up to the order of its float additions.  The workspace must stay
 * untouched between the two calls; n_dst / fanout / d / n_src / idx32 must be the same in both; n_dst * d < 2^27.
 * (plan: the row's columns in slot order + per-group record counts -> scan -> every (destination, slot) segment's place in a
 * group-major record array; apply: one wave per destination writes its (column, lane, value) records into those places, then one
 * block per source group streams its records into the slab — no dependent read on the backward's critical path.) */
int ogl_pool_bwd_x3_plan(const int32_t* argmax, const float* relu_out, int64_t ldr, const int32_t* idx32, int64_t n_dst, int fanout,
                         int d, int64_t n_src, void* workspace, int64_t workspace_bytes, ogl_stream_t stream);
int ogl_pool_bwd_x3_apply(const float* dout, int64_t ldo, const int32_t* idx32, int64_t n_dst, int fanout, int d, int64_t n_src,
                          void* image, const void* workspace, int64_t workspace_bytes, ogl_stream_t stream);
int64_t ogl_x3_image_bytes(int64_t rows, int64_t K);
int ogl_x3_split(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t R, int K, int append,
                 const float* append_vec, void* image, ogl_stream_t stream);
int ogl_x3_split_t(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t M, int N, int ones_row,
                   int64_t interleave, void* image, ogl_stream_t stream);
/* Diagnostics (tools/gemm_x3_bench.py clock): while `buf` (device memory, >= 8 192 bytes) is set, every image-GEMM launch
 * records per block {s_memtime, s_memrealtime} at entry and exit (u64[4] per block); the shader clock the chip held inside
 * the kernel is d(s_memtime) / d(s_memrealtime) x 100 MHz.  Pass NULL to switch off.  `reserved` is ignored.  Not part of
 * the hot path. */
int ogl_x3_debug_stamps(void* buf, int reserved);
/* Diagnostics (bench.py): the image-GEMM instantiation the LAST ogl_linear_*_x3* call launched, template arguments as written at the
 * launch site (trailing defaults omitted), e.g. "k_gemm_x3p<4, 2, 2, 2, 2, false, true>"; "" before the first launch.  Static storage.
 * bench.py compares it with the kernel name recorded in the committed rocprofv3 --pmc pass before quoting that pass's HBM-side
 * traffic beside a launch it timed (a tile / kernel change makes the quoted constant stale).  Not part of the hot path. */
const char* ogl_x3_last_kernel(void);

int ogl_linear_fwd_x3(const void* x_img, int64_t x_img_rows, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K,
                      const void* w_img, int N, int relu, float* y, int64_t ldy, ogl_stream_t stream);
/* ogl_linear_fwd_x3 with a second A part, a per-row addend and / or an image of the output (k_gemm_x3p<..., EXT>):
 *   y[i, :] = act( x_img[row(i)] . w[:, part 1]^T + x2_img[row2(i)] . w[:, part 2]^T + add[add_rows[i], :] )
 * - x2_img (nullable, K2 = 0): the second part of a K-concatenated product — fc_self(x[dst]) + fc_neigh(neigh) of the combine
 *   (R/train/graphsage/pytorch/aggregator_dgl.py:199-206) as ONE product; the w image has ceil(K1 / 32) + ceil(K2 / 32) groups per
 *   row (build its parts with ogl_x3_split_into); K1 / K2 are the reduction lengths the A images were built with.
 * - add (nullable): per-row addend from a table, as ogl_linear_fwd_addrows.
 * - out_img (nullable): ALSO write the bf16x3 image of y (row-major, M + 1 rows, reduction length N, + 1 when out_append_ones:
 *   1.0 at column N in every row incl. the zero row) = what ogl_x3_split(y, append = out_append_ones) would build: the A operand
 *   of the next layer's product, without a pass of its own.  Size it with ogl_x3_image_bytes(M, N + out_append_ones).
 * - mask (nullable; [M, N] fp32, ld_mask and N multiples of 4, 16-byte aligned): y[i, j] is zeroed where mask[i, j] <= 0, after the
 *   addend — when this product is an INPUT GRADIENT dX = dY . W (+ head) and `mask` the forward output of the fused-ReLU layer that
 *   produced X (autograd of F.relu, R/train/graphsage/pytorch/graphsage_dgl.py:29-31), the gradient leaves the kernel already
 *   masked, with its image: no ogl_relu_bwd_img pass.
 * - y_keep (nullable, [M] bytes; needs out_img): the fp32 row i of y is stored only where y_keep[i] != 0 — the hidden layer of an
 *   inference pass is read as an IMAGE by the next layer's fc_pool and as fp32 only at the next block's destination rows
 *   (`h[:n_dst]` feeding fc_self, R/train/graphsage/pytorch/aggregator_dgl.py:145-146): 96 % of the fp32 rows are never read. */
int ogl_linear_fwd_x3_ext(const void* x_img, int64_t x_img_rows, const int64_t* x_rows, int64_t x_nrows, int K1,
                          const void* x2_img, int64_t x2_img_rows, const int64_t* x2_rows, int64_t x2_nrows, int K2, int64_t M,
                          const void* w_img, int N, const float* add, int64_t ld_add, const int64_t* add_rows, int64_t add_nrows,
                          int relu, float* y, int64_t ldy, void* out_img, int out_append_ones, const float* mask, int64_t ld_mask,
                          const unsigned char* y_keep, ogl_stream_t stream);
/* Up to 8 small images in ONE launch — the weight images of a train step (the parameters of nn.Linear in
 * R/train/graphsage/pytorch/aggregator_dgl.py:75-84, re-split after every optimiser step).  Part i becomes groups
 * [group_offset, group_offset + ceil((K + append) / 32)) of every row of `image` (rows image_row_bytes apart, R + 1 of them:
 * the last one zero), so several parts can fill one K-concatenated image.  transpose = 1: image row r is COLUMN r of src
 * (src is [K, R]: the image of W^T for an input-gradient product).  append = 1: reduction element K holds
 * vec1[r] + vec2[r] (either may be NULL = 0): the (summed) bias.
 * transpose = 2: not an image at all but a VECTOR SUM riding in the same launch: image = float[R] <- vec1 + vec2 (the summed
 * bias b_self + b_neigh of a dual projection that runs on fp32 operands; src / K / append / image_row_bytes / group_offset unused). */
typedef struct ogl_x3_split_part {
  const float* src; int64_t ld;
  int64_t R; int32_t K;
  int32_t transpose, append;
  const float* vec1; const float* vec2;
  void* image; int64_t image_row_bytes; int64_t group_offset;
} ogl_x3_split_part;
/* step_dev / scalars_dev (both or neither; NULL: a plain split launch): the optimiser's per-step scalars riding in the same launch (what
 * ogl_adam_step_multi_slabs with prepare = 1 computes in a one-thread launch of its own at the END of a step: ++*step_dev, scalars_dev[0] =
 * lr / (1 - beta1^t), scalars_dev[1] = 1 / sqrt(1 - beta2^t), double arithmetic) moved to the launch that STARTS the step; the step's
 * optimiser launch then passes prepare = 0 to ogl_adam_step_multi_slabs.  n_parts >= 1. */
int ogl_x3_split_multi(const ogl_x3_split_part* parts, int n_parts, int64_t* step_dev, float* scalars_dev, double lr, double beta1,
                       double beta2, ogl_stream_t stream);
/* ogl_relu_bwd that also writes the bf16x3 image of its result (M + 1 rows, reduction length N, no appended slot) = what
 * ogl_x3_split(out) would build: the masked gradient of a fused-ReLU projection (autograd of F.relu in
 * R/train/graphsage/pytorch/graphsage_dgl.py:29-31) is the A operand of the input-gradient product that follows. */
int ogl_relu_bwd_img(const float* dy, int64_t ldy, const float* y, int64_t ldyy, int64_t M, int N, float* out, int64_t ldo,
                     void* image, ogl_stream_t stream);
/* One part of a K-concatenated (weight) image: rows image_row_bytes apart, this part from group `group_offset` on. */
int ogl_x3_split_into(const float* src, int64_t ld, int64_t R, int K, int append, const float* append_vec, void* image,
                      int64_t image_row_bytes, int64_t group_offset, ogl_stream_t stream);
int64_t ogl_linear_bwd_weight_x3_workspace_bytes(int64_t M, int N, int K);
/* ogl_linear_bwd_weight_x3 with x as a ROW-MAJOR image (what ogl_x3_split / the image emitters build): dw[N, K] = dy^T . x[x_rows],
 * reduction over M rows of x, no transposed image of x (autograd of nn.Linear's weight in
 * R/train/graphsage/pytorch/aggregator_dgl.py:199-206).  dyT_img as for ogl_linear_bwd_weight_x3 ([N rows, reduction], group-major;
 * interleave = G when it is ogl_pool_bwd_x3's image: reduction index m then stands for row (m % 32) * G + m / 32 of x[x_rows];
 * interleave = -1: `dyT_img` is the ROW-MAJOR image of dy itself ([M + 1 rows, N], as ogl_relu_bwd_img / ogl_x3_split write it) and is
 * read k-major like x — no transposed image at all).
 * x image [x_img_rows (+ zero row), K (+ 1 when has_ones: the ones slot, which yields db / db2 = both copies of the bias
 * gradient)]; x_rows (nullable) gathers M rows, ids outside [0, x_nrows) read the zero row.  Images must be < 4 GB. */
int64_t ogl_linear_bwd_weight_x3k_workspace_bytes(int64_t M, int64_t interleave, int N, int K, int has_ones);
int ogl_linear_bwd_weight_x3k(const void* dyT_img, int64_t interleave, const void* x_img, int64_t x_img_rows, const int64_t* x_rows,
                              int64_t x_nrows, int64_t M, int N, int K, int has_ones, float* dw, int64_t lddw, float* db, float* db2,
                              void* workspace, int64_t workspace_bytes, ogl_stream_t stream);
/* BOTH weight gradients of a dual-input projection y = x[x_rows] . w1^T + x2 . w2^T in ONE product (linear_x3.hip, two-part B operand):
 * slabs [nsplit][N][ws_ld] with dw1 in columns [0, K1), the bias gradient in column K1 (has_ones: the x image carries the ones slot), dw2
 * in columns [*col2_out, *col2_out + K2).  dy_img: row-major image of dy [M + 1 rows, N]; x_img: row-major image of x (gathered by x_rows);
 * x2_img: row-major image of x2 [>= M rows, K2] (no gather).  autograd of fc_self / fc_neigh of the live layer's combine
 * (R/inference_optimized.py:136-139,276) and of fc_neigh(cat(h_self, h_neigh)) of the in-repo layer (R/train/graphsage/pytorch/
 * aggregator_dgl.py:206).  The workspace query returns 0 and the launch OGL_EINVAL when the plan has a single split (two single
 * products then). */
int64_t ogl_linear_bwd_weight_x3k_dual_workspace_bytes(int64_t M, int N, int K1, int has_ones, int K2);
int ogl_linear_bwd_weight_x3k_dual_slabs(const void* dy_img, int64_t M, int N, const void* x_img, int64_t x_img_rows, const int64_t* x_rows,
                                         int64_t x_nrows, int K1, int has_ones, const void* x2_img, int64_t x2_img_rows, int K2,
                                         void* workspace, int64_t workspace_bytes, int* nsplit_out, int64_t* ws_ld_out, int* col2_out,
                                         ogl_stream_t stream);
int ogl_linear_bwd_weight_x3(const void* dyT_img, const void* xT_img, int64_t M, int N, int K, float* dw,
                             int64_t lddw, float* db, void* workspace, int64_t workspace_bytes,
                             ogl_stream_t stream);

/* Batched forms of the sampler and of the block build: nb independent batches in one set of launches (a loader samples
 * every batch of a snapshot, layer by layer; per-batch launches of these microsecond-sized kernels are latency-bound).
 * dst_start / dst_count / ctr are HOST arrays [nb]: batch b's destinations are dst_base[dst_start[b] .. + dst_count[b]),
 * its Philox batch counter is ctr[b].  Outputs are PACKED by the running sum r_b of the counts: picks / local_idx rows
 * r_b .. r_b + dst_count[b], src_ids at r_b * (1 + fanout) (capacity dst_count[b] * (1 + fanout)), n_src_out[b].
 * Per batch the results are bit-identical to ogl_sample_layer / ogl_build_block. */
int ogl_sample_layer_batched(const ogl_graph_t* g, const int64_t* dst_base, const int64_t* dst_start,
                             const int64_t* dst_count, int nb, int fanout, uint64_t seed, const uint64_t* ctr,
                             int layer, int64_t* picks, ogl_stream_t stream);
/* n_ids <= 0: no bound on the vertex ids (an open-addressing hash table per batch).  n_ids > 0: the caller knows an upper bound of
 * the vertex ids (every id in dst / picks is < n_ids, e.g. the graph's vertex count; ids outside [0, n_ids) are treated like negative
 * ones: no source row, local index -1): a direct-address table of n_ids entries per batch replaces the hash — per position one LDS atomic
 * (id range cut into LDS-sized pieces, n_ids <= 589 824) or one no-return global atomicMin, instead of atomicCAS + atomicMin + probing;
 * the same results bit for bit.  Workspace then: 8 bytes x n_ids per batch of a 64-batch chunk. */
int64_t ogl_block_workspace_bytes_batched(const int64_t* dst_count, int nb, int fanout, int64_t n_ids);
int ogl_build_block_batched(const int64_t* dst_base, const int64_t* dst_start, const int64_t* dst_count, int nb,
                            const int64_t* picks, int fanout, int64_t n_ids, int64_t* src_ids, int64_t* n_src_out,
                            int32_t* local_idx, void* workspace, int64_t workspace_bytes, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * nn.CrossEntropyLoss (R/train/graphsage/pytorch/model.py:20,105,147,198,244):
 *   loss_rows[i] = logsumexp(logits[i,:]) - logits[i, labels[i]]            (reduction='none')
 *   dlogits (nullable) = grad_scale * (softmax - onehot)   (grad_scale = 1/B for the mean loss)
 * ---------------------------------------------------------------------------------------- */
int ogl_ce_fwd_bwd(const float* logits, int64_t ldl, const int64_t* labels, int64_t B, int C,
                   float grad_scale, float* loss_rows, float* dlogits, int64_t lddl,
                   ogl_stream_t stream);
/* The same for B <= 1024 rows in ONE workgroup, with loss_mean[0] = mean of the row losses from the same launch
 * (reduction='mean', R/train/graphsage/pytorch/model.py:20: the loss the RBR / no-rehearsal strategies differentiate), with (a) the
 * label gather inside the launch — label of row i = label_table[label_ids[i]], an id outside
 * [0, n_labels) = no label (what ogl_gather_i64 writes as -1); label_ids null: label_table is the label vector itself — and (b) an
 * optional zero fill of a small caller buffer (zero_floats <= 65 536, a multiple of 4, 16-byte aligned: the atomic-scatter target of
 * the backward pass that follows).  graph.ndata['target'][seeds] (R/train/graphsage/pytorch/model.py:91,183) and that fill were a
 * launch each on the 32-seed rungs. */
int ogl_ce_fwd_bwd_mean_gather(const float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels, const int64_t* label_ids,
                               int64_t B, int C, float grad_scale, float* loss_rows, float* dlogits, int64_t lddl, float* loss_mean,
                               float* zero_buf, int64_t zero_floats, int64_t* step_dev, float* scalars_dev, double lr, double beta1,
                               double beta2, ogl_stream_t stream);
/* (step_dev / scalars_dev, both or neither: the launch also prepares the optimiser's step — what ogl_x3_split_multi does for steps that
 * have a weight-image launch —: ++*step_dev, scalars_dev[0] = lr / (1 - beta1^t), scalars_dev[1] = 1 / sqrt(1 - beta2^t) — torch.optim.Adam's
 * bias corrections (R/train/graphsage/pytorch/model.py:24-25) in double arithmetic; the optimiser launch of the step then runs with
 * prepare = 0.) */
/* The same for ANY batch size (one wave per row over a grid): the mean is summed by the last block to finish, in the fixed
 * order of the one-workgroup form (bit-identical to it).  `counter`: one zero-initialised device word the caller allocates once
 * (the last block resets it; one counter per stream that may run this concurrently).  loss_rows is required.  zero_buf
 * (nullable; 16-byte aligned, zero_floats a multiple of 4): ALSO filled with zeros by the same grid — the atomic-scatter target
 * of the backward pass that follows (autograd of the max-pool in R/train/graphsage/pytorch/graphsage_dgl.py:3's SAGEConv). */
int ogl_ce_fwd_bwd_mean_grid(const float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels, const int64_t* label_ids,
                             int64_t B, int C, float grad_scale, float* loss_rows, float* dlogits, int64_t lddl, float* loss_mean,
                             unsigned int* counter, float* zero_buf, int64_t zero_floats, ogl_stream_t stream);
/* (label_ids NULL: label_table holds the B labels themselves.  Else the label gather is inside the launch: label of row i =
 * label_table[label_ids[i]] (an id outside [0, n_labels) = no label: what ogl_gather_i64 would have written as -1) —
 * graph.ndata['target'][seeds] (R/train/graphsage/pytorch/model.py:91,183) costs no launch of its own.) */

/* torch.optim.Adam(lr) single-tensor update (R/train/graphsage/pytorch/model.py:24-25,107).
 * Hyper-parameters are doubles (as Python floats are) so 1-beta is rounded to fp32 once, like torch. */
int ogl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step,
                  double lr, double beta1, double beta2, double eps, ogl_stream_t stream);

/* Evaluation on the device: pred[i] = argmax_c logits[i, c] (first maximum) and confusion[true*C + pred] += 1
 * (int64 [C*C], ACCUMULATED: zero it before the first batch).  Replaces argmax + sklearn confusion_matrix,
 * R/train/graphsage/model.py:84-87; pred / confusion / labels are each optional. */
int ogl_argmax_confusion(const float* logits, int64_t ldl, const int64_t* labels, int64_t B, int C,
                         int64_t* pred, int64_t* confusion, ogl_stream_t stream);

/* (The same update for `count` tensors in one launch: ogl_adam_step_multi_slabs, below; p/g/m/v/n are HOST arrays of device pointers /
 * lengths.) */

/* ------------------------------------------------------------------------------------------
 * Entry points for a CAPTURED step (hipGraph): everything a replay must see new values of lives in device memory.
 * A train step of the small rungs (32 seeds) is ~40 microsecond-sized launches: enqueued one by one from the host it is
 * launch- and sync-bound; captured once on upper-bound shapes and replayed, it costs one graph launch.
 *   ogl_sample_layer_dev     = ogl_sample_layer with the Philox batch counter read from *ctr_dev on the device;
 *   ogl_adam_step_multi_slabs(…, step_dev, scalars_dev, prepare = 1, …) = Adam with the step count in *step_dev (incremented by the
 *                              call itself on the device, bias corrections derived from it there; scalars_dev = float[2] scratch);
 *   ogl_stage_segments       copies the sampled batch of a loader into the static buffers of a captured step: up to 8
 *                              segments of 4- or 8-byte elements in one launch, each copied for `count` elements and filled
 *                              with `pad` (-1: "no vertex" / "no neighbour") up to `capacity`.
 * Padding contract (what makes fixed shapes exact): a destination id of -1 samples no neighbour and keeps its block row;
 * a source id of -1 gathers the all-zero row; padded rows therefore contribute exact zeros to every product and gradient.
 *   ogl_build_block_padded   = ogl_build_block with src_ids[n_src .. src_cap) set to -1 (the source count stays on the device;
 *                              workspace as ogl_block_workspace_bytes): ONE launch up to 4 096 flat positions
 *                              n_dst (1 + fanout), the parallel phases above that (the fill folded into the table reset).
 */
int ogl_sample_layer_dev(const ogl_graph_t* g, const int64_t* dst, int64_t n_dst, int fanout, uint64_t seed,
                         const uint64_t* ctr_dev, int layer, int64_t* picks, ogl_stream_t stream);
int ogl_build_block_padded(const int64_t* dst, int64_t n_dst, const int64_t* picks, int fanout, int64_t* src_ids,
                           int64_t src_cap, int64_t* n_src_out, int32_t* local_idx, void* workspace,
                           int64_t workspace_bytes, ogl_stream_t stream);

/* Split-K weight gradients whose reduction is left to the optimiser launch (csrc/linear_x3.hip, csrc/loss_optim.hip).  A k-major
 * weight gradient runs as nsplit partial products over ranges of the reduction, each written to its own slab, and a reduction launch
 * then sums the slabs — four ~8 us launches per Reddit train step whose only reader is the optimiser (zero_grad / backward / step,
 * R/train/graphsage/pytorch/model.py:87,106-107,193,201-202).
 *   ogl_linear_bwd_weight_x3k_slabs   = ogl_linear_bwd_weight_x3k without that launch: *nsplit_out > 1 -> workspace holds
 *                                       nsplit slabs [N rows][*ws_ld_out floats] (column K of a row: the bias gradient) and dw / db /
 *                                       db2 are NOT written; *nsplit_out == 1 -> dw / db / db2 are final.
 *   ogl_adam_step_multi_slabs         Adam over `count` tensors; tensor i with ws[i] != NULL takes its gradient from slabs:
 *                                       g_i[r, c] = sum_s ws[i][s * slab_stride[i] + r * ws_ld[i] + col0[i] + c]  (slab order: the bits
 *                                       of the reduction launch), stores it into g[i] (p.grad holds the gradient afterwards) and
 *                                       applies it; rows = n[i] / ncols[i].  A bias takes column K of its weight's slabs: ncols 1,
 *                                       col0 K.  A tensor with ws[i] == NULL reads g[i] as it is (plain Adam).  step_dev NULL:
 *                                       host step count `step`; else the device-side count, incremented (and the scalars refreshed)
 *                                       only when `prepare` != 0 —
 *                                       a step applied in two launches (gradients that are ready early on a side branch, the rest
 *                                       at the end) prepares once.
 *   ogl_x3_slab_reduce                the plain reduction of such slabs into out[rows, ncols] (a gradient somebody reads before
 *                                       the optimiser runs). */
int ogl_linear_bwd_weight_x3k_slabs(const void* dyT_img, int64_t interleave, const void* x_img, int64_t x_img_rows,
                                    const int64_t* x_rows, int64_t x_nrows, int64_t M, int N, int K, int has_ones, float* dw,
                                    int64_t lddw, float* db, float* db2, void* workspace, int64_t workspace_bytes, int* nsplit_out,
                                    int64_t* ws_ld_out, ogl_stream_t stream);
/* TWO-RANGE tensors (split / col0b, both nullable): tensor i with split[i] > 0 takes its columns [0, split) from slab column col0[i] and its columns
 * [split, ncols) from slab column col0b[i] — the concat weight [N, K1 + K2] of the in-repo layer
 * (R/train/graphsage/pytorch/aggregator_dgl.py:94,206) behind ogl_linear_bwd_weight_x3k_dual_slabs, whose slabs are laid out
 * [dw1 | db | pad | dw2]: no reduction launch between the product and the optimiser. */
int ogl_adam_step_multi_slabs(int count, float* const* p, float* const* g, float* const* m, float* const* v, const int64_t* n,
                               const float* const* ws, const int64_t* slab_stride, const int* ws_ld, const int* nsplit, const int* ncols,
                               const int* col0, const int* split, const int* col0b, int step, int64_t* step_dev, float* scalars_dev,
                               int prepare, double lr, double beta1, double beta2, double eps, ogl_stream_t stream);
int ogl_x3_slab_reduce(const float* ws, int64_t slab_stride, int64_t ws_ld, int nsplit, int64_t rows, int ncols, int col0, float* out,
                       int64_t ldo, ogl_stream_t stream);
/* K loader batches as ONE block (inference passes: every kernel of the forward is row-independent, so a pass runs them once per
 * chunk of batches instead of once per batch of `batch_full` seeds, R/train/graphsage/pytorch/model.py:224-248).  local_idx is the
 * packed [rows, fanout] index array of the chunk's output blocks, with destination rows [seg_row[s], seg_row[s + 1]) belonging to
 * batch s (seg_row is relative to the array passed); every valid index gets seg_off[s] — the position of batch s's source list in the
 * fused source list — added in place, and dst_pos[r] (nullable, [rows]) receives the fused position of destination r's own row
 * (seg_off[s] + its rank inside the batch: a block's destinations are the first entries of its source list).  nseg <= 64.
 * dst_flag (nullable; one byte per row of the fused SOURCE list, zeroed by the caller): set to 1 at every destination's own row —
 * the rows of the hidden layer whose fp32 values the next layer's self term reads (ogl_linear_fwd_x3_ext's y_keep). */
int ogl_fuse_block_segments(int32_t* local_idx, int64_t* dst_pos, int nseg, const int64_t* seg_row, const int64_t* seg_off, int fanout,
                            unsigned char* dst_flag, ogl_stream_t stream);
int ogl_stage_segments(int nseg, const void* const* src, void* const* dst, const int64_t* count,
                       const int64_t* capacity, const int* elem_bytes, int64_t pad, ogl_stream_t stream);
/* The read-back of a captured sample graph without a copy node: dst_host_mapped = int64 [n + 1] in pinned HOST memory (mapped
 * into the device); the kernel stores src[0..n) there, then — behind a system-scope fence — the sequence number ++*seq_dev at
 * dst_host_mapped[n], which the host polls for. */
int ogl_publish_i64(const int64_t* src, int n, int64_t* seq_dev, int64_t* dst_host_mapped, ogl_stream_t stream);
/* The whole SAMPLING phase of a small batch in one launch (block.hip, k_sample_blocks_small): [Philox batch counter | B seeds] is read
 * from host-mapped memory into head_dev, the output block (fanout picks per seed, layer 1) and the input block (fanout picks per
 * source found, layer 0) are sampled and relabelled, and the two source counts go to the host as ogl_publish_i64 would send them
 * (counts_host_mapped[0 .. 1], then ++*seq_dev behind a system-scope fence into [2]).  Results as ogl_sample_layer_dev +
 * ogl_build_block_padded on the same shapes: src1 [B (1 + fanout)] / src0 [B (1 + fanout)^2] padded with -1, lidx1 [B, fanout],
 * lidx0 [B (1 + fanout), fanout] (-1 rows for the padded destinations), counts[0] = n1, counts[1] = n0 = B (1 + fanout) + new sources.
 * One 1024-thread workgroup; B (1 + fanout)^2 <= 131 072 (B = 32 at fanout 45: 67 712).  Replaces the per-batch NodeDataLoader iteration of
 * R/train/graphsage/pytorch/model.py:76-117 for the 32-seed rungs (R/settings/pubmed.json, arxiv.json). */
int64_t ogl_sample_blocks_small_workspace_bytes(int B, int fanout);
/* ... with src0 padded with -1 only up to round_up(n0, src0_fill_multiple) (0: up to its capacity, as above): for a caller that reads
 * the input block's source list up to the size bucket of the train graph it replays and no further (21 632 entries per step otherwise). */
int ogl_sample_blocks_small_fill(const ogl_graph_t* g, const int64_t* head_host_mapped, int64_t* head_dev, int B, int fanout, uint64_t seed,
                                 int64_t* src1, int32_t* lidx1, int64_t* src0, int32_t* lidx0, int64_t* counts, int64_t* seq_dev,
                                 int64_t* counts_host_mapped, void* workspace, int64_t workspace_bytes, int64_t src0_fill_multiple,
                                 ogl_stream_t stream);

/* Backward of the combine of a 'pool' layer with few output columns (the output layer: N <= 64 classes), two launches (csrc/out_layer.hip;
 * autograd of fc_self(h[:n_dst]) + fc_neigh(max-pooled rows), R/train/graphsage/pytorch/aggregator_dgl.py:171,199-206):
 *   ogl_out_layer_bwd_inputs   dx_self[n_dst, K] = dy . w_self, and dy . w_neigh scattered to the max winners without being stored:
 *                              dP[argmax[d, c], c] += (dy . w_neigh)[d, c] where neigh[d, c] > 0 (dP zero-initialised by the caller:
 *                              what ogl_linear_bwd_input x 2 + ogl_reduce_bwd(max, relu_out = neigh) compute in three launches).
 *   ogl_out_layer_bwd_weights  dw_self = dy^T . x_self, dw_neigh = dy^T . x_neigh (both [N, K]), db = db2 = column sums of dy (nullable),
 *                              M <= 4096 rows; dy rows 16-byte aligned with lddy a multiple of 4 (as ogl_linear_bwd_weight's skinny path);
 *                              x_self_rows (nullable) gathers x_self's M rows from a table (ids outside [0, x_self_nrows): zero rows) —
 *                              the first layer's combine at the 32-seed rungs. */
int ogl_out_layer_bwd_inputs(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                             const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh, int64_t ldn, int64_t n_src,
                             float* dx_self, int64_t ldx, float* dP, int64_t ldp, ogl_stream_t stream);
/* ogl_out_layer_bwd_inputs that also finishes the loss of the forward launch before it (ogl_out_layer_fwd_ce with counter == NULL: the deferred form, *loss_mean = NaN until this call):
 * *loss_mean = sum(loss_rows[0 .. n_loss)) / n_loss in the one-workgroup order of ogl_ce_fwd_bwd_mean — a kernel boundary instead of a
 * last-block-done count (which costs every block of the forward a device-scope fence: an L2 write-back per block on this part). */
int ogl_out_layer_bwd_inputs_mean(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                                  const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh, int64_t ldn,
                                  int64_t n_src, float* dx_self, int64_t ldx, float* dP, int64_t ldp, const float* loss_rows,
                                  int64_t n_loss, float* loss_mean, ogl_stream_t stream);
int ogl_out_layer_bwd_weights(const float* dy, int64_t lddy, int64_t M, int N, int K, const float* x_self, int64_t ldxs,
                              const int64_t* x_self_rows, int64_t x_self_nrows, const float* x_neigh, int64_t ldxn, float* dw_self,
                              int64_t lddws, float* dw_neigh, int64_t lddwn, float* db, float* db2, ogl_stream_t stream);
/* FORWARD of that layer fused with the loss, one launch (csrc/out_layer.hip: k_out_fwd_ce) — the tail of a train step's forward pass:
 *     neigh[d, :]  = max_j P[idx[d, j], :]            (P = relu(fc_pool(h)) [n_src, K]; argmax[d, c] = first winning row, -1: no neighbour;
 *                                                      neigh = 0 there: ogl_reduce_fwd(OGL_REDUCE_MAX)'s rule)
 *     logits[d, :] = h[d, :] . w_self^T + neigh[d, :] . w_neigh^T + b_self + b_neigh          (N <= 64 classes; biases nullable)
 *     loss_rows[d] = logsumexp(logits[d, :]) - logits[d, label(d)],  label(d) = label_table[label_ids ? label_ids[d] : d]
 *                    (a label / id out of range: loss 0, no target term — ogl_ce_fwd_bwd's rule)
 *     dlogits      = grad_scale * (softmax - onehot)        (nullable)
 *     *loss_mean   = sum(loss_rows) / n_dst, summed by the last block in the one-workgroup order of ogl_ce_fwd_bwd_mean (nullable;
 *                    `counter`: one zeroed device word, reset by the call; counter == NULL with a loss_mean: the DEFERRED form —
 *                    this launch stores NaN there and ogl_out_layer_bwd_inputs_mean, its successor, the value: no device-scope
 *                    fence per block, 25 us instead of 58 at 512 seeds)
 * and zero_buf[0 .. zero_floats) cleared by the same grid (the atomic-scatter target of ogl_out_layer_bwd_inputs; nullable).
 * Replaces ogl_reduce_fwd + ogl_linear_fwd (dual input) + ogl_ce_fwd_bwd_mean_grid: three latency-bound launches for 512 seeds.
 * Requirements (ogl_out_layer_fwd_ce_fits): K a multiple of 4 and <= 1024, fanout <= 64, N <= 64; every matrix 16-byte aligned with a
 * row stride that is a multiple of 4 floats; argmax dense (row stride K).  rows_per_block: 0 = automatic (1, 2 or 4 to pin it).
 * fp32 FMA arithmetic on the vector ALU; the summation order of a logit differs from ogl_linear_fwd's (same tolerances). */
int ogl_out_layer_fwd_ce_fits(int64_t n_dst, int fanout, int K, int N);
int ogl_out_layer_fwd_ce(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, const float* h,
                         int64_t ldh, int K, const float* w_self, int64_t ldws, const float* w_neigh, int64_t ldwn, const float* b_self,
                         const float* b_neigh, int N, float* neigh, int64_t ldn, int32_t* argmax, float* logits, int64_t ldl,
                         const int64_t* label_table, int64_t n_labels, const int64_t* label_ids, float grad_scale, float* loss_rows,
                         float* dlogits, int64_t lddl, float* loss_mean, unsigned int* counter, float* zero_buf, int64_t zero_floats,
                         int rows_per_block, ogl_stream_t stream);
/* The in-repo 'mean' layer as the LAST layer of a train step (round 5): ogl_out_layer_fwd_ce with the neighbour MEAN over the sampled rows
 * of P = the layer's own input (slot order, divided by fanout: ogl_reduce_fwd(OGL_REDUCE_MEAN)'s arithmetic) and w_self / w_neigh = the two
 * column blocks of fc_neigh's concat weight (ldws = ldwn = its width; b_neigh NULL) — mailbox.mean + torch.cat + nn.Linear +
 * nn.CrossEntropyLoss, R/train/graphsage/pytorch/aggregator_dgl.py:156-159,199-206, pytorch/model.py:105 — and
 * ogl_out_layer_bwd_inputs_dense: dx_self = dy . w_self and dneigh = dy . w_neigh both stored ([n_dst, K]; the mean's own backward,
 * ogl_reduce_bwd_seg_apply, follows), optionally finishing the deferred loss mean (loss_rows / loss_mean NULL together). */
int ogl_out_layer_fwd_ce_mean(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, const float* h,
                              int64_t ldh, int K, const float* w_self, int64_t ldws, const float* w_neigh, int64_t ldwn,
                              const float* b_self, const float* b_neigh, int N, float* neigh, int64_t ldn, float* logits, int64_t ldl,
                              const int64_t* label_table, int64_t n_labels, const int64_t* label_ids, float grad_scale, float* loss_rows,
                              float* dlogits, int64_t lddl, float* loss_mean, unsigned int* counter, int rows_per_block,
                              ogl_stream_t stream);
int ogl_out_layer_bwd_inputs_dense(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                                   const float* w_neigh, int64_t ldwn, float* dx_self, int64_t ldx, float* dneigh, int64_t lddn,
                                   const float* loss_rows, int64_t n_loss, float* loss_mean, ogl_stream_t stream);
/* *loss_mean = sum(loss_rows[0 .. n)) / n in ogl_out_layer_bwd_inputs_mean's order: the deferred mean when no such launch follows. */
int ogl_loss_mean_finish(const float* loss_rows, int64_t n, float* loss_mean, ogl_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A whole SMALL 'pool' SAGEConv layer in two launches forward, two launches backward (small_layer.hip):
 *     neigh = max_j relu(h . Wp^T + bp)[idx[:, j]],   y = act(h[:n_dst] . Ws^T + neigh . Wn^T + bs + bn)
 * exact fp32 FMA on the vector ALU, one thread per output element.  For the layers of the reference's small settings
 * (R/settings/pubmed.json, arxiv.json: embedding_size 32, batch 32 -> the output layer of a batch is <= 832 source rows x 32
 * features x <= 40 classes): there the 15 general launches this replaces are pure latency.  ogl_small_pool_layer_fits tells
 * whether a shape qualifies (Hin, Hout <= 64, n_dst * max(Hin, Hout) <= 8192, n_src <= 65536).
 *   workspace: float [n_src * Hin] (forward: the projected rows; backward: the winners' routed gradient, n_dst * Hin of them).
 *   fwd: neigh [n_dst, Hin] and argmax (int32 [n_dst, Hin], nullable: the winning block-local source row, -1 = none) are
 *        outputs kept for the backward; biases nullable.
 *   bwd: dy [n_dst, Hout] (masked here by y > 0 when relu_out); every gradient output nullable; dh [n_src, Hin] is written
 *        completely (zeros where nothing flows; the winners' rows by float atomics).
 * ---------------------------------------------------------------------------------------- */
int ogl_small_pool_layer_fits(int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout);
int ogl_small_pool_layer_fwd(const float* h, int64_t ldh, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, int Hin,
                             const float* Wp, int64_t ldwp, const float* bp, const float* Ws, int64_t ldws, const float* bs,
                             const float* Wn, int64_t ldwn, const float* bn, int Hout, int relu_out, float* neigh, int64_t ldn,
                             int32_t* argmax, float* y, int64_t ldy, float* workspace, ogl_stream_t stream);
int ogl_small_pool_layer_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, int relu_out, const float* h, int64_t ldh,
                             int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout, const float* neigh, int64_t ldn,
                             const int32_t* argmax, const float* Wp, int64_t ldwp, const float* Ws, int64_t ldws, const float* Wn,
                             int64_t ldwn, float* dWp, int64_t lddwp, float* dbp, float* dWs, int64_t lddws, float* dbs,
                             float* dWn, int64_t lddwn, float* dbn, float* dh, int64_t lddh, float* workspace,
                             ogl_stream_t stream);

/* The LAST small 'pool' layer of a train step together with nn.CrossEntropyLoss(reduction='mean') (R/train/graphsage/pytorch/
 * model.py:87-107: logits = model(blocks, x); loss = loss_fcn(logits, labels); loss.backward()) as TWO launches, cut where the data
 * crosses destinations (before: ogl_small_pool_layer_fwd + ogl_ce_fwd_bwd_mean_gather + ogl_small_pool_layer_bwd = five launches and
 * a fill; the same bits).  ogl_small_pool_loss_fits: ogl_small_pool_layer_fits and n_dst <= 128, fanout <= 64.
 *   ogl_small_pool_layer_fwd_ce_bwd: one workgroup per destination projects that destination's neighbour rows, takes the max, forms
 *        the logits and the row's cross entropy, dlogits = (softmax - onehot) * grad_scale, and the two input gradients of the combine.
 *        labels: label of row d = label_table[label_ids[d]] (label_ids NULL: label_table[d]); an id outside [0, n_labels) = no label.
 *        Outputs: y = logits [n_dst, Hout], loss_rows [n_dst], dlogits [n_dst, Hout], neigh [n_dst, Hin], argmax int32 [n_dst, Hin],
 *        G float [n_dst * Hin] (the winners' routed gradient), dh [n_src, Hin] (nullable; written completely: the fc_self path in the
 *        destinations' rows, zeros behind them; dh_head_only: dh is [n_dst, Hin], only those head rows are written — for a consumer
 *        that gathers the rest itself, ogl_small_first_layer_bwd's route).  *loss_mean is set to NaN: its value is the backward
 *        launch's.
 *        zero_buf / zero_floats (multiple of 4, 16-byte aligned, nullable): a caller buffer cleared by fill-only blocks of the grid.
 *   ogl_small_pool_layer_bwd_pool: block 0 sums what crosses destinations — dWs / dWn [Hout, Hin], dbs / dbn [Hout] (nullable) from
 *        dlogits, h's head rows and neigh; *loss_mean = mean(loss_rows); step_dev / scalars_dev (both or neither): the optimiser's
 *        per-step scalars as in ogl_ce_fwd_bwd_mean_gather — the other blocks run fc_pool through the winners: dWp [Hin, Hin],
 *        dbp (nullable), the winners' rows added into dh with float atomics.  A root gradient other than 1: scale dlogits, G and dh
 *        before this call. */
int ogl_small_pool_loss_fits(int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout);
int ogl_small_pool_layer_fwd_ce_bwd(const float* h, int64_t ldh, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, int Hin,
                                    const float* Wp, int64_t ldwp, const float* bp, const float* Ws, int64_t ldws, const float* bs,
                                    const float* Wn, int64_t ldwn, const float* bn, int Hout, const int64_t* label_table,
                                    int64_t n_labels, const int64_t* label_ids, float grad_scale, float* neigh, int64_t ldn,
                                    int32_t* argmax, float* y, int64_t ldy, float* loss_rows, float* loss_mean, float* dlogits,
                                    int64_t lddl, float* G, float* dh, int64_t lddh, int dh_head_only, float* zero_buf,
                                    int64_t zero_floats, ogl_stream_t stream);
int ogl_small_pool_layer_bwd_pool(const float* h, int64_t ldh, int64_t n_dst, int Hin, int Hout, const int32_t* argmax, const float* G,
                                  const float* Wp, int64_t ldwp, const float* neigh, int64_t ldn, const float* dlogits, int64_t lddl,
                                  const float* loss_rows, float* dWp, int64_t lddwp, float* dbp, float* dWs, int64_t lddws, float* dbs,
                                  float* dWn, int64_t lddwn, float* dbn, float* dh, int64_t lddh, float* loss_mean, int64_t* step_dev,
                                  float* scalars_dev, double lr, double beta1, double beta2, ogl_stream_t stream);

/* y = act(x[x_rows] . w^T + bias) for a SMALL product (fc_pool of a 32-seed step's first layer: 700-2 800 gathered rows x 128-500
 * features, `nn.Linear` + relu of DGL SAGEConv 'pool', R/train/graphsage/pytorch/graphsage_dgl.py:26-31) on the exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulate) with 32 x 64 tiles whose four waves split columns and every k-slab — where the
 * general kernels' tiles leave most of the chip idle and pay an operand conversion per step.  x [n_table, K] (x_rows NULL: row m itself; a row id outside the table: zeros), w [N, K]; K % 4 == 0, ldx % 4 == 0,
 * ldw % 4 == 0, 16-byte aligned bases; M <= 65 536, N <= 4 096.  m_live_dev (nullable): a device scalar — only rows below
 * round_up(*m_live_dev, 32) are computed (the rest of y is left untouched): an upper-bound launch whose real size lives on the device. */
int ogl_small_proj_rows(const float* x, int64_t ldx, const int64_t* x_rows, int64_t n_table, int64_t M, int K, const float* w, int64_t ldw,
                        int N, const float* bias, int relu, float* y, int64_t ldy, const int64_t* m_live_dev, ogl_stream_t stream);

/* The FIRST 'pool' layer of a 32-seed step behind its fc_pool product (the live layer, R/train/graphsage/pytorch/graphsage_dgl.py:26-31
 * -> DGL SAGEConv 'pool', at the reference's small settings: in_feats 500 / 128, embedding_size 32, <= 832 destinations):
 *   ogl_small_first_layer_fwd: neigh[d] = max_j P[idx[d, j]] (+ argmax: the winning row of P, -1 = none; the order of ogl_reduce_fwd) and
 *        y[d] = act(X[ids[d]] . Ws^T + neigh[d] . Wn^T + bs + bn) in ONE launch, one wave per destination (before: the max aggregator + a
 *        skinny dual-input product).  P [n_src, F] and the table X [n_table, F] with 16-byte rows (ld % 4 == 0, ld >= 4 ceil(F / 4));
 *        ids int64 [>= n_dst] (NULL: row d); Ws / Wn [H, F] with ld % 4 == 0; argmax int32 [n_dst, F] (nullable).
 *   ogl_small_first_layer_bwd: dy = dout . [y > 0] (relu_out) and dneigh[d] = dy[d] . Wn in ONE launch (before: the ReLU mask + an
 *        input-gradient product), then the dense dneigh [n_dst, F] (mask_dneigh: zero where no winner takes it) and / or the winners'
 *        scatter dP[argmax[d, k], k] += dneigh[d, k] . [neigh[d, k] > 0] with float atomics into a ZEROED dP [n_src, F] (before: a
 *        third launch).
 *        route_* (route_arg NULL: none; then dout is read): the gradient of y ROUTED IN from the small last layer that consumed it —
 *        dout[d, k] = route_head[d, k] (d < route_n_head) + sum over the records q < route_n with route_arg[q] == d of route_G[q] .
 *        route_W[q % route_H, k] (route_arg / route_G = that layer's argmax / G from ogl_small_pool_layer_fwd_ce_bwd with dh_head_only,
 *        route_W its fc_pool weight, route_head its dh head rows): the winners' scatter of that layer as a gather by its consumer, in
 *        record order — no atomics, no zeroed [n_src, H] matrix, and ogl_small_pool_layer_bwd_pool leaves the step.  route_n <= 2048.
 *   n_live_dev (nullable, both calls): a device scalar — destinations d >= *n_live_dev are PADDED rows of a captured step's upper-bound
 *        block (index row all -1, source id -1): they get what the full path would write (forward: zeros, no winners, act(bias);
 *        backward with a route: zero gradients) without its loads.
 *   ogl_small_first_layer_fits: n_dst <= 8192, fanout <= 64, 16 <= F <= 1024, H <= 32.
 *   Sums over F run lane-parallel: fp32 rounding differs from the GEMM kernels' order; max / argmax are exact. */
int ogl_small_first_layer_fits(int64_t n_src, int64_t n_dst, int fanout, int F, int H);
int ogl_small_first_layer_fwd(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, int F,
                              const float* table, int64_t ldt, const int64_t* ids, int64_t n_table, const float* Ws, int64_t ldws,
                              const float* bs, const float* Wn, int64_t ldwn, const float* bn, int H, int relu_out, float* neigh,
                              int64_t ldn, int32_t* argmax, float* y, int64_t ldy, const int64_t* n_live_dev, ogl_stream_t stream);
int ogl_small_first_layer_bwd(const float* dout, int64_t lddo, const float* y, int64_t ldy, int relu_out, int64_t n_dst, int H, int F,
                              const float* Wn, int64_t ldwn, const float* neigh, int64_t ldn, const int32_t* argmax, float* dy,
                              int64_t lddy, float* dneigh, int64_t lddn, float* dP, int64_t lddp, int64_t n_src, int mask_dneigh,
                              const int32_t* route_arg, const float* route_G, int64_t route_n, int route_H, const float* route_W,
                              int64_t route_ldw, const float* route_head, int64_t route_ldh, int64_t route_n_head,
                              const int64_t* n_live_dev, ogl_stream_t stream);
/* Weight gradients from RECORDS, the general form: up to six row groups in ONE launch.  Group i writes, for r < n_out,
 *     dW[r, :] = sum_{d < n_dst} G[d * ldg + r] . rows[id(d, r), :],   db[r] = db2[r] = sum_d G[d * ldg + r]  (db, db2 nullable)
 * where id(d, r) = ids[w] (ids NULL: w), w = arg[d * ldarg + r] (arg NULL: d); a record whose weight is zero, whose w is outside
 * [0, n_idx) or whose id is outside [0, n_rows) contributes nothing.  rows: [n_rows, F] read as float4 (ldr % 4 == 0, 16-byte aligned);
 * n_dst <= 2048; sums in destination order (reproducible, no atomics).  The optional tail (loss_mean and / or step_dev non-NULL):
 * *loss_mean = mean(loss_rows[0 .. n_loss)) in the order of ogl_ce_fwd_bwd_mean, and the optimiser's per-step scalars as in
 * ogl_ce_fwd_bwd_mean_gather.  What a 32-seed step uses it for: the first layer's three weight gradients AND the last layer's
 * three (G = its routed gradient / dlogits, rows = its input rows / neigh), the deferred mean and Adam's scalars — one launch. */
typedef struct {
  const float* G; int64_t ldg;
  const int32_t* arg; int64_t ldarg; int64_t n_idx;
  const int64_t* ids;
  const float* rows; int64_t ldr; int64_t n_rows; int F;
  int64_t n_dst; int n_out;
  float* dW; int64_t lddw; float* db; float* db2;
  const int64_t* n_live;   /* optional device scalar: only destinations d < min(n_dst, *n_live) are visited (the rest have zero weights) */
} ogl_rec_seg_t;
int ogl_record_weight_grads(const ogl_rec_seg_t* segs, int nseg, const float* loss_rows, int64_t n_loss, float* loss_mean,
                            int64_t* step_dev, float* scalars_dev, double lr, double beta1, double beta2, ogl_stream_t stream);

/* The three weight gradients of that layer from records, ONE launch (n_dst <= 2048), each "output row r = sum_d g(d, r) . row(d, r)":
 *   dWp[j, :] = sum_d G[d, j] X[ids[argmax[d, j]], :], dbp[j] = sum_d G[d, j]   (G [n_dst, F] = the MASKED dneigh: ogl_small_first_layer_bwd
 *        with mask_dneigh = 1) — n_dst F records of F MACs instead of the dense [F, n_src] x [n_src, F] product, its zeroed scatter target
 *        and its operand images (callers gate it: n_dst F^2 floats of L2 reads, ops.SMALL_FIRST_DW_MAX_BYTES);
 *   dWs[c, :] = sum_d dy[d, c] X[ids[d], :],  dWn[c, :] = sum_d dy[d, c] neigh[d, :],  dbs[c] = dbn[c] = sum_d dy[d, c]   (before:
 *        ogl_out_layer_bwd_weights, a launch of its own).
 * Every output group nullable (at least one); one workgroup per output row; sums in destination order (reproducible, no atomics). */

/* ------------------------------------------------------------------------------------------
 * Device-side prioritised replay structure (replay.hip): the sum tree of R/train/prioritized_replay/segment_tree.py:69-125
 * and the priority arithmetic of R/train/prioritized_replay/replay_buffer.py:110-245 on arrays in HBM, fed from the per-seed
 * loss tensor of the PBR passes (R/train/graphsage/pytorch/model.py:204-207,248-254) without a device->host transfer.
 *   node   double [2 * cap], cap a power of two: node i has children 2i, 2i + 1, leaves start at cap (zero-initialised);
 *   state  double [4] = {max log-priority, min log-priority, max clipped priority, min clipped priority}: the reference's
 *          running extrema; initialise to {-1, 99999999, -1, 99999999}.
 *   ogl_replay_update   _normalize + _scaled + leaf writes + ancestor refresh for n entries: leaf idx[i] <-
 *          ((log(clip(p_i)) - lo) / (hi - lo) + offset) ** alpha with the RUNNING extrema lo / hi (offset 1e-5 on insert, 1e-6
 *          on update, as the reference).  p = prio32[i] or prio64[i]; both NULL = the admission priority of
 *          R/train/graph/train_test_graph.py:78-93 (start_priority while nothing was scored, else min + 0.95 (max - min) of
 *          the clipped priorities seen).  scratch: double [n].  *err_flag (device int, zero it first) becomes 1 when a scaled
 *          priority is negative / NaN (the reference asserts), 2 when an index is outside [0, cap).
 *   ogl_replay_rebuild  every internal node from the leaves (after growing the tree).
 *   ogl_replay_sample   the tree walks of _sample_proportional for a batch: *out_ptotal = sum of leaves [0, n_items - 1) (the
 *          reference's end-exclusive-twice quirk), out_idx[i] = find_prefixsum_idx(u_strat[i] * stride + i * stride) for
 *          i < batch (stride = p_total / batch) and out_idx[batch + t] = find_prefixsum_idx(u_redraw[t] * p_total); the
 *          uniforms come from the caller's stream (the reference draws them from Python's `random`).
 *   ogl_replay_note_keys  map[keys[i]] = start + i: the vertex id -> leaf index map of the buffer.
 * ---------------------------------------------------------------------------------------- */
int ogl_replay_update(double* node, int64_t cap, const int64_t* idx, const float* prio32, const double* prio64, int64_t n,
                      double clip_lo, double clip_hi, double offset, double alpha, double start_priority, double* state,
                      double* scratch, int* err_flag, ogl_stream_t stream);
int ogl_replay_rebuild(double* node, int64_t cap, ogl_stream_t stream);
int ogl_replay_sample(const double* node, int64_t cap, int64_t n_items, int64_t batch, const double* u_strat,
                      const double* u_redraw, int64_t n_redraw, int64_t* out_idx, double* out_ptotal, ogl_stream_t stream);
int ogl_replay_note_keys(const int64_t* keys, int64_t n, int64_t start, int64_t* map, int64_t map_size, ogl_stream_t stream);
/* TrendPriority / HybridPriority (R/train/prioritized_replay/generate_priority.py:11-58) with their per-vertex state in HBM:
 * values / prev_loss double[n_vertices], init uint8[n_vertices] (1 = never scored), stats double[2] = {mean of the scored vertices'
 * values, their count}.  For the n DISTINCT vertex ids of a batch and their per-seed losses (exactly one of loss32 / loss64):
 * newcomers start from the running mean, values <- alpha values + (1 - alpha) max(0, loss - prev_loss), the mean is carried, and
 * out[i] = the trend (loss_contrib < 0) or loss_contrib * loss + (1 - loss_contrib) * trend (HybridPriority).  *err is set to 1
 * when an id is outside [0, n_vertices). */
int ogl_priority_trend(const int64_t* ids, const float* loss32, const double* loss64, int64_t n, int64_t n_vertices, double* values,
                       double* prev_loss, unsigned char* init, double* stats, double alpha, double loss_contrib, double* out, int* err,
                       ogl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OGL_HIP_H */
