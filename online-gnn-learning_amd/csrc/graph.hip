// Snapshot adjacency handle + k-hop sampler layer.
// Replaces: DGLGraph snapshot state (R/train/graph/dynamic_graph_vertex.py:132-141,
// R/train/graph/dynamic_graph_edge.py:190-218) and dgl.sampling.sample_neighbors inside
// MultiLayerNeighborSampler (R/train/graphsage/pytorch/model.py:44,128,174,224,280,312).
#include <algorithm>
#include "ogl_common.h"

thread_local int g_ogl_last_hip_error = 0;

extern "C" int ogl_version(void) { return OGL_VERSION; }
#ifndef OGL_SOURCE_HASH
#define OGL_SOURCE_HASH "unstamped"
#endif
static const char k_source_stamp[] = "OGL_SOURCE_STAMP:" OGL_SOURCE_HASH;     // (build.py finds the marker in the file's bytes)
extern "C" const char* ogl_source_hash(void) { return k_source_stamp + 17; }
extern "C" int ogl_last_hip_error(void) { return g_ogl_last_hip_error; }
extern "C" int ogl_debug_set(int knob, int value, int* previous) {
  int prev = 0, rc = OGL_EINVAL;
  switch (knob) {
    case OGL_KNOB_X3_TILE: rc = oglx_knob_x3_tile(value, &prev); break;
    case OGL_KNOB_X3_STAGGER: rc = oglx_knob_x3_stagger(value, &prev); break;
    case OGL_KNOB_BLOCK_MIN_LDS: rc = oglx_knob_block_min_lds(value, &prev); break;
    case OGL_KNOB_REDUCE_HALF: rc = oglx_knob_reduce_half(value, &prev); break;
    case OGL_KNOB_SEG_ROWS: rc = oglx_knob_seg_rows(value, &prev); break;
    default: break;
  }
  if (rc == OGL_OK && previous) *previous = prev;
  return rc;
}
extern "C" const char* ogl_status_string(int s) {
  switch (s) {
    case OGL_OK: return "OGL_OK";
    case OGL_EINVAL: return "OGL_EINVAL: bad argument";
    case OGL_ENOMEM: return "OGL_ENOMEM: device allocation failed";
    case OGL_EHIP: return "OGL_EHIP: HIP runtime error";
    case OGL_EWORKSPACE: return "OGL_EWORKSPACE: workspace missing or too small";
    default: return "OGL_E?: unknown status";
  }
}

// ---- prefix-degree cut: one thread per vertex, lower_bound(keys[adj(v)], cut) ---------------
__global__ void __launch_bounds__(256) k_snapshot_degrees(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ keys,
                                                          int64_t n, int64_t n_present, int64_t cut,
                                                          int32_t* __restrict__ deg) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  int32_t d = 0;
  if (v < n_present) {
    int64_t lo = indptr[v], hi = indptr[v + 1];
    const int64_t base = lo;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < cut) lo = mid + 1; else hi = mid;
    }
    d = (int32_t)(lo - base);
  }
  deg[v] = d;
}

extern "C" int ogl_graph_create(const int64_t* indptr, const int32_t* indices, const int32_t* keys,
                                int64_t n, int64_t nnz, ogl_graph_t** out) {
  if (!indptr || !out || n < 0 || nnz < 0 || (nnz > 0 && !indices)) return OGL_EINVAL;
  if (n >= ((int64_t)1 << 31)) return OGL_EINVAL;  // block hash + indices are int32
  ogl_graph* g = new (std::nothrow) ogl_graph();
  if (!g) return OGL_ENOMEM;
  g->indptr = indptr; g->indices = indices; g->keys = keys ? keys : indices;
  g->n = n; g->nnz = nnz; g->deg = nullptr; g->n_present = 0; g->cut = 0;
  if (n > 0) {
    hipError_t e = hipMalloc((void**)&g->deg, sizeof(int32_t) * (size_t)n);
    if (e != hipSuccess) { g_ogl_last_hip_error = (int)e; delete g; return OGL_ENOMEM; }
    e = hipMemset(g->deg, 0, sizeof(int32_t) * (size_t)n);
    if (e != hipSuccess) { g_ogl_last_hip_error = (int)e; (void)hipFree(g->deg); delete g; return OGL_EHIP; }
  }
  *out = g;
  return OGL_OK;
}

extern "C" int ogl_graph_set_snapshot(ogl_graph_t* g, int64_t n_present, int64_t cut, ogl_stream_t stream) {
  if (!g || n_present < 0 || n_present > g->n) return OGL_EINVAL;
  g->n_present = n_present; g->cut = cut;
  if (g->n == 0) return OGL_OK;
  dim3 grid((unsigned)ogl_cdiv(g->n, 256));
  hipLaunchKernelGGL(k_snapshot_degrees, grid, dim3(256), 0, (hipStream_t)stream,
                     g->indptr, g->keys, g->n, n_present, cut, g->deg);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_graph_degrees(const ogl_graph_t* g, const int32_t** deg_out) {
  if (!g || !deg_out) return OGL_EINVAL;
  *deg_out = g->deg;
  return OGL_OK;
}

extern "C" int ogl_graph_copy_degrees(const ogl_graph_t* g, int32_t* out, ogl_stream_t stream) {
  if (!g || (g->n > 0 && !out)) return OGL_EINVAL;
  if (g->n == 0) return OGL_OK;
  OGL_CHECK_HIP(hipMemcpyAsync(out, g->deg, sizeof(int32_t) * (size_t)g->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OGL_OK;
}

extern "C" int ogl_graph_destroy(ogl_graph_t* g) {
  if (!g) return OGL_OK;
  if (g->deg) (void)hipFree(g->deg);
  delete g;
  return OGL_OK;
}

// ---- sampler: one thread per (dst, 4 slots) ------------------------------------------------
// HBM-bound integer work: per dst 8 B id + 4 B degree + 8 B indptr, per slot one 4-B neighbour
// read (random) and one 8-B pick write (coalesced: a thread owns 32 contiguous bytes).
__global__ void __launch_bounds__(256) k_sample_layer(const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices,
                                                      const int32_t* __restrict__ deg_t, int64_t n,
                                                      const int64_t* __restrict__ dst, int64_t n_dst,
                                                      int fanout, int quads, uint32_t k0, uint32_t k1,
                                                      uint32_t c3, uint32_t layer_bits,
                                                      int64_t* __restrict__ picks) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_dst * quads) return;
  int64_t i = t / quads;
  int q = (int)(t - i * quads);
  int64_t d = dst[i];
  uint32_t deg = 0;
  int64_t base = 0;
  if (d >= 0 && d < n) { deg = (uint32_t)deg_t[d]; base = indptr[d]; }
  int j0 = q * 4;
  int64_t* out = picks + i * fanout + j0;
  int cnt = min(4, fanout - j0);
  if (deg == 0) {
    for (int j = 0; j < cnt; ++j) out[j] = -1;
    return;
  }
  philox4 r = philox4x32_10((uint32_t)q | layer_bits, (uint32_t)((uint64_t)d & 0xFFFFFFFFu),
                            (uint32_t)((uint64_t)d >> 32), c3, k0, k1);
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < cnt) {
      uint32_t off = (uint32_t)(((uint64_t)w[j] * (uint64_t)deg) >> 32);
      out[j] = (int64_t)indices[base + off];
    }
  }
}

extern "C" int ogl_sample_layer(const ogl_graph_t* g, const int64_t* dst, int64_t n_dst, int fanout,
                                uint64_t seed, uint64_t ctr, int layer, int64_t* picks,
                                ogl_stream_t stream) {
  if (!g || n_dst < 0 || fanout < 0 || layer < 0 || layer > 0xFFFF) return OGL_EINVAL;
  if (n_dst == 0 || fanout == 0) return OGL_OK;
  if (!dst || !picks) return OGL_EINVAL;
  int quads = (fanout + 3) / 4;
  int64_t total = n_dst * quads;
  uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu);
  uint32_t k1 = (uint32_t)((seed >> 32) ^ (ctr >> 32));
  uint32_t c3 = (uint32_t)(ctr & 0xFFFFFFFFu);
  dim3 grid((unsigned)ogl_cdiv(total, 256));
  hipLaunchKernelGGL(k_sample_layer, grid, dim3(256), 0, (hipStream_t)stream, g->indptr, g->indices,
                     g->deg, g->n, dst, n_dst, fanout, quads, k0, k1, c3, (uint32_t)layer << 16, picks);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}


// ---- batched sampler: the loader samples EVERY batch of a snapshot layer in one launch -------------------------------
// (per-batch launches of this microsecond-sized kernel are latency-bound: 50 batches x 2 layers per snapshot.)
// Batch b's destinations are dst_base[dst_start[b] .. + dst_count[b]); its picks rows start at row_off[b] = the running
// sum of the counts; its Philox counter is ctr[b].  Same draws as ogl_sample_layer per batch (bit-exact).
__global__ void __launch_bounds__(256) k_sample_layer_batched(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                              const int32_t* __restrict__ deg_t, int64_t n,
                                                              const int64_t* __restrict__ dst_base, ogl_batch_desc bd, int fanout,
                                                              int quads, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_bits,
                                                              int64_t* __restrict__ picks) {
  const int b = blockIdx.y;
  const int64_t n_dst = bd.row_off[b + 1] - bd.row_off[b];
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_dst * quads) return;
  int64_t i = t / quads;
  int q = (int)(t - i * quads);
  int64_t d = dst_base[bd.dst_start[b] + i];
  uint32_t deg = 0;
  int64_t base = 0;
  if (d >= 0 && d < n) { deg = (uint32_t)deg_t[d]; base = indptr[d]; }
  int j0 = q * 4;
  int64_t* out = picks + (bd.row_off[b] + i) * fanout + j0;
  int cnt = min(4, fanout - j0);
  if (deg == 0) {
    for (int j = 0; j < cnt; ++j) out[j] = -1;
    return;
  }
  const uint64_t ctr = bd.ctr[b];
  philox4 r = philox4x32_10((uint32_t)q | layer_bits, (uint32_t)((uint64_t)d & 0xFFFFFFFFu), (uint32_t)((uint64_t)d >> 32),
                            (uint32_t)(ctr & 0xFFFFFFFFu), seed_lo, seed_hi ^ (uint32_t)(ctr >> 32));
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < cnt) {
      uint32_t off = (uint32_t)(((uint64_t)w[j] * (uint64_t)deg) >> 32);
      out[j] = (int64_t)indices[base + off];
    }
  }
}

extern "C" int ogl_sample_layer_batched(const ogl_graph_t* g, const int64_t* dst_base, const int64_t* dst_start,
                                        const int64_t* dst_count, int nb, int fanout, uint64_t seed, const uint64_t* ctr,
                                        int layer, int64_t* picks, ogl_stream_t stream) {
  if (!g || nb < 0 || fanout < 0 || layer < 0 || layer > 0xFFFF) return OGL_EINVAL;
  if (nb == 0 || fanout == 0) return OGL_OK;
  if (!dst_base || !dst_start || !dst_count || !ctr || !picks) return OGL_EINVAL;
  const int quads = (fanout + 3) / 4;
  int64_t row = 0;
  for (int b0 = 0; b0 < nb; b0 += OGL_MAX_BATCH) {
    ogl_batch_desc bd;
    const int m = std::min(nb - b0, (int)OGL_MAX_BATCH);
    int64_t mx = 0;
    for (int b = 0; b < m; ++b) {
      if (dst_count[b0 + b] < 0 || dst_start[b0 + b] < 0) return OGL_EINVAL;
      bd.dst_start[b] = dst_start[b0 + b]; bd.row_off[b] = row; bd.ctr[b] = ctr[b0 + b];
      row += dst_count[b0 + b];
      mx = std::max(mx, dst_count[b0 + b]);
    }
    bd.row_off[m] = row;
    if (mx == 0) continue;
    dim3 grid((unsigned)ogl_cdiv(mx * quads, 256), (unsigned)m);
    hipLaunchKernelGGL(k_sample_layer_batched, grid, dim3(256), 0, (hipStream_t)stream, g->indptr, g->indices, g->deg, g->n,
                       dst_base, bd, fanout, quads, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), (uint32_t)layer << 16,
                       picks);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}


// ---- sampler with a DEVICE-side batch counter: replayable inside a captured hipGraph ------------------------------------
// Same draws as ogl_sample_layer(ctr = *ctr_dev); a host-side ctr would be frozen into the graph's kernel arguments.
__global__ void __launch_bounds__(256) k_sample_layer_dev(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const int32_t* __restrict__ deg_t, int64_t n,
                                                          const int64_t* __restrict__ dst, int64_t n_dst, int fanout, int quads,
                                                          uint32_t seed_lo, uint32_t seed_hi, const uint64_t* __restrict__ ctr_dev,
                                                          uint32_t layer_bits, int64_t* __restrict__ picks) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_dst * quads) return;
  int64_t i = t / quads;
  int q = (int)(t - i * quads);
  int64_t d = dst[i];
  uint32_t deg = 0;
  int64_t base = 0;
  if (d >= 0 && d < n) { deg = (uint32_t)deg_t[d]; base = indptr[d]; }
  int j0 = q * 4;
  int64_t* out = picks + i * fanout + j0;
  int cnt = min(4, fanout - j0);
  if (deg == 0) {
    for (int j = 0; j < cnt; ++j) out[j] = -1;
    return;
  }
  const uint64_t ctr = *ctr_dev;
  philox4 r = philox4x32_10((uint32_t)q | layer_bits, (uint32_t)((uint64_t)d & 0xFFFFFFFFu), (uint32_t)((uint64_t)d >> 32),
                            (uint32_t)(ctr & 0xFFFFFFFFu), seed_lo, seed_hi ^ (uint32_t)(ctr >> 32));
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < cnt) {
      uint32_t off = (uint32_t)(((uint64_t)w[j] * (uint64_t)deg) >> 32);
      out[j] = (int64_t)indices[base + off];
    }
  }
}

extern "C" int ogl_sample_layer_dev(const ogl_graph_t* g, const int64_t* dst, int64_t n_dst, int fanout, uint64_t seed,
                                    const uint64_t* ctr_dev, int layer, int64_t* picks, ogl_stream_t stream) {
  if (!g || n_dst < 0 || fanout < 0 || layer < 0 || layer > 0xFFFF) return OGL_EINVAL;
  if (n_dst == 0 || fanout == 0) return OGL_OK;
  if (!dst || !picks || !ctr_dev) return OGL_EINVAL;
  const int quads = (fanout + 3) / 4;
  hipLaunchKernelGGL(k_sample_layer_dev, dim3((unsigned)ogl_cdiv(n_dst * quads, 256)), dim3(256), 0, (hipStream_t)stream,
                     g->indptr, g->indices, g->deg, g->n, dst, n_dst, fanout, quads, (uint32_t)(seed & 0xFFFFFFFFu),
                     (uint32_t)(seed >> 32), ctr_dev, (uint32_t)layer << 16, picks);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- staging of a sampled batch into the static buffers of a captured step ------------------------------------------------
// Up to OGL_STAGE_MAX segments in one launch: copy `count` elements, then fill with `pad` (as int32 / int64 pattern) up to
// `capacity`.  elem = 4 or 8 bytes.
#define OGL_STAGE_MAX 8
struct StageDesc {
  const void* src[OGL_STAGE_MAX];
  void* dst[OGL_STAGE_MAX];
  int64_t count[OGL_STAGE_MAX];
  int64_t capacity[OGL_STAGE_MAX];
  int elem[OGL_STAGE_MAX];
};

__global__ void __launch_bounds__(256) k_stage_segments(StageDesc d, int64_t pad) {
  const int s = blockIdx.y;
  const int64_t cap = d.capacity[s], cnt = d.count[s];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    if (d.elem[s] == 8) ((int64_t*)d.dst[s])[i] = i < cnt ? ((const int64_t*)d.src[s])[i] : pad;
    else ((int32_t*)d.dst[s])[i] = i < cnt ? ((const int32_t*)d.src[s])[i] : (int32_t)pad;
  }
}

extern "C" int ogl_stage_segments(int nseg, const void* const* src, void* const* dst, const int64_t* count,
                                  const int64_t* capacity, const int* elem_bytes, int64_t pad, ogl_stream_t stream) {
  if (nseg < 0 || nseg > OGL_STAGE_MAX) return OGL_EINVAL;
  if (nseg == 0) return OGL_OK;
  if (!src || !dst || !count || !capacity || !elem_bytes) return OGL_EINVAL;
  StageDesc d;
  int64_t mx = 0;
  for (int s = 0; s < nseg; ++s) {
    if (count[s] < 0 || capacity[s] < count[s] || (elem_bytes[s] != 4 && elem_bytes[s] != 8)) return OGL_EINVAL;
    if (capacity[s] > 0 && (!dst[s] || (count[s] > 0 && !src[s]))) return OGL_EINVAL;
    d.src[s] = src[s]; d.dst[s] = dst[s]; d.count[s] = count[s]; d.capacity[s] = capacity[s]; d.elem[s] = elem_bytes[s];
    mx = capacity[s] > mx ? capacity[s] : mx;
  }
  if (mx == 0) return OGL_OK;
  dim3 grid((unsigned)std::min<int64_t>(ogl_cdiv(mx, 256), 1024), (unsigned)nseg);
  hipLaunchKernelGGL(k_stage_segments, grid, dim3(256), 0, (hipStream_t)stream, d, pad);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}


// ---- several loader batches as ONE block (inference passes) ---------------------------------------------------------------------
// An inference pass runs the same row-independent kernels on every batch; K batches laid end to end are one block whose
// sources are the K source lists end to end.  The batches' block-local indices (positions in their own source list) only need
// their list's offset added, and a destination's own row — the FIRST rows of its batch's source list in a bipartite block, so
// `h[:n_dst]` there — becomes an explicit position list.  seg_row[s] .. seg_row[s + 1]: the destination rows of batch s in the
// packed index array; seg_off[s]: where its source list starts in the fused source list.
#define OGL_FUSE_MAX 64
struct FuseDesc {
  int64_t row[OGL_FUSE_MAX + 1];
  int64_t off[OGL_FUSE_MAX];
  int n;
};

__global__ void __launch_bounds__(256) k_fuse_block_segments(int32_t* __restrict__ local_idx, int64_t* __restrict__ dst_pos, int fanout,
                                                             FuseDesc d, unsigned char* __restrict__ dst_flag) {
  const int64_t rows = d.row[d.n] - d.row[0];
  const int64_t total = rows * fanout;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = d.row[0] + e / fanout;
    int s = 0;
    while (s + 1 < d.n && r >= d.row[s + 1]) ++s;
    const int32_t v = local_idx[e];
    if (v >= 0) local_idx[e] = v + (int32_t)d.off[s];
    if (e % fanout == 0) {
      const int64_t pos = d.off[s] + (r - d.row[s]);
      if (dst_pos) dst_pos[e / fanout] = pos;
      if (dst_flag) dst_flag[pos] = 1;
    }
  }
}

extern "C" int ogl_fuse_block_segments(int32_t* local_idx, int64_t* dst_pos, int nseg, const int64_t* seg_row, const int64_t* seg_off,
                                       int fanout, unsigned char* dst_flag, ogl_stream_t stream) {
  if (nseg < 0 || nseg > OGL_FUSE_MAX || fanout <= 0) return OGL_EINVAL;
  if (nseg == 0) return OGL_OK;
  if (!local_idx || !seg_row || !seg_off) return OGL_EINVAL;
  FuseDesc d;
  d.n = nseg;
  for (int s = 0; s <= nseg; ++s) {
    d.row[s] = seg_row[s];
    if (s > 0 && seg_row[s] < seg_row[s - 1]) return OGL_EINVAL;
  }
  for (int s = 0; s < nseg; ++s) {
    d.off[s] = seg_off[s];
    if (seg_off[s] < 0 || seg_off[s] > 0x7FFFFFFF) return OGL_EINVAL;
  }
  const int64_t total = (seg_row[nseg] - seg_row[0]) * fanout;
  if (total == 0) return OGL_OK;
  hipLaunchKernelGGL(k_fuse_block_segments, dim3((unsigned)std::min<int64_t>(ogl_cdiv(total, 256), 2048)), dim3(256), 0,
                     (hipStream_t)stream, local_idx, dst_pos, fanout, d, dst_flag);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- the 16-byte read-back of a captured sample graph, without a copy node ----------------------------------------------------
// dst is HOST memory mapped into the device (a pinned allocation): the kernel stores the n values, then a sequence number
// (++*seq_dev) behind a system-scope fence; the host polls dst[n] for the number it expects.  Replaces a device->host copy
// node + an event record + the wake-up of a blocking synchronise (~15 us of a ~250 us step).
__global__ void k_publish_i64(const int64_t* __restrict__ src, int n, int64_t* __restrict__ seq_dev, volatile int64_t* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int i = 0; i < n; ++i) dst[i] = src[i];
    const int64_t s = *seq_dev + 1;
    *seq_dev = s;
    __threadfence_system();
    dst[n] = s;
    __threadfence_system();
  }
}

extern "C" int ogl_publish_i64(const int64_t* src, int n, int64_t* seq_dev, int64_t* dst_host_mapped, ogl_stream_t stream) {
  if (n < 0 || n > 64 || !src || !seq_dev || !dst_host_mapped) return OGL_EINVAL;
  hipLaunchKernelGGL(k_publish_i64, dim3(1), dim3(64), 0, (hipStream_t)stream, src, n, seq_dev, (volatile int64_t*)dst_host_mapped);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
