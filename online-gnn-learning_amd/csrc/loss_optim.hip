// Cross-entropy (forward + gradient) and the Adam update.
// Replaces nn.CrossEntropyLoss(reduction='mean'|'none') (R/train/graphsage/pytorch/model.py:20,105,147,198,244)
// and torch.optim.Adam(lr=1e-3).step() (R/train/graphsage/pytorch/model.py:24-25,107,202).
// Both are tiny HBM-bound elementwise/row kernels.
#include <algorithm>
#include "ogl_common.h"
#include "x6_arith.h"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wavefront per row
__global__ void __launch_bounds__(256) k_ce_fwd_bwd(const float* __restrict__ logits, int64_t ldl,
                                                    const int64_t* __restrict__ labels, int64_t B, int C,
                                                    float grad_scale, float* __restrict__ loss_rows,
                                                    float* __restrict__ dlogits, int64_t lddl) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* x = logits + row * ldl;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
  s = wave_sum(s);
  const float lse = m + logf(s);
  const int64_t y = labels[row];
  const bool ok = y >= 0 && y < C;
  if (lane == 0 && loss_rows) loss_rows[row] = ok ? lse - x[y] : 0.f;
  if (dlogits) {
    float* g = dlogits + row * lddl;
    for (int c = lane; c < C; c += 64) {
      float p = expf(x[c] - lse);
      g[c] = grad_scale * (p - ((ok && c == (int)y) ? 1.f : 0.f));
    }
  }
}

extern "C" int ogl_ce_fwd_bwd(const float* logits, int64_t ldl, const int64_t* labels, int64_t B, int C,
                              float grad_scale, float* loss_rows, float* dlogits, int64_t lddl,
                              ogl_stream_t stream) {
  if (B < 0 || C <= 0 || ldl < C || (dlogits && lddl < C)) return OGL_EINVAL;
  if (B == 0) return OGL_OK;
  if (!logits || !labels) return OGL_EINVAL;
  hipLaunchKernelGGL(k_ce_fwd_bwd, dim3((unsigned)ogl_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                     labels, B, C, grad_scale, loss_rows, dlogits, lddl);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// The same for a SMALL batch with the mean of the row losses from the same launch (one workgroup: wave w takes rows w, w + 16, ...;
// the row losses are summed in row order by one wave) — the mean as a second launch is pure latency on the 32-seed rungs.
#define CE_SMALL_MAX_B 1024
__global__ void __launch_bounds__(1024) k_ce_fwd_bwd_mean(const float* __restrict__ logits, int64_t ldl,
                                                          const int64_t* __restrict__ labels, int B, int C, float grad_scale,
                                                          float* __restrict__ loss_rows, float* __restrict__ dlogits, int64_t lddl,
                                                          float* __restrict__ loss_mean, const int64_t* __restrict__ label_ids,
                                                          int64_t n_labels, float4* __restrict__ zero_buf, int zero_n4,
                                                          int64_t* __restrict__ adam_step = nullptr, float* __restrict__ adam_scal = nullptr,
                                                          double adam_lr = 0.0, double adam_b1 = 0.0, double adam_b2 = 0.0) {
  __shared__ float rows[CE_SMALL_MAX_B];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (adam_step && threadIdx.x == 1023) {
    // optional passenger (ogl_ce_fwd_bwd_mean_gather with step_dev): the optimiser's per-step scalars — what k_adam_prepare does as a launch of
    // its own at the END of the step — computed by one otherwise idle thread of the loss launch
    const int64_t t = *adam_step + 1;
    *adam_step = t;
    adam_scal[0] = (float)(adam_lr / (1.0 - pow(adam_b1, (double)t)));
    adam_scal[1] = (float)(1.0 / sqrt(1.0 - pow(adam_b2, (double)t)));
  }
  for (int i = threadIdx.x; i < zero_n4; i += 1024) zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // (a small scatter target: see the grid form)
  for (int row = wv; row < B; row += 16) {
    const float* x = logits + (int64_t)row * ldl;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    int64_t y;
    if (label_ids) {                                           // the label gather of ogl_gather_i64 (an id outside the table: -1)
      const int64_t id = label_ids[row];
      y = (id >= 0 && id < n_labels) ? labels[id] : -1;
    } else y = labels[row];
    const bool ok = y >= 0 && y < C;
    const float l = ok ? lse - x[y] : 0.f;
    if (lane == 0) { rows[row] = l; if (loss_rows) loss_rows[row] = l; }
    if (dlogits) {
      float* g = dlogits + (int64_t)row * lddl;
      for (int c = lane; c < C; c += 64) g[c] = grad_scale * (expf(x[c] - lse) - ((ok && c == (int)y) ? 1.f : 0.f));
    }
  }
  __syncthreads();
  if (wv == 0) {                                                  // fixed order: lane l sums rows l, l + 64, ...; then the lanes
    float t = 0.f;
    for (int r = lane; r < B; r += 64) t += rows[r];
    t = wave_sum(t);
    if (lane == 0) *loss_mean = t / (float)B;
  }
}

// The same with the label gather inside the launch (label of row i = label_table[label_ids[i]]; label_ids null: label_table IS the
// label vector) and an optional zero fill of a SMALL caller buffer (zero_floats <= 65 536: the one workgroup clears it; the atomic-
// scatter target of the backward pass that follows) — on the 32-seed rungs each of those was a ~5 us launch of a ~0.2 ms step.
#define CE_SMALL_MAX_ZERO 65536
extern "C" int ogl_ce_fwd_bwd_mean_gather(const float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels,
                                          const int64_t* label_ids, int64_t B, int C, float grad_scale, float* loss_rows, float* dlogits,
                                          int64_t lddl, float* loss_mean, float* zero_buf, int64_t zero_floats, int64_t* step_dev,
                                          float* scalars_dev, double lr, double beta1, double beta2, ogl_stream_t stream) {
  if (B <= 0 || B > CE_SMALL_MAX_B || C <= 0 || ldl < C || (dlogits && lddl < C) || n_labels < 0) return OGL_EINVAL;
  if (!logits || !label_table || !loss_mean || ((step_dev == nullptr) != (scalars_dev == nullptr))) return OGL_EINVAL;
  if (zero_floats < 0 || zero_floats > CE_SMALL_MAX_ZERO || (zero_floats > 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) || (zero_floats & 3))))
    return OGL_EINVAL;
  // step_dev / scalars_dev (both or neither): the optimiser's per-step scalars ride along (the 32-seed steps have no weight-image
  // launch for them to ride in: ogl_x3_split_multi): ++*step_dev; scalars_dev[0] = lr / (1 - beta1^t), scalars_dev[1] = 1 / sqrt(1 -
  // beta2^t).  The optimiser launch of the same step then runs with prepare = 0.
  hipLaunchKernelGGL(k_ce_fwd_bwd_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, ldl, label_table, (int)B, C, grad_scale,
                     loss_rows, dlogits, lddl, loss_mean, label_ids, n_labels, (float4*)zero_buf, (int)(zero_floats / 4), step_dev,
                     scalars_dev, lr, beta1, beta2);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// Any batch size, the mean from the same launch: the multi-block kernel above plus (a) the mean of the row losses, summed by the
// LAST block to finish in the same fixed order as the one-workgroup kernel (lane l takes rows l, l + 64, ...; then the lanes) — the
// blocks count themselves on `counter` (a zeroed device word the last block resets, so the caller allocates it once), and
// (b) an optional zero fill of a caller buffer by the same grid (the scatter target of the backward pass that follows needs
// zeros; as a launch of its own that fill is a ~9 us hole in a ~1 ms step).  Replaces rows.mean() as a second launch.
__global__ void __launch_bounds__(256) k_ce_fwd_bwd_mean_grid(const float* __restrict__ logits, int64_t ldl,
                                                              const int64_t* __restrict__ labels, int64_t B, int C, float grad_scale,
                                                              float* __restrict__ loss_rows, float* __restrict__ dlogits, int64_t lddl,
                                                              float* __restrict__ loss_mean, unsigned* __restrict__ counter,
                                                              int row_blocks, float4* __restrict__ zero_buf, int64_t zero_n4,
                                                              const int64_t* __restrict__ label_ids, int64_t n_labels) {
  if (zero_n4 > 0) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n4; i += (int64_t)gridDim.x * blockDim.x) zero_buf[i] = z;
  }
  if ((int)blockIdx.x >= row_blocks) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < B) {
    const float* x = logits + row * ldl;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    int64_t y;
    if (label_ids) {                                           // the label gather of ogl_gather_i64 (an id outside the table: -1)
      const int64_t id = label_ids[row];
      y = (id >= 0 && id < n_labels) ? labels[id] : -1;
    } else y = labels[row];
    const bool ok = y >= 0 && y < C;
    if (lane == 0) loss_rows[row] = ok ? lse - x[y] : 0.f;
    if (dlogits) {
      float* g = dlogits + row * lddl;
      for (int c = lane; c < C; c += 64) g[c] = grad_scale * (expf(x[c] - lse) - ((ok && c == (int)y) ? 1.f : 0.f));
    }
  }
  __shared__ int last;
  __threadfence();                                             // this block's row losses are visible device-wide ...
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == (unsigned)row_blocks - 1u;   // ... before it counts itself
  __syncthreads();
  if (last && threadIdx.x < 64) {
    __threadfence();
    float t = 0.f;
    for (int64_t r = lane; r < B; r += 64) t += __builtin_nontemporal_load(loss_rows + r);
    t = wave_sum(t);
    if (lane == 0) { *loss_mean = t / (float)B; *counter = 0u; }
  }
}

static int ce_mean_grid(const float* logits, int64_t ldl, const int64_t* labels, const int64_t* label_ids, int64_t n_labels, int64_t B, int C,
                        float grad_scale, float* loss_rows, float* dlogits, int64_t lddl, float* loss_mean, unsigned int* counter,
                        float* zero_buf, int64_t zero_floats, ogl_stream_t stream) {
  if (B <= 0 || C <= 0 || ldl < C || (dlogits && lddl < C) || zero_floats < 0 || n_labels < 0) return OGL_EINVAL;
  if (!logits || !labels || !loss_rows || !loss_mean || !counter) return OGL_EINVAL;
  if (zero_floats > 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) || (zero_floats & 3))) return OGL_EINVAL;
  const int row_blocks = (int)ogl_cdiv(B, 4);
  const int64_t fill_blocks = std::min<int64_t>(1024, ogl_cdiv(zero_floats / 4, 1024));
  hipLaunchKernelGGL(k_ce_fwd_bwd_mean_grid, dim3((unsigned)std::max<int64_t>(row_blocks, fill_blocks)), dim3(256), 0,
                     (hipStream_t)stream, logits, ldl, labels, B, C, grad_scale, loss_rows, dlogits, lddl, loss_mean, counter,
                     row_blocks, (float4*)zero_buf, zero_floats / 4, label_ids, n_labels);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// label_ids NULL: label_table holds the B labels themselves (n_labels ignored); else the labels are gathered inside the launch: label of
// row i = label_table[label_ids[i]] (an id outside [0, n_labels): no label, as ogl_gather_i64 writes -1) — graph.ndata['target'][seeds]
// (R/train/graphsage/pytorch/model.py:91,183) without a launch.
extern "C" int ogl_ce_fwd_bwd_mean_grid(const float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels,
                                        const int64_t* label_ids, int64_t B, int C, float grad_scale, float* loss_rows,
                                        float* dlogits, int64_t lddl, float* loss_mean, unsigned int* counter, float* zero_buf,
                                        int64_t zero_floats, ogl_stream_t stream) {
  return ce_mean_grid(logits, ldl, label_table, label_ids, label_ids ? n_labels : 0, B, C, grad_scale, loss_rows, dlogits, lddl, loss_mean, counter,
                      zero_buf, zero_floats, stream);
}

// The update of one element, with every rounding spelled out (explicit fused multiply-adds, no contraction left to the compiler):
// the kernels below inline this into different loops — plain gradients, gradients summed from split-K slabs, one or two launches per
// step — and all of them must produce the same bits from the same inputs.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float one_minus_b1, float b2, float one_minus_b2,
                                         float inv_sqrt_bc2, float step_size, float eps) {
#pragma clang fp contract(off)
  m = __fmaf_rn(one_minus_b1, g - m, m);                     // exp_avg.lerp_(g, 1 - beta1)
  v = __fmaf_rn(one_minus_b2 * g, g, v * b2);                // exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)
  const float denom = __fmaf_rn(sqrtf(v), inv_sqrt_bc2, eps);
  p = __fmaf_rn(-step_size, m / denom, p);
}

// torch.optim.Adam single-tensor form: m.lerp_(g, 1-b1); v = b2*v + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, int64_t n,
                                              float one_minus_b1, float b2, float one_minus_b2,
                                              float inv_sqrt_bc2, float step_size, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

extern "C" int ogl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                             double beta1, double beta2, double eps, ogl_stream_t stream) {
  if (n < 0 || step < 1) return OGL_EINVAL;
  if (n == 0) return OGL_OK;
  if (!p || !g || !m || !v) return OGL_EINVAL;
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(k_adam, dim3((unsigned)min((int64_t)2048, ogl_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     p, g, m, v, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), inv_sqrt_bc2, step_size,
                     (float)eps);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// All parameter tensors of the model in ONE launch (the reference's optimizer.step() loops over 12 tensors).
// Pointers travel by value in the kernel argument block: no device-side table, no extra copy.
#define OGL_ADAM_MAX_TENSORS 32
struct AdamBatch {
  float* p[OGL_ADAM_MAX_TENSORS];
  const float* g[OGL_ADAM_MAX_TENSORS];
  float* m[OGL_ADAM_MAX_TENSORS];
  float* v[OGL_ADAM_MAX_TENSORS];
  int64_t n[OGL_ADAM_MAX_TENSORS];
};

// ... with, per tensor, optional split-K SLABS of its gradient (ogl_linear_bwd_weight_x3k_slabs): the gradient of element
// (row r, column c) of a [rows, ncols] tensor is sum_s ws[s * slab_stride + r * ws_ld + col0 + c] in slab order — summed here, written
// to g (so that p.grad holds the gradient afterwards, as if a reduction launch had run) and applied.  ws null: g is read as usual.
struct AdamSlabs {
  const float* ws[OGL_ADAM_MAX_TENSORS];
  int64_t slab_stride[OGL_ADAM_MAX_TENSORS];
  int32_t ws_ld[OGL_ADAM_MAX_TENSORS], nsplit[OGL_ADAM_MAX_TENSORS], ncols[OGL_ADAM_MAX_TENSORS], col0[OGL_ADAM_MAX_TENSORS];
  // a TWO-RANGE tensor (the in-repo layer's concat weight [N, K1 + K2] behind ogl_linear_bwd_weight_x3k_dual_slabs): its columns
  // [0, split) sit at slab column col0, its columns [split, ncols) at slab column col0b; split == 0: one range
  int32_t split[OGL_ADAM_MAX_TENSORS], col0b[OGL_ADAM_MAX_TENSORS];
};

// one tensor, grid-strided over blockIdx.x: four elements per thread and trip (16-byte accesses) when the four arrays allow it
__device__ __forceinline__ void adam_range(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                                           float inv_sqrt_bc2, float step_size, float eps) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  int64_t done = 0;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 p4 = ((float4*)p)[i], m4 = ((float4*)m)[i], v4 = ((float4*)v)[i];
      const float4 g4 = ((const float4*)g)[i];
      adam_one(p4.x, g4.x, m4.x, v4.x, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(p4.y, g4.y, m4.y, v4.y, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(p4.z, g4.z, m4.z, v4.z, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(p4.w, g4.w, m4.w, v4.w, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      ((float4*)m)[i] = m4; ((float4*)v)[i] = v4; ((float4*)p)[i] = p4;
    }
    done = n4 << 2;
  }
  for (int64_t i = done + tid; i < n; i += nth) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

// a slab-backed tensor: one thread per 4 consecutive ELEMENTS of the tensor (its own arrays in aligned 16-byte accesses, fully
// coalesced); the elements' slab entries are 4 consecutive floats of a slab row (rows are ws_ld floats apart, the tensor's ncols:
// only 4-byte alignment in common — one unaligned 16-byte load per slab, gfx950 runs in unaligned-access mode) unless the four
// straddle a row end (once per row: element by element).  Slab order per element: the bits of k_x3_splitk_reduce.
__device__ __forceinline__ float slab_sum1(const float* __restrict__ src, int64_t slab_stride, int nsplit) {
  float a = 0.f;
  for (int s = 0; s < nsplit; ++s) a += src[(int64_t)s * slab_stride];
  return a;
}

__device__ __forceinline__ void adam_range_slabs(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t n, const float* __restrict__ ws, int64_t slab_stride,
                                                 int ws_ld, int nsplit, int ncols, int col0, float one_minus_b1, float b2,
                                                 float one_minus_b2, float inv_sqrt_bc2, float step_size, float eps, int split = 0,
                                                 int col0b = 0) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  auto scol = [&](int col) __attribute__((always_inline)) { return (split == 0 || col < split) ? col0 + col : col0b + (col - split); };
  int64_t done = 0;
  if (ncols >= 4 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const int64_t e0 = i << 2;
      const int64_t row = e0 / ncols;
      const int col = (int)(e0 - row * ncols);
      float a[4];
      // (the parameter's own loads go out first, then the slabs four at a time: a thread has ONE group of four elements, so every
      // load it does not overlap is latency it pays in full — the launch sits at the end of the step, nothing runs beside it.
      // The sum keeps the slab order: the reduction launch's bits.)
      const float4 p4 = ((const float4*)p)[i], m4 = ((const float4*)m)[i], v4 = ((const float4*)v)[i];
      if (col + 4 <= ncols && (split == 0 || col + 4 <= split || col >= split)) {
        const float* src = ws + row * ws_ld + scol(col);
        a[0] = a[1] = a[2] = a[3] = 0.f;
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
          float4 t[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) t[u] = ld16(src + (int64_t)(s + u) * slab_stride);
#pragma unroll
          for (int u = 0; u < 4; ++u) { a[0] += t[u].x; a[1] += t[u].y; a[2] += t[u].z; a[3] += t[u].w; }
        }
        for (; s < nsplit; ++s) {
          const float4 t = ld16(src + (int64_t)s * slab_stride);
          a[0] += t.x; a[1] += t.y; a[2] += t.z; a[3] += t.w;
        }
      } else {                                                 // the four straddle a row end (one group per row; its wave waits for it)
        const float* q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t r = (e0 + e) / ncols;
          q[e] = ws + r * ws_ld + scol((int)(e0 + e - r * ncols));
          a[e] = 0.f;
        }
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
          float t[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[u][e] = q[e][(int64_t)(s + u) * slab_stride];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += t[u][e];
        }
        for (; s < nsplit; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] += q[e][(int64_t)s * slab_stride];
      }
      float4 pn = p4, mn = m4, vn = v4;
      adam_one(pn.x, a[0], mn.x, vn.x, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(pn.y, a[1], mn.y, vn.y, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(pn.z, a[2], mn.z, vn.z, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      adam_one(pn.w, a[3], mn.w, vn.w, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
      ((float4*)g)[i] = make_float4(a[0], a[1], a[2], a[3]);
      ((float4*)m)[i] = mn; ((float4*)v)[i] = vn; ((float4*)p)[i] = pn;
    }
    done = n4 << 2;
  }
  for (int64_t i = done + tid; i < n; i += nth) {
    const int64_t row = i / ncols;
    const float a = slab_sum1(ws + row * ws_ld + scol((int)(i - row * ncols)), slab_stride, nsplit);
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, a, mi, vi, one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
    g[i] = a; m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

__global__ void __launch_bounds__(256) k_adam_multi_slabs(AdamBatch b, AdamSlabs sl, float one_minus_b1, float b2, float one_minus_b2,
                                                          float inv_sqrt_bc2_host, float step_size_host, const float* __restrict__ scal,
                                                          float eps) {
  // scal (device): {step size, 1 / sqrt(1 - beta2^t)} of the device-side step count; null: the host's values
  const float step_size = scal ? scal[0] : step_size_host, inv_sqrt_bc2 = scal ? scal[1] : inv_sqrt_bc2_host;
  const int t = blockIdx.y;
  if (sl.ws[t])
    adam_range_slabs(b.p[t], (float*)b.g[t], b.m[t], b.v[t], b.n[t], sl.ws[t], sl.slab_stride[t], sl.ws_ld[t], sl.nsplit[t],
                     sl.ncols[t], sl.col0[t], one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps, sl.split[t], sl.col0b[t]);
  else
    adam_range(b.p[t], b.g[t], b.m[t], b.v[t], b.n[t], one_minus_b1, b2, one_minus_b2, inv_sqrt_bc2, step_size, eps);
}

// ---- Adam with a DEVICE-side step count: replayable inside a captured hipGraph -------------------------------------------
// (a host-side step would freeze the bias corrections into the graph's kernel arguments).  k_adam_prepare increments the
// counter and writes {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} in double arithmetic, rounded to fp32 once, like the host path.
__global__ void k_adam_prepare(int64_t* step, double lr, double beta1, double beta2, float* scal) {
  const int64_t t = *step + 1;
  *step = t;
  scal[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
  scal[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)t)));
}

// Adam over tensors some of whose gradients still are split-K slabs (AdamSlabs; all-null slab pointers: plain Adam).
//   step_dev == null: host step count `step` (as ogl_adam_step_multi);
//   step_dev != null: the device-side count (as ogl_adam_step_multi_dev); `prepare` != 0 increments it and refreshes the two
//                     scalars first, prepare == 0 applies with the scalars as they are — a step's SECOND launch (the optimiser
//                     applied in two parts: everything whose gradient is ready early on a side branch, the rest at the end).
// ws / slab_stride / ws_ld / nsplit / ncols / col0: host arrays of `count` entries (ws[i] null: tensor i has a plain gradient).
static int adam_step_multi_slabs(int count, float* const* p, float* const* g, float* const* m, float* const* v, const int64_t* n,
                                 const float* const* ws, const int64_t* slab_stride, const int* ws_ld, const int* nsplit,
                                 const int* ncols, const int* col0, const int* split, const int* col0b, int step, int64_t* step_dev,
                                 float* scalars_dev, int prepare, double lr, double beta1, double beta2, double eps, ogl_stream_t stream) {
  if (count < 0 || (!step_dev && step < 1) || (step_dev && !scalars_dev)) return OGL_EINVAL;
  if (count > 0 && (!p || !g || !m || !v || !n || !ws || !slab_stride || !ws_ld || !nsplit || !ncols || !col0)) return OGL_EINVAL;
  float step_size = 0.f, inv_sqrt_bc2 = 0.f;
  if (step_dev) {
    if (prepare) {
      hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr, beta1, beta2, scalars_dev);
      OGL_CHECK_LAUNCH();
    }
  } else {
    step_size = (float)(lr / (1.0 - pow(beta1, step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, step)));
  }
  for (int base = 0; base < count; base += OGL_ADAM_MAX_TENSORS) {
    AdamBatch b;
    AdamSlabs sl;
    const int c = min(OGL_ADAM_MAX_TENSORS, count - base);
    int64_t nmax = 0;
    for (int i = 0; i < c; ++i) {
      const int j = base + i;
      if (n[j] < 0 || (n[j] > 0 && (!p[j] || !g[j] || !m[j] || !v[j]))) return OGL_EINVAL;
      b.p[i] = p[j]; b.g[i] = g[j]; b.m[i] = m[j]; b.v[i] = v[j]; b.n[i] = n[j];
      sl.ws[i] = ws[j]; sl.slab_stride[i] = slab_stride[j]; sl.ws_ld[i] = ws_ld[j]; sl.nsplit[i] = nsplit[j]; sl.ncols[i] = ncols[j];
      sl.col0[i] = col0[j];
      sl.split[i] = split ? split[j] : 0; sl.col0b[i] = col0b ? col0b[j] : 0;
      if (ws[j]) {
        const int sp = sl.split[i];
        if (nsplit[j] < 1 || ncols[j] < 1 || col0[j] < 0 || n[j] % ncols[j] != 0 || slab_stride[j] < (n[j] / ncols[j]) * (int64_t)ws_ld[j])
          return OGL_EINVAL;
        if (sp == 0 ? ws_ld[j] < col0[j] + ncols[j]
                    : (sp < 0 || sp >= ncols[j] || sl.col0b[i] < col0[j] + sp || ws_ld[j] < sl.col0b[i] + (ncols[j] - sp)))
          return OGL_EINVAL;
      }
      nmax = n[j] > nmax ? n[j] : nmax;
    }
    for (int i = c; i < OGL_ADAM_MAX_TENSORS; ++i) sl.ws[i] = nullptr;
    if (nmax == 0) continue;
    dim3 grid((unsigned)min((int64_t)512, ogl_cdiv(nmax, 1024)), (unsigned)c);
    hipLaunchKernelGGL(k_adam_multi_slabs, grid, dim3(256), 0, (hipStream_t)stream, b, sl, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), inv_sqrt_bc2, step_size, (const float*)(step_dev ? scalars_dev : nullptr), (float)eps);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}

// TWO-RANGE tensors (split / col0b, both nullable): tensor i with split[i] > 0 has its columns [0, split) at slab column col0[i] and its columns
// [split, ncols) at slab column col0b[i] — the concat weight [N, K1 + K2] of the in-repo layer behind
// ogl_linear_bwd_weight_x3k_dual_slabs (slab layout [dw1 | db | pad | dw2]): no reduction launch between the product and the optimiser
extern "C" int ogl_adam_step_multi_slabs(int count, float* const* p, float* const* g, float* const* m, float* const* v, const int64_t* n,
                                          const float* const* ws, const int64_t* slab_stride, const int* ws_ld, const int* nsplit,
                                          const int* ncols, const int* col0, const int* split, const int* col0b, int step,
                                          int64_t* step_dev, float* scalars_dev, int prepare, double lr, double beta1, double beta2,
                                          double eps, ogl_stream_t stream) {
  return adam_step_multi_slabs(count, p, g, m, v, n, ws, slab_stride, ws_ld, nsplit, ncols, col0, split, col0b, step, step_dev,
                               scalars_dev, prepare, lr, beta1, beta2, eps, stream);
}

// Evaluation on the device: argmax of every logits row (first maximum, like numpy) and
// the C x C confusion matrix confusion[true][pred], so only C*C counters cross PCIe instead of [n, C] logits.
// Replaces output_data.argmax(axis=1) + sklearn.metrics.confusion_matrix (R/train/graphsage/model.py:84-87).
__global__ void __launch_bounds__(256) k_argmax_confusion(const float* __restrict__ logits, int64_t ldl,
                                                          const int64_t* __restrict__ labels, int64_t B, int C,
                                                          int64_t* __restrict__ pred, unsigned long long* __restrict__ confusion) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* x = logits + row * ldl;
  float best = -INFINITY;
  int arg = C;                                         // lanes without a column never win
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > best || (arg == C)) { if (v > best || arg == C) { best = v; arg = c; } }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oa = __shfl_xor(arg, o);
    if (oa < C && (arg >= C || ob > best || (ob == best && oa < arg))) { best = ob; arg = oa; }
  }
  if (lane == 0) {
    if (pred) pred[row] = arg;
    const int64_t y = labels ? labels[row] : -1;
    if (confusion && y >= 0 && y < C && arg < C) atomicAdd(&confusion[y * C + arg], 1ull);
  }
}

extern "C" int ogl_argmax_confusion(const float* logits, int64_t ldl, const int64_t* labels, int64_t B, int C,
                                    int64_t* pred, int64_t* confusion, ogl_stream_t stream) {
  if (B < 0 || C <= 0 || ldl < C) return OGL_EINVAL;
  if (B == 0) return OGL_OK;
  if (!logits || (confusion && !labels)) return OGL_EINVAL;
  hipLaunchKernelGGL(k_argmax_confusion, dim3((unsigned)ogl_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, ldl, labels,
                     B, C, pred, (unsigned long long*)confusion);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
