// reduce.hip -- HBM-bound reductions.  One primitive covers per-channel sums / dot products over [N, C, HW]
// (bias and batch-norm parameter gradients), per-example dot products over C*H*W (the gradient-penalty
// norms: N = 1, C = batch), column sums over the batch (feature means: HW = 1) and full sums (C = 1).
// Each wavefront reduces with cross-lane shuffles (64 lanes), each workgroup with 4 LDS words, and
// workgroups combine with one hardware fp32 atomic per (channel, segment) -- or, for up to 64 rows (the penalty's
// per-example norms, full sums), through workspace partials that the row's last workgroup adds in a fixed order.
// Roofline: HBM; algorithmic bytes = 4 * elements read per operand.
#include "common.h"
#include "split_finish.h"
#include <string.h>

namespace srgan {

float* partial_workspace(size_t bytes, hipStream_t stream);     // gather_gemm_kernels.hip: the caller's per-stream workspace
int workspace_index(hipStream_t stream);                         // its small integer id (-1: none registered)

constexpr int RED_SEG = 256 * 16;   // smallest run of one row handled by one workgroup (a multiple of 1024 elements)

// out[c] += scale[c] * sum over rows r = n*C + c, i in [0, HW) of a[r, i] * ((b ? b[r, i] : 1) - mean[c])
// (mean, scale optional: with them this is the batch-norm gamma gradient in one pass).
// One workgroup reduces a run of `seg` elements of one row: four float4 loads per thread in flight, wave64 shuffle
// reduction, one atomic.  The host picks `seg` so that long rows -- the per-example squared norms of the gradient penalty
// over 3 x 512 x 512 elements (reference srgan.py:371) -- are cut into a few thousand workgroups' worth of long runs
// instead of 16 KB crumbs whose reduction and atomic cost as much as their loads (round 3: 2.2 TB/s).
__device__ __forceinline__ float row_run_sum(const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ mean, int C, int64_t HW, int segs, int64_t seg,
                                             float* scratch) {      // valid in thread 0
  const int c = blockIdx.x;
  const int n = blockIdx.y / segs, part = blockIdx.y - n * segs;
  const int64_t base = ((int64_t)n * C + c) * HW, beg = (int64_t)part * seg;
  const int64_t end = beg + seg < HW ? beg + seg : HW;
  const float mu = mean ? mean[c] : 0.f;
  float acc = 0.f;
  if (((base | beg) & 3) == 0) {       // 16-byte aligned run: float4 loads
    const int64_t n4 = (end - beg) >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a + base + beg);
    const float4* b4 = b ? reinterpret_cast<const float4*>(b + base + beg) : nullptr;
    const bool square = a == b;          // a squared norm (the gradient penalty's): one load stream, not two
    float part_sum[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = threadIdx.x;
    if (square) {                         // (hoisted: a per-element select between a register and a load serialises the loads)
      for (; i + 768 < n4; i += 1024) {
        float4 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = a4[i + 256 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          part_sum[j] += u[j].x * (u[j].x - mu) + u[j].y * (u[j].y - mu) + u[j].z * (u[j].z - mu) + u[j].w * (u[j].w - mu);
      }
    } else {
      for (; i + 768 < n4; i += 1024) {
        float4 u[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = a4[i + 256 * j];
        if (b4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = b4[i + 256 * j];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            part_sum[j] += u[j].x * (v[j].x - mu) + u[j].y * (v[j].y - mu) + u[j].z * (v[j].z - mu) + u[j].w * (v[j].w - mu);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) part_sum[j] += (u[j].x + u[j].y + u[j].z + u[j].w) * (1.f - mu);
        }
      }
    }
    for (; i < n4; i += 256) {
      const float4 u = a4[i];
      if (b4) {
        const float4 v = b4[i];
        acc += u.x * (v.x - mu) + u.y * (v.y - mu) + u.z * (v.z - mu) + u.w * (v.w - mu);
      } else {
        acc += (u.x + u.y + u.z + u.w) * (1.f - mu);
      }
    }
    acc += (part_sum[0] + part_sum[1]) + (part_sum[2] + part_sum[3]);
    for (int64_t k = beg + (n4 << 2) + threadIdx.x; k < end; k += 256)
      acc += a[base + k] * ((b ? b[base + k] : 1.f) - mu);
  } else {
    for (int64_t k = beg + threadIdx.x; k < end; k += 256) acc += a[base + k] * ((b ? b[base + k] : 1.f) - mu);
  }
  return block_sum_256(acc, scratch);
}

__global__ __launch_bounds__(256) void chan_reduce_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ scale,
                                                               float* __restrict__ out, int C, int64_t HW, int segs,
                                                               int64_t seg) {
  __shared__ float scratch[4];
  const float total = row_run_sum(a, b, mean, C, HW, segs, seg, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + blockIdx.x, total * (scale ? scale[blockIdx.x] : 1.f));
}

// The same reduction for FEW rows (<= ROW_TICKETS: the per-example norms of the gradient penalty, full sums) in ONE launch
// and in a FIXED order: every workgroup leaves the sum of its run in `partial` (the caller's workspace), takes a ticket of its
// row, and the workgroup that draws the last one adds the row's partials -- thread t the entries t, t + 256, ..., then the
// usual tree -- and writes (or adds to) out[c].  No zero-fill launch in front, no fp32 atomics: two runs give the same bits,
// whatever else is in flight.  The tickets are device globals, one set per registered (device, stream) workspace -- launches
// on a stream are ordered, streams do not share a set -- and the last workgroup puts its row's back to zero.
constexpr int ROW_TICKETS = 2048, TICKET_SETS = 64;      // (round 5: 2048 rows -- bias / parameter sums of wide layers take the ordered form too)
__device__ unsigned int g_row_tickets[TICKET_SETS * ROW_TICKETS];

__device__ unsigned int g_reduce_finish_tickets[SPLIT_TICKET_SETS * ROW_FINISH_ROWS];

__global__ __launch_bounds__(256) void chan_reduce_rows_ordered_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ scale,
                                                                       float* __restrict__ out, float* __restrict__ partial,
                                                                       int ticket_set, int C, int64_t HW, int segs, int64_t seg,
                                                                       int accumulate) {
  __shared__ float scratch[4];
  __shared__ int last;
  const int c = blockIdx.x, parts = (int)gridDim.y;
  const float total = row_run_sum(a, b, mean, C, HW, segs, seg, scratch);
  float* row = partial + (int64_t)c * parts;
  unsigned int* ticket = g_row_tickets + ticket_set * ROW_TICKETS + c;
  if (threadIdx.x == 0) {
    // Device-scope atomics only (they act at the memory side: coherent across the eight XCDs' L2s by themselves); a
    // device-scope FENCE here would write back / invalidate a whole L2 per workgroup (measured: 76 us for the 30 us
    // reduction).  The partial (a write-through `sc1` store) must be ACKNOWLEDGED before the ticket is taken: store and
    // ticket live in different L2 channels, so without the wait the row's last workgroup -- possibly on another XCD --
    // could draw the final ticket and still read the previous launch's partial (a workgroup-scope release fence
    // emits no instruction at all here; the ISA had the atomic straight behind the store).  Stores count in vmcnt on gfx9.
    __hip_atomic_store(row + blockIdx.y, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(parts - 1);
  }
  __syncthreads();
  if (!last) return;
  float acc = 0.f;
  for (int i = threadIdx.x; i < parts; i += 256) acc += __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();                                     // (scratch is reused)
  const float sum = block_sum_256(acc, scratch);
  if (threadIdx.x == 0) {
    const float value = sum * (scale ? scale[c] : 1.f);
    out[c] = accumulate ? out[c] + value : value;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One workgroup per channel, no atomics and no pre-zeroing: used when there are enough channels to fill the chip.
// Optionally masked (terms with mask <= 0 dropped) and producing the plain sum of a next to the weighted one:
// the two parameter gradients of a frozen batch-norm (+ReLU) in ONE pass over g and x.
__global__ __launch_bounds__(256) void chan_reduce_block_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ scale,
                                                                float* __restrict__ out, float* __restrict__ out_plain,
                                                                int N, int C, int64_t HW, int accumulate) {
  __shared__ float scratch[4];
  const int c = blockIdx.x;
  const float mu = mean ? mean[c] : 0.f;
  float acc = 0.f, plain = 0.f;
  for (int n = 0; n < N; ++n) {
    const int64_t base = ((int64_t)n * C + c) * HW;
    for (int64_t i = threadIdx.x; i < HW; i += 256) {
      float g = a[base + i];
      if (mask) g = mask[base + i] > 0.f ? g : 0.f;
      acc += g * ((b ? b[base + i] : 1.f) - mu);
      plain += g;
    }
  }
  const float total = block_sum_256(acc, scratch);
  __syncthreads();
  const float total_plain = out_plain ? block_sum_256(plain, scratch) : 0.f;
  if (threadIdx.x == 0) {
    const float v = total * (scale ? scale[c] : 1.f);
    out[c] = accumulate ? out[c] + v : v;
    if (out_plain) out_plain[c] = accumulate ? out_plain[c] + total_plain : total_plain;
  }
}

// Whole backward of a frozen batch-norm (+ReLU) in ONE pass over (g, x): the activation mask is recomputed from x
// with the forward's own arithmetic (fma(x, a, b) > 0), the input gradient g*[y>0]*a is written (or accumulated
// into a channel-slice view), and both parameter gradients leave as two fp32 atomics per workgroup.  One workgroup
// per (channel, image) row.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_act_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ inv_std,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ gx,
                                                              float* __restrict__ out_gamma,
                                                              float* __restrict__ out_beta, int N, int C, int64_t HW,
                                                              int64_t g_bs, int64_t x_bs, int64_t gx_bs, int accumulate,
                                                              int unscaled, int images_per_block, float* finish_partial,
                                                              unsigned int* finish_tickets) {
  // One workgroup = channel c of images [n0, n1): several images per workgroup when the planes are small (a
  // 16 x 16 plane is one float4 per four lanes), so that every launch has ~1024 workgroups of useful size.
  __shared__ float scratch[4];
  const int c = blockIdx.x;
  const int n0 = (int)blockIdx.y * images_per_block, n1 = min(N, n0 + images_per_block);
  const float mu = mean[c];
  float a, b;
  bn_coefficients(mu, inv_std[c], gamma[c], beta[c], a, b);   // the forward's own (a, b): bit-identical mask
  const float os = unscaled ? 1.f : a;                        // output scale
  const int64_t plane = (int64_t)c * HW;
  float acc = 0.f, plain = 0.f;
  const bool nontemporal_store = !accumulate;          // a fresh gradient tensor: consumed by a later kernel, not this one
  const bool vec = ((g_bs | x_bs | gx_bs | HW) & 3) == 0 &&
                   (((uintptr_t)g | (uintptr_t)x | (uintptr_t)gx) & 15) == 0;
  if (vec) {
    const int hw4 = (int)(HW >> 2);
    const int total = (n1 - n0) * hw4;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      const int nl = idx / hw4, i = idx - nl * hw4;
      const int n = n0 + nl;
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f graw = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g + (int64_t)n * g_bs + plane) + i);   // read once
      float4 gv = make_float4(graw.x, graw.y, graw.z, graw.w);
      const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)n * x_bs + plane)[i];
      if (RELU) {
        gv.x = fmaf(xv.x, a, b) > 0.f ? gv.x : 0.f; gv.y = fmaf(xv.y, a, b) > 0.f ? gv.y : 0.f;
        gv.z = fmaf(xv.z, a, b) > 0.f ? gv.z : 0.f; gv.w = fmaf(xv.w, a, b) > 0.f ? gv.w : 0.f;
      }
      acc += gv.x * (xv.x - mu) + gv.y * (xv.y - mu) + gv.z * (xv.z - mu) + gv.w * (xv.w - mu);
      plain += gv.x + gv.y + gv.z + gv.w;
      if (gx) {
        float4* o4 = reinterpret_cast<float4*>(gx + (int64_t)n * gx_bs + plane) + i;
        float4 o = make_float4(gv.x * os, gv.y * os, gv.z * os, gv.w * os);
        if (accumulate) { const float4 old = *o4; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        if (nontemporal_store) {
          v4f oraw; oraw.x = o.x; oraw.y = o.y; oraw.z = o.z; oraw.w = o.w;
          __builtin_nontemporal_store(oraw, reinterpret_cast<v4f*>(o4));
        } else {
          *o4 = o;
        }
      }
    }
  } else {
    for (int n = n0; n < n1; ++n) {
      const int64_t base = (int64_t)n * g_bs + plane, xbase = (int64_t)n * x_bs + plane, obase = (int64_t)n * gx_bs + plane;
      for (int64_t i = threadIdx.x; i < HW; i += 256) {
        float gv = g[base + i];
        const float xv = x[xbase + i];
        if (RELU) gv = fmaf(xv, a, b) > 0.f ? gv : 0.f;
        acc += gv * (xv - mu);
        plain += gv;
        if (gx) gx[obase + i] = accumulate ? gx[obase + i] + gv * os : gv * os;
      }
    }
  }
  if (out_gamma == nullptr) return;
  const float total_acc = block_sum_256(acc, scratch);
  __syncthreads();
  const float total_plain = block_sum_256(plain, scratch);
  if (finish_partial) {      // ordered: the channel's image groups meet in a fixed order, ONE adder per channel (split_finish.h)
    float v[2] = {total_acc, total_plain};
    __syncthreads();
    if (ordered_row_finish<2>(v, finish_partial + (int64_t)c * gridDim.y * 2, (int)blockIdx.y, (int)gridDim.y, finish_tickets + c,
                              scratch)) {
      out_gamma[c] += v[0] * inv_std[c];
      out_beta[c] += v[1];
    }
    return;
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(out_gamma + c, total_acc * inv_std[c]);
    unsafeAtomicAdd(out_beta + c, total_plain);
  }
}

// HW == 1: a is [N, C]; 32 columns x 8 row-lanes per workgroup (lanes along c: coalesced 128-byte rows), the rows
// strided over the 8 row-lanes and combined through LDS -- a single thread per column would walk N dependent loads.
__global__ __launch_bounds__(256) void chan_reduce_cols_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ scale,
                                                               float* __restrict__ out, int N, int C, int accumulate) {
  __shared__ float scratch[8][33];
  const int col = (int)threadIdx.x & 31, rl = (int)threadIdx.x >> 5;
  const int c = (int)blockIdx.x * 32 + col;
  const float mu = (mean && c < C) ? mean[c] : 0.f;
  float acc = 0.f;
  if (c < C)
    for (int n = rl; n < N; n += 8) {
      const int64_t i = (int64_t)n * C + c;
      acc += a[i] * ((b ? b[i] : 1.f) - mu);
    }
  scratch[rl][col] = acc;
  __syncthreads();
  if (rl != 0 || c >= C) return;
#pragma unroll
  for (int r = 1; r < 8; ++r) acc += scratch[r][col];
  acc *= scale ? scale[c] : 1.f;
  out[c] = accumulate ? out[c] + acc : acc;
}

// Double backward of norm -> relu -> conv (fused.py, reference srgan.py:360-375 through crowd/models.py:338-345):
// everything that is per weight element, in one launch over w[CO][inner] (inner = CI * taps, channel = j / taps):
//   w_scaled = w * a[ci]                        (the linearised forward runs on the masked UNSCALED tangent)
//   w_grad  += q * a[ci]                        (q = weight gradient w.r.t. the scaled weights)
//   gamma_grad[ci] += inv_std[ci] * sum_{co, tap} w * q
// Lanes along the contiguous inner index.  Round 5: a workgroup covers a span of inner indices that is a WHOLE number of
// channels (64 - 64 % taps: 64 for the 1x1, 63 = 7 channels for the 3x3 convolutions) and ALL rows of CO -- 16 at a time, four
// row-lanes x four independent iterations (the chain of dependent loads, not bandwidth, bounds this tiny kernel) -- so that a
// channel's gamma gradient is completed inside ONE workgroup in a fixed order and added once: no atomics (round 4: one fp32
// atomic per (channel, tap, 16-row block), in arrival order).
constexpr int TW_ROWS = 16;
__device__ __forceinline__ void tangent_weight_body(const float* __restrict__ w, const float* __restrict__ q,
                                                    const float* __restrict__ inv_std, const float* __restrict__ gamma,
                                                    float* __restrict__ w_scaled, float* __restrict__ w_grad,
                                                    float* __restrict__ gamma_grad, int CO, int inner, int taps,
                                                    float (*scratch)[64]) {
  const int jl = (int)threadIdx.x & 63, cl = (int)threadIdx.x >> 6;
  const int span = 64 - 64 % taps;                              // (taps <= 64, checked by the launchers)
  const int j = (int)blockIdx.x * span + jl;
  const bool live = jl < span && j < inner;
  const int ci = live ? j / taps : 0;
  const float a = __fmul_rn(inv_std[ci], gamma[ci]);
  float dot = 0.f;
  for (int row0 = 0; row0 < CO; row0 += TW_ROWS) {
    const int co0 = row0 + cl;
    float wv[TW_ROWS / 4], qv[TW_ROWS / 4], gv[TW_ROWS / 4];
#pragma unroll
    for (int e = 0; e < TW_ROWS / 4; ++e) {                     // all loads first
      const int co = co0 + 4 * e;
      const bool ok = live && co < CO;
      const int64_t at = ok ? (int64_t)co * inner + j : 0;
      wv[e] = ok ? w[at] : 0.f;
      qv[e] = (ok && q) ? q[at] : 0.f;
      gv[e] = (ok && q) ? w_grad[at] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < TW_ROWS / 4; ++e) {
      const int co = co0 + 4 * e;
      if (!live || co >= CO) continue;
      const int64_t at = (int64_t)co * inner + j;
      if (w_scaled) w_scaled[at] = wv[e] * a;
      if (q) {
        w_grad[at] = gv[e] + qv[e] * a;
        dot = fmaf(wv[e], qv[e], dot);
      }
    }
  }
  if (q == nullptr) return;
  scratch[cl][jl] = live ? dot : 0.f;
  __syncthreads();
  if (cl == 0) scratch[0][jl] = (scratch[0][jl] + scratch[1][jl]) + (scratch[2][jl] + scratch[3][jl]);     // (own column only)
  __syncthreads();
  if (cl != 0 || !live || j % taps != 0) return;                // the thread of a channel's first tap adds its taps in order
  float total = scratch[0][jl];
  for (int t = 1; t < taps; ++t) total += scratch[0][jl + t];   // (the span holds whole channels: jl + t < span)
  gamma_grad[ci] += total * inv_std[ci];
}

__global__ __launch_bounds__(256) void tangent_weight_kernel(const float* __restrict__ w, const float* __restrict__ q,
                                                             const float* __restrict__ inv_std,
                                                             const float* __restrict__ gamma, float* __restrict__ w_scaled,
                                                             float* __restrict__ w_grad, float* __restrict__ gamma_grad,
                                                             int CO, int inner, int taps) {
  __shared__ float scratch[4][64];
  tangent_weight_body(w, q, inv_std, gamma, w_scaled, w_grad, gamma_grad, CO, inner, taps, scratch);
}

// The same for MANY convolutions in one launch (every norm -> relu -> conv of a dense block's double backward): job
// blockIdx.z of a device-resident table; the scaled weights / the weight gradients q of all jobs live in two per-step
// buffers at the job's `offset` (elements), everything else is where the parameters live.  `scaled_base` != NULL: write
// the scaled weights (start of the double backward); `q_base` != NULL: the gradient part (its end).
struct TangentWeightJob {
  const float* w; const float* inv_std; const float* gamma; float* w_grad; float* gamma_grad;
  int64_t offset;
  int32_t CO, inner, taps, pad;
};
static_assert(sizeof(TangentWeightJob) == 64, "one 64-byte table slot per job");

__global__ __launch_bounds__(256) void tangent_weight_grouped_kernel(const TangentWeightJob* __restrict__ jobs,
                                                                     float* __restrict__ scaled_base,
                                                                     const float* __restrict__ q_base) {
  __shared__ float scratch[4][64];
  const TangentWeightJob job = jobs[blockIdx.z];
  if ((int64_t)blockIdx.x * (64 - 64 % job.taps) >= job.inner) return;                                // (workgroup-uniform)
  tangent_weight_body(job.w, q_base ? q_base + job.offset : nullptr, job.inv_std, job.gamma,
                      scaled_base ? scaled_base + job.offset : nullptr, job.w_grad, job.gamma_grad, job.CO, job.inner,
                      job.taps, scratch);
}

// out[b] = max_f x[b, f]  (the stabiliser of logsumexp, reference utility.py:179); F is small (bins).
__global__ __launch_bounds__(256) void row_max_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int F) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float best = x[(int64_t)b * F];
  for (int f = 1; f < F; ++f) best = fmaxf(best, x[(int64_t)b * F + f]);
  out[b] = best;
}

// onehot[b, k] = 1 at the first k minimising |y[b] - bins[k]|, else 0  (reference utility.py:141-144)
__global__ __launch_bounds__(256) void nearest_bin_onehot_kernel(const float* __restrict__ y,
                                                                 const float* __restrict__ bins,
                                                                 float* __restrict__ onehot, int B, int K) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int best = 0;
  float best_d = fabsf(y[b] - bins[0]);
  for (int k = 1; k < K; ++k) {
    const float d = fabsf(y[b] - bins[k]);
    if (d < best_d) { best_d = d; best = k; }
  }
  for (int k = 0; k < K; ++k) onehot[(int64_t)b * K + k] = k == best ? 1.f : 0.f;
}

// rows[b] += sum_{hw} (1/Cm) * sum_c |maps[b,c,hw] - target[b,hw]|      (reference crowd/srgan.py:252)
__global__ __launch_bounds__(256) void crowd_map_l1_kernel(const float* __restrict__ maps,
                                                           const float* __restrict__ target, float* __restrict__ rows,
                                                           int Cm, int64_t HW, int segs, float* finish_partial,
                                                           unsigned int* finish_tickets) {
  __shared__ float scratch[4];
  const int b = blockIdx.x, seg = blockIdx.y;
  const int64_t beg = (int64_t)seg * RED_SEG, end = beg + RED_SEG < HW ? beg + RED_SEG : HW;
  float acc = 0.f;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    const float t = target[(int64_t)b * HW + i];
    float s = 0.f;
    for (int c = 0; c < Cm; ++c) s += fabsf(maps[((int64_t)b * Cm + c) * HW + i] - t);
    acc += s / (float)Cm;
  }
  const float total = block_sum_256(acc, scratch);
  if (finish_partial) {      // ordered: an example's segments meet in a fixed order (this sum IS the labeled loss's map term)
    float v[1] = {total};
    __syncthreads();
    if (ordered_row_finish<1>(v, finish_partial + (int64_t)b * segs, seg, segs, finish_tickets + b, scratch)) rows[b] += v[0];
    return;
  }
  if (threadIdx.x == 0) unsafeAtomicAdd(rows + b, total);
}

// gmaps[b,c,hw] = g[b] * sign(maps[b,c,hw] - target[b,hw]) / Cm
__global__ __launch_bounds__(256) void crowd_map_l1_bwd_kernel(const float* __restrict__ maps,
                                                               const float* __restrict__ target,
                                                               const float* __restrict__ g, float* __restrict__ gmaps,
                                                               int Cm, int64_t HW, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256, image = (int64_t)Cm * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int64_t b = i / image, hw = (i - b * image) % HW;
    const float d = maps[i] - target[b * HW + hw];
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    gmaps[i] = g[b] * sgn / (float)Cm;
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" {

int srgan_chan_reduce(const float* a, const float* b, const float* mean, const float* scale, float* out, int32_t N,
                      int32_t C, int64_t HW, int accumulate, void* stream) {
  SRGAN_REQUIRE(a && out && N > 0 && C > 0 && HW > 0, SRGAN_EINVAL, "srgan_chan_reduce arguments");
  hipStream_t s = (hipStream_t)stream;
  if (HW == 1) {
    hipLaunchKernelGGL(chan_reduce_cols_kernel, dim3((C + 31) / 32), dim3(256), 0, s, a, b, mean, scale, out, N, C,
                       accumulate);
    return launch_status();
  }
  if (C >= 256 && (int64_t)N * HW <= 8192) {   // many short rows: one workgroup per channel, no atomics, no memset
    hipLaunchKernelGGL(chan_reduce_block_kernel, dim3(C), dim3(256), 0, s, a, b, (const float*)nullptr, mean, scale, out,
                       (float*)nullptr, N, C, HW, accumulate);
    return launch_status();
  }
  // runs of at least RED_SEG elements, doubled while the rows still yield >= 256 workgroups (one per CU): the 50 MB of the
  // gradient penalty's squared norms at 1536 / 768 / 384 workgroups: 2.4 / 3.2 / 4.2 TB/s (fewer tickets, fewer ramps)
  int64_t seg = RED_SEG;
  static const int min_wgs = getenv("SRGAN_REDUCE_MIN_WGS") ? atoi(getenv("SRGAN_REDUCE_MIN_WGS")) : 256;
  static const int64_t seg_cap = getenv("SRGAN_REDUCE_SEG_CAP") ? atol(getenv("SRGAN_REDUCE_SEG_CAP")) : 65536;
  while (seg < seg_cap && (int64_t)N * C * ((HW + 2 * seg - 1) / (2 * seg)) >= min_wgs) seg *= 2;
  const int segs = (int)((HW + seg - 1) / seg);
  SRGAN_REQUIRE((int64_t)N * segs <= 65535, SRGAN_ERANGE, "srgan_chan_reduce grid");
  static const bool unordered = getenv("SRGAN_REDUCE_UNORDERED") != nullptr;
  const int ticket_set = (C <= ROW_TICKETS && !unordered) ? workspace_index(s) : -1;
  if (ticket_set >= 0 && ticket_set < TICKET_SETS) {
    float* partial = partial_workspace((size_t)C * N * segs * sizeof(float), s);
    if (partial) {
      hipLaunchKernelGGL(chan_reduce_rows_ordered_kernel, dim3(C, N * segs), dim3(256), 0, s, a, b, mean, scale, out, partial,
                         ticket_set, C, HW, segs, seg, accumulate);
      return launch_status();
    }
  }
  if (!accumulate) if (const int status = zero_floats(out, C, s)) return status;
  hipLaunchKernelGGL(chan_reduce_rows_kernel, dim3(C, N * segs), dim3(256), 0, s, a, b, mean, scale, out, C, HW,
                     segs, seg);
  return launch_status();
}

int srgan_bn_conv_tangent_weights(const float* w, const float* q, const float* inv_std, const float* gamma,
                                  float* w_scaled, float* w_grad, float* gamma_grad, int32_t CO, int32_t CI, int32_t taps,
                                  void* stream) {
  SRGAN_REQUIRE(w && inv_std && gamma && CO > 0 && CI > 0 && taps > 0 && (w_scaled || q), SRGAN_EINVAL,
                "srgan_bn_conv_tangent_weights arguments");
  SRGAN_REQUIRE(q == nullptr || (w_grad && gamma_grad), SRGAN_EINVAL, "srgan_bn_conv_tangent_weights gradient outputs");
  const int64_t inner = (int64_t)CI * taps;
  SRGAN_REQUIRE(inner < ((int64_t)1 << 31), SRGAN_ERANGE, "srgan_bn_conv_tangent_weights size");
  SRGAN_REQUIRE(taps <= 64, SRGAN_EUNSUPPORTED, "srgan_bn_conv_tangent_weights: at most 64 taps");
  const int span = 64 - 64 % taps;              // whole channels per workgroup, every row of CO: see tangent_weight_body
  hipLaunchKernelGGL(tangent_weight_kernel, dim3((unsigned)((inner + span - 1) / span), 1), dim3(256), 0, (hipStream_t)stream, w, q,
                     inv_std, gamma, w_scaled, w_grad, gamma_grad, CO, (int)inner, taps);
  return launch_status();
}

int srgan_bn_conv_tangent_weights_job(const float* w, const float* inv_std, const float* gamma, float* w_grad,
                                      float* gamma_grad, int64_t offset, int32_t CO, int32_t CI, int32_t taps, void* job) {
  SRGAN_REQUIRE(w && inv_std && gamma && job && offset >= 0 && CO > 0 && CI > 0 && taps > 0 && taps <= 64, SRGAN_EINVAL,
                "srgan_bn_conv_tangent_weights_job arguments");
  const int64_t inner = (int64_t)CI * taps;
  SRGAN_REQUIRE(inner < ((int64_t)1 << 31), SRGAN_ERANGE, "srgan_bn_conv_tangent_weights_job size");
  TangentWeightJob slot;
  slot.w = w; slot.inv_std = inv_std; slot.gamma = gamma; slot.w_grad = w_grad; slot.gamma_grad = gamma_grad;
  slot.offset = offset; slot.CO = CO; slot.inner = (int32_t)inner; slot.taps = taps; slot.pad = 0;
  memcpy(job, &slot, sizeof(slot));
  return SRGAN_OK;
}

int srgan_bn_conv_tangent_weights_grouped(const void* jobs, int32_t count, int32_t max_inner, int32_t max_co, float* scaled_base,
                                          const float* q_base, void* stream) {
  SRGAN_REQUIRE(jobs && count >= 1 && count <= 65535 && max_inner >= 1 && max_co >= 1 && (scaled_base || q_base), SRGAN_EINVAL,
                "srgan_bn_conv_tangent_weights_grouped arguments");
  // (a job's span of inner indices per workgroup is 64 - 64 % taps >= 33: the grid covers the smallest)
  hipLaunchKernelGGL(tangent_weight_grouped_kernel, dim3((unsigned)((max_inner + 32) / 33), 1, (unsigned)count), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const TangentWeightJob*>(jobs), scaled_base, q_base);
  return launch_status();
}

int srgan_bn_act_bwd(const float* g, const float* x, const float* mean, const float* inv_std, const float* gamma,
                     const float* beta, int relu, float* gx, float* g_gamma, float* g_beta, int32_t N, int32_t C,
                     int64_t HW, int64_t g_batch_stride, int64_t x_batch_stride, int64_t gx_batch_stride,
                     int accumulate_gx, int unscaled, void* stream) {
  SRGAN_REQUIRE(g && x && mean && inv_std && gamma && beta && N > 0 && C > 0 && HW > 0 && N <= 65535, SRGAN_EINVAL,
                "srgan_bn_act_bwd arguments");
  SRGAN_REQUIRE((g_gamma == nullptr) == (g_beta == nullptr) && (gx || g_gamma), SRGAN_EINVAL,
                "srgan_bn_act_bwd outputs");
  const int64_t dense = (int64_t)C * HW;
  const int64_t g_bs = g_batch_stride ? g_batch_stride : dense, x_bs = x_batch_stride ? x_batch_stride : dense;
  const int64_t gx_bs = gx_batch_stride ? gx_batch_stride : dense;
  // Images per workgroup: an HBM-bound pass wants ~10 MB in flight (2048 resident workgroups x 12 KB), and a
  // workgroup at least a few iterations of work (4096 elements) to amortise its reduction and two atomics.
  int per = 1;
  while (per < N && ((int64_t)C * ((N + per - 1) / per) > 2048 || (int64_t)per * HW < 4096) &&
         (int64_t)C * ((N + 2 * per - 1) / (2 * per)) >= 1024)
    per *= 2;
  const int blocks_n = (N + per - 1) / per;
  unsigned int* tickets = nullptr;
  float* partial = (g_gamma && blocks_n > 1) ? row_finish_workspace(C, blocks_n, 2, g_reduce_finish_tickets, (hipStream_t)stream, &tickets)
                                             : nullptr;
  if (relu) hipLaunchKernelGGL(bn_act_bwd_rows_kernel<true>, dim3(C, blocks_n), dim3(256), 0, (hipStream_t)stream, g, x,
                               mean, inv_std, gamma, beta, gx, g_gamma, g_beta, N, C, HW, g_bs, x_bs, gx_bs, accumulate_gx,
                               unscaled, per, partial, tickets);
  else hipLaunchKernelGGL(bn_act_bwd_rows_kernel<false>, dim3(C, blocks_n), dim3(256), 0, (hipStream_t)stream, g, x, mean,
                          inv_std, gamma, beta, gx, g_gamma, g_beta, N, C, HW, g_bs, x_bs, gx_bs, accumulate_gx, unscaled,
                          per, partial, tickets);
  return launch_status();
}

int srgan_row_max(const float* x, float* out, int32_t B, int32_t F, void* stream) {
  SRGAN_REQUIRE(x && out && B > 0 && F > 0, SRGAN_EINVAL, "srgan_row_max arguments");
  hipLaunchKernelGGL(row_max_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, B, F);
  return launch_status();
}

int srgan_nearest_bin_onehot(const float* y, const float* bins, float* onehot, int32_t B, int32_t K, void* stream) {
  SRGAN_REQUIRE(y && bins && onehot && B > 0 && K > 0, SRGAN_EINVAL, "srgan_nearest_bin_onehot arguments");
  hipLaunchKernelGGL(nearest_bin_onehot_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, bins,
                     onehot, B, K);
  return launch_status();
}

int srgan_crowd_map_l1_fwd(const float* maps, const float* target, float* rows, int32_t B, int32_t Cm, int64_t HW,
                           void* stream) {
  SRGAN_REQUIRE(maps && target && rows && B > 0 && Cm > 0 && HW > 0, SRGAN_EINVAL, "srgan_crowd_map_l1_fwd arguments");
  hipStream_t s = (hipStream_t)stream;
  const int segs = (int)((HW + RED_SEG - 1) / RED_SEG);
  if (const int status = zero_floats(rows, B, s)) return status;
  unsigned int* tickets = nullptr;
  float* partial = segs > 1 ? row_finish_workspace(B, segs, 1, g_reduce_finish_tickets, s, &tickets) : nullptr;
  hipLaunchKernelGGL(crowd_map_l1_kernel, dim3(B, segs), dim3(256), 0, s, maps, target, rows, Cm, HW, segs, partial, tickets);
  return launch_status();
}

int srgan_crowd_map_l1_bwd(const float* maps, const float* target, const float* g_rows, float* g_maps, int32_t B,
                           int32_t Cm, int64_t HW, void* stream) {
  SRGAN_REQUIRE(maps && target && g_rows && g_maps && B > 0 && Cm > 0 && HW > 0, SRGAN_EINVAL,
                "srgan_crowd_map_l1_bwd arguments");
  const int64_t n = (int64_t)B * Cm * HW;
  hipLaunchKernelGGL(crowd_map_l1_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, maps,
                     target, g_rows, g_maps, Cm, HW, n);
  return launch_status();
}

}  // extern "C"
