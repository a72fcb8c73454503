// pool.hip -- max / average pooling for NCHW fp32 (DenseNet stem max-pool k3 s2 p1, VGG k2 s2, transition
// average pools, final global pool).  HBM-bound gather / scatter kernels, one thread per output element.
// Max-pool keeps the flat in-plane arg-max so that backward (scatter) and double-backward (gather) reuse
// the forward's choice; ties resolve to the first maximum in row-major window order, as torch's CPU kernel.
#include "common.h"
#include "split_finish.h"
#include <math.h>

namespace srgan {

// The geometric kernels are instantiated for 32-bit element indices (every tensor of the training step: 64-bit
// division is emulated in ~100 instructions and made these kernels ALU-bound) and for 64-bit ones (> 2^31 elements).
template <typename I>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int32_t* __restrict__ idx, int H, int W, int k, int s, int p,
                                                          int OH, int OW, int64_t n) {
  const I stride = (I)gridDim.x * 256, count = (I)n, plane_out = (I)OW * OH;
  for (I o = (I)blockIdx.x * 256 + threadIdx.x; o < count; o += stride) {
    const I plane = o / plane_out, rest = o - plane * plane_out;
    const int oh = (int)(rest / OW), ow = (int)(rest - (I)oh * OW);
    const float* src = x + (int64_t)plane * H * W;
    const int h0 = oh * s - p, w0 = ow * s - p;
    float best = -INFINITY;
    int best_i = -1;
    for (int r = 0; r < k; ++r) {
      const int h = h0 + r;
      if (h < 0 || h >= H) continue;
      for (int c = 0; c < k; ++c) {
        const int w = w0 + c;
        if (w < 0 || w >= W) continue;
        const float v = src[h * W + w];
        if (v > best || best_i < 0 || v != v) { best = v; best_i = h * W + w; }
      }
    }
    y[o] = best;
    idx[o] = best_i;
  }
}

// out (zeroed by the caller) [plane, idx[o]] += g[o]
__global__ __launch_bounds__(256) void pool_scatter_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx,
                                                           float* __restrict__ out, int64_t in_plane,
                                                           int64_t out_plane, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) {
    const int64_t plane = o / out_plane;
    unsafeAtomicAdd(out + plane * in_plane + idx[o], g[o]);
  }
}

// Max-pool backward in gather form: gx[h, w] = sum of g over the windows that contain (h, w) AND chose it.  Every input
// element is written exactly once (no zero-fill, no atomics); the <= ceil(k/s)^2 window look-ups per element hit L2.
template <typename I>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx,
                                                          float* __restrict__ gx, int H, int W, int k, int s, int p,
                                                          int OH, int OW, int64_t n) {
  const I stride = (I)gridDim.x * 256, count = (I)n, plane_in = (I)W * H;
  for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const I plane = i / plane_in, rest = i - plane * plane_in;
    const int h = (int)(rest / W), w = (int)(rest - (I)h * W);
    const float* src = g + (int64_t)plane * OH * OW;
    const int32_t* chosen = idx + (int64_t)plane * OH * OW;
    int oh_lo = h + p - k + 1; oh_lo = oh_lo > 0 ? (oh_lo + s - 1) / s : 0;
    int ow_lo = w + p - k + 1; ow_lo = ow_lo > 0 ? (ow_lo + s - 1) / s : 0;
    int oh_hi = (h + p) / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_hi = (w + p) / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
    const int32_t me = h * W + w;
    float acc = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow)
        if (chosen[oh * OW + ow] == me) acc += src[oh * OW + ow];
    gx[i] = acc;
  }
}

// The same for compile-time window geometry and four consecutive pixels per thread (W % 4 == 0, 16-byte aligned rows):
// the windows of neighbouring pixels overlap, so the four pixels share their (arg-max, g) look-ups, the window bounds
// are shifts instead of divisions, and the result leaves as one float4.
template <int K, int S, int P>
__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx,
                                                               float* __restrict__ gx, int H, int W, int OH, int OW,
                                                               uint32_t quads) {
  constexpr int SPAN = (3 + P) / S - (P - K + 1 > 0 ? (P - K + 1 + S - 1) / S : 0) + 2;   // upper bound of windows per quad row
  const uint32_t stride = gridDim.x * 256u, w4 = (uint32_t)W >> 2;
  for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < quads; q += stride) {
    const uint32_t row = q / w4;                       // plane * H + h
    const int w0 = (int)(q - row * w4) * 4;
    const uint32_t plane = row / (uint32_t)H;
    const int h = (int)(row - plane * (uint32_t)H);
    const float* src = g + (int64_t)plane * OH * OW;
    const int32_t* chosen = idx + (int64_t)plane * OH * OW;
    int oh_lo = h + P - K + 1; oh_lo = oh_lo > 0 ? (oh_lo + S - 1) / S : 0;
    int oh_hi = (h + P) / S; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = w0 + P - K + 1; ow_lo = ow_lo > 0 ? (ow_lo + S - 1) / S : 0;
    int ow_hi = (w0 + 3 + P) / S; if (ow_hi > OW - 1) ow_hi = OW - 1;
    const int32_t me = h * W + w0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
#pragma unroll
      for (int j = 0; j < SPAN; ++j) {
        const int ow = ow_lo + j;
        if (ow > ow_hi) break;
        const int32_t d = chosen[oh * OW + ow] - me;   // 0 .. 3 when the window chose one of this thread's pixels
        if ((uint32_t)d < 4u) {
          const float v = src[oh * OW + ow];
          acc[0] += d == 0 ? v : 0.f; acc[1] += d == 1 ? v : 0.f; acc[2] += d == 2 ? v : 0.f; acc[3] += d == 3 ? v : 0.f;
        }
      }
    *reinterpret_cast<float4*>(gx + (int64_t)row * W + w0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// DenseNet stem, norm0 -> relu0 -> pool0 (reference crowd/models.py:1072-1076) in one pass: y = maxpool(relu(fma(x, a, b)))
// with the frozen batch-norm folded into (a, b) per channel; the activated tensor (4 x the pooled one) never exists.
// Same scan order and tie rule as maxpool_fwd_kernel, so the arg-max equals the two-kernel form's.
template <typename I>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ inv_std,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ y,
                                                                  int32_t* __restrict__ idx, int C, int H, int W, int k, int s,
                                                                  int p, int OH, int OW, int64_t n) {
  const I stride = (I)gridDim.x * 256, count = (I)n, plane_out = (I)OW * OH;
  for (I o = (I)blockIdx.x * 256 + threadIdx.x; o < count; o += stride) {
    const I plane = o / plane_out, rest = o - plane * plane_out;
    const int oh = (int)(rest / OW), ow = (int)(rest - (I)oh * OW);
    const int c = (int)(plane % (I)C);
    float a, b;
    bn_coefficients(mean[c], inv_std[c], gamma[c], beta[c], a, b);
    const float* src = x + (int64_t)plane * H * W;
    const int h0 = oh * s - p, w0 = ow * s - p;
    float best = -INFINITY;
    int best_i = -1;
    for (int r = 0; r < k; ++r) {
      const int h = h0 + r;
      if (h < 0 || h >= H) continue;
      for (int q = 0; q < k; ++q) {
        const int w = w0 + q;
        if (w < 0 || w >= W) continue;
        const float v = fmaxf(fmaf(src[h * W + w], a, b), 0.f);
        if (v > best || best_i < 0 || v != v) { best = v; best_i = h * W + w; }
      }
    }
    y[o] = best;
    idx[o] = best_i;
  }
}

// Its whole backward in one pass over (gy, argmax, x): the pooled gradient gathered per input pixel (as in
// maxpool_bwd_quad_kernel), masked by the recomputed activation and scaled -- gx = S * [fma(x, a, b) > 0] * a -- and both
// parameter sums (beta: sum of the masked S; gamma: inv_std * sum of masked S * (x - mean)) reduced per workgroup and
// added atomically.  One workgroup = QUADS_PER_BLOCK float4 of ONE plane (blockIdx.y), so a block's sums are one channel's.
constexpr int POOL_BWD_QUADS = 1024;
__device__ unsigned int g_pool_finish_tickets[SPLIT_TICKET_SETS * ROW_FINISH_ROWS];
template <int K, int S, int P>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx,
                                                                  const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ inv_std,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ gx,
                                                                  float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                                  int C, int H, int W, int OH, int OW, float* finish_partial,
                                                                  unsigned int* finish_tickets) {
  constexpr int SPAN = (3 + P) / S - (P - K + 1 > 0 ? (P - K + 1 + S - 1) / S : 0) + 2;
  __shared__ float scratch[2][4];
  const uint32_t plane = blockIdx.y, w4 = (uint32_t)W >> 2, plane_quads = (uint32_t)H * w4;
  const int c = (int)(plane % (uint32_t)C);
  float a, b;
  const float mu = mean[c];
  bn_coefficients(mu, inv_std[c], gamma[c], beta[c], a, b);
  const float* src = g + (int64_t)plane * OH * OW;
  const int32_t* chosen = idx + (int64_t)plane * OH * OW;
  float sum_plain = 0.f, sum_centred = 0.f;
  const uint32_t first = blockIdx.x * POOL_BWD_QUADS;
  for (uint32_t q = first + threadIdx.x; q < min(first + POOL_BWD_QUADS, plane_quads); q += 256u) {
    const int h = (int)(q / w4);
    const int w0 = (int)(q - (uint32_t)h * w4) * 4;
    int oh_lo = h + P - K + 1; oh_lo = oh_lo > 0 ? (oh_lo + S - 1) / S : 0;
    int oh_hi = (h + P) / S; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = w0 + P - K + 1; ow_lo = ow_lo > 0 ? (ow_lo + S - 1) / S : 0;
    int ow_hi = (w0 + 3 + P) / S; if (ow_hi > OW - 1) ow_hi = OW - 1;
    const int32_t me = h * W + w0;
    const int64_t at = ((int64_t)plane * H + h) * W + w0;
    const float4 xv = *reinterpret_cast<const float4*>(x + at);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
#pragma unroll
      for (int j = 0; j < SPAN; ++j) {
        const int ow = ow_lo + j;
        if (ow > ow_hi) break;
        const int32_t d = chosen[oh * OW + ow] - me;   // 0 .. 3 when the window chose one of this thread's pixels
        if ((uint32_t)d < 4u) {
          const float v = src[oh * OW + ow];
          acc[0] += d == 0 ? v : 0.f; acc[1] += d == 1 ? v : 0.f; acc[2] += d == 2 ? v : 0.f; acc[3] += d == 3 ? v : 0.f;
        }
      }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = fmaf(xs[j], a, b) > 0.f ? acc[j] : 0.f;
      sum_plain += acc[j];
      sum_centred = fmaf(acc[j], xs[j] - mu, sum_centred);
    }
    *reinterpret_cast<float4*>(gx + at) = make_float4(acc[0] * a, acc[1] * a, acc[2] * a, acc[3] * a);
  }
  if (g_gamma == nullptr) return;
  const float plain = wave_sum(sum_plain), centred = wave_sum(sum_centred);
  const int wave = (int)threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { scratch[0][wave] = plain; scratch[1][wave] = centred; }
  __syncthreads();
  float v[2] = {(scratch[0][0] + scratch[0][1]) + (scratch[0][2] + scratch[0][3]),
                (scratch[1][0] + scratch[1][1]) + (scratch[1][2] + scratch[1][3])};       // (every thread; thread 0's are used)
  if (finish_partial) {        // ordered: the channel's workgroups (image x plane block) meet in a fixed order, ONE adder per channel
    const int parts = (int)(gridDim.y / (unsigned)C) * (int)gridDim.x;
    const int part = (int)(plane / (uint32_t)C) * (int)gridDim.x + (int)blockIdx.x;
    __syncthreads();
    if (ordered_row_finish<2>(v, finish_partial + (int64_t)c * parts * 2, part, parts, finish_tickets + c, &scratch[0][0])) {
      g_beta[c] += v[0];
      g_gamma[c] += v[1] * inv_std[c];
    }
    return;
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(g_beta + c, v[0]);
    unsafeAtomicAdd(g_gamma + c, v[1] * inv_std[c]);
  }
}

// ---- avg_pool2d(relu(batch_norm_eval(x)), 2, 2) as one pass each way: the DenseNet transitions evaluated as
// norm -> relu -> pool -> conv (reference crowd/models.py:364-371 has conv -> pool; a 1x1 convolution and the average
// pooling commute).  A thread owns two vertically adjacent float4 of x (rows 2 oh, 2 oh + 1; W % 4 == 0) = two pooled pixels.
// Forward: the activated tensor -- four times the pooled one -- is never written.  Backward: gx = 0.25 * g[oh, ow] * mask * a
// and the two parameter sums (beta: sum of the masked 0.25 g; gamma: inv_std * sum of masked 0.25 g * (x - mean)), one
// workgroup = POOL_BWD_QUADS thread items of ONE plane (blockIdx.y), reduced per workgroup and added atomically.
__global__ __launch_bounds__(256) void bn_relu_avgpool2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ inv_std,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ y,
                                                                   int C, int H, int W) {
  const uint32_t plane = blockIdx.y, w4 = (uint32_t)W >> 2, items = ((uint32_t)H >> 1) * w4;
  const int c = (int)(plane % (uint32_t)C);
  float a, b;
  bn_coefficients(mean[c], inv_std[c], gamma[c], beta[c], a, b);
  const float* src = x + (int64_t)plane * H * W;
  float* dst = y + (int64_t)plane * (H >> 1) * (W >> 1);
  const uint32_t first = blockIdx.x * POOL_BWD_QUADS;
  for (uint32_t q = first + threadIdx.x; q < min(first + POOL_BWD_QUADS, items); q += 256u) {
    const uint32_t oh = q / w4, wq = q - oh * w4;
    const float4 top = *reinterpret_cast<const float4*>(src + (int64_t)(2 * oh) * W + 4 * wq);
    const float4 low = *reinterpret_cast<const float4*>(src + (int64_t)(2 * oh + 1) * W + 4 * wq);
    auto act = [&](float v) { return fmaxf(fmaf(v, a, b), 0.f); };
    // (the order avgpool_fwd_kernel adds a window in: row by row)
    const float left = (((act(top.x) + act(top.y)) + act(low.x)) + act(low.y)) * 0.25f;
    const float right = (((act(top.z) + act(top.w)) + act(low.z)) + act(low.w)) * 0.25f;
    *reinterpret_cast<float2*>(dst + (int64_t)oh * (W >> 1) + 2 * wq) = make_float2(left, right);
  }
}

__global__ __launch_bounds__(256) void bn_relu_avgpool2_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ inv_std,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ gx,
                                                                   float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                                   int C, int H, int W, float* finish_partial,
                                                                   unsigned int* finish_tickets) {
  __shared__ float scratch[2][4];
  const uint32_t plane = blockIdx.y, w4 = (uint32_t)W >> 2, items = ((uint32_t)H >> 1) * w4;
  const int c = (int)(plane % (uint32_t)C);
  float a, b;
  const float mu = mean[c];
  bn_coefficients(mu, inv_std[c], gamma[c], beta[c], a, b);
  const float* src = g + (int64_t)plane * (H >> 1) * (W >> 1);
  float sum_plain = 0.f, sum_centred = 0.f;
  const uint32_t first = blockIdx.x * POOL_BWD_QUADS;
  for (uint32_t q = first + threadIdx.x; q < min(first + POOL_BWD_QUADS, items); q += 256u) {
    const uint32_t oh = q / w4, wq = q - oh * w4;
    const float2 pooled = *reinterpret_cast<const float2*>(src + (int64_t)oh * (W >> 1) + 2 * wq);
    const float gl = pooled.x * 0.25f, gr = pooled.y * 0.25f;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const int64_t at = ((int64_t)plane * H + 2 * oh + row) * W + 4 * wq;
      const float4 xv = *reinterpret_cast<const float4*>(x + at);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      float v[4] = {gl, gl, gr, gr};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = fmaf(xs[j], a, b) > 0.f ? v[j] : 0.f;
        sum_plain += v[j];
        sum_centred = fmaf(v[j], xs[j] - mu, sum_centred);
      }
      *reinterpret_cast<float4*>(gx + at) = make_float4(v[0] * a, v[1] * a, v[2] * a, v[3] * a);
    }
  }
  if (g_gamma == nullptr) return;
  const float plain = wave_sum(sum_plain), centred = wave_sum(sum_centred);
  const int wave = (int)threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { scratch[0][wave] = plain; scratch[1][wave] = centred; }
  __syncthreads();
  float v[2] = {(scratch[0][0] + scratch[0][1]) + (scratch[0][2] + scratch[0][3]),
                (scratch[1][0] + scratch[1][1]) + (scratch[1][2] + scratch[1][3])};       // (every thread; thread 0's are used)
  if (finish_partial) {        // ordered: the channel's workgroups (image x plane block) meet in a fixed order, ONE adder per channel
    const int parts = (int)(gridDim.y / (unsigned)C) * (int)gridDim.x;
    const int part = (int)(plane / (uint32_t)C) * (int)gridDim.x + (int)blockIdx.x;
    __syncthreads();
    if (ordered_row_finish<2>(v, finish_partial + (int64_t)c * parts * 2, part, parts, finish_tickets + c, &scratch[0][0])) {
      g_beta[c] += v[0];
      g_gamma[c] += v[1] * inv_std[c];
    }
    return;
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(g_beta + c, v[0]);
    unsafeAtomicAdd(g_gamma + c, v[1] * inv_std[c]);
  }
}

__global__ __launch_bounds__(256) void pool_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t in_plane, int64_t out_plane,
                                                          int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) {
    const int64_t plane = o / out_plane;
    out[o] = src[plane * in_plane + idx[o]];
  }
}

template <typename I>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                          int W, int k, int s, int OH, int OW, int64_t n) {
  const I stride = (I)gridDim.x * 256, count = (I)n, plane_out = (I)OW * OH;
  const float inv = 1.f / (float)(k * k);
  for (I o = (I)blockIdx.x * 256 + threadIdx.x; o < count; o += stride) {
    const I plane = o / plane_out, rest = o - plane * plane_out;
    const int oh = (int)(rest / OW), ow = (int)(rest - (I)oh * OW);
    const float* src = x + (int64_t)plane * H * W + (int64_t)(oh * s) * W + ow * s;
    float acc = 0.f;
    for (int r = 0; r < k; ++r)
      for (int c = 0; c < k; ++c) acc += src[r * W + c];
    y[o] = acc * inv;
  }
}

// gx[h, w] = (1 / k^2) * sum of g over the output windows that contain (h, w)
template <typename I>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx, int H,
                                                          int W, int k, int s, int OH, int OW, int64_t n) {
  const I stride = (I)gridDim.x * 256, count = (I)n, plane_in = (I)W * H;
  const float inv = 1.f / (float)(k * k);
  for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const I plane = i / plane_in, rest = i - plane * plane_in;
    const int h = (int)(rest / W), w = (int)(rest - (I)h * W);
    const float* src = g + (int64_t)plane * OH * OW;
    int oh_lo = h - k + 1; oh_lo = oh_lo > 0 ? (oh_lo + s - 1) / s : 0;
    int ow_lo = w - k + 1; ow_lo = ow_lo > 0 ? (ow_lo + s - 1) / s : 0;
    int oh_hi = h / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_hi = w / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
    float acc = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) acc += src[oh * OW + ow];
    gx[i] = acc * inv;
  }
}

// The same for rows of whole float4s (W % 4 == 0, 16-byte aligned gx): a thread owns four consecutive pixels of a row and
// stores them as one float4 (round 3: the scalar form wrote the 16 x 16 planes behind the global average pool at 1 TB/s).
template <typename I>
__global__ __launch_bounds__(256) void avgpool_bwd_vec4_kernel(const float* __restrict__ g, float* __restrict__ gx, int H,
                                                               int W, int k, int s, int OH, int OW, int64_t n4) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const I stride = (I)gridDim.x * 256, count = (I)n4, quads = (I)(W >> 2), plane_quads = quads * (I)H;
  const float inv = 1.f / (float)(k * k);
  for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const I plane = i / plane_quads, rest = i - plane * plane_quads;
    const int h = (int)(rest / quads), w0 = (int)(rest - (I)h * quads) * 4;
    const float* src = g + (int64_t)plane * OH * OW;
    int oh_lo = h - k + 1; oh_lo = oh_lo > 0 ? (oh_lo + s - 1) / s : 0;
    int oh_hi = h / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
    v4f out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int w = w0 + e;
      int ow_lo = w - k + 1; ow_lo = ow_lo > 0 ? (ow_lo + s - 1) / s : 0;
      int ow_hi = w / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
      float acc = 0.f;
      for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) acc += src[oh * OW + ow];
      out[e] = acc * inv;
    }
    __builtin_nontemporal_store(out, reinterpret_cast<v4f*>(gx) + i);
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" {

int srgan_maxpool2d_fwd(const float* x, float* y, int32_t* argmax, int32_t planes, int32_t H, int32_t W, int32_t k,
                        int32_t s, int32_t p, int32_t OH, int32_t OW, void* stream) {
  SRGAN_REQUIRE(x && y && argmax && planes > 0 && H > 0 && W > 0 && k > 0 && s > 0 && p >= 0 && OH > 0 && OW > 0,
                SRGAN_EINVAL, "srgan_maxpool2d_fwd arguments");
  SRGAN_REQUIRE((OH - 1) * s - p < H && (OW - 1) * s - p < W && p < k, SRGAN_EINVAL, "srgan_maxpool2d_fwd geometry");
  const int64_t n = (int64_t)planes * OH * OW;
  if (n < ((int64_t)1 << 31) - ((int64_t)2048 * 256))      // (the grid-stride loop adds at most grid * 256 to a valid index)
    hipLaunchKernelGGL(maxpool_fwd_kernel<uint32_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       argmax, H, W, k, s, p, OH, OW, n);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<int64_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       argmax, H, W, k, s, p, OH, OW, n);
  return launch_status();
}

int srgan_bn_relu_maxpool_fwd(const float* x, const float* mean, const float* inv_std, const float* gamma, const float* beta,
                              float* y, int32_t* argmax, int32_t N, int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p,
                              int32_t OH, int32_t OW, void* stream) {
  SRGAN_REQUIRE(x && mean && inv_std && gamma && beta && y && argmax && N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && s > 0 &&
                p >= 0 && OH > 0 && OW > 0, SRGAN_EINVAL, "srgan_bn_relu_maxpool_fwd arguments");
  SRGAN_REQUIRE((OH - 1) * s - p < H && (OW - 1) * s - p < W && p < k, SRGAN_EINVAL, "srgan_bn_relu_maxpool_fwd geometry");
  const int64_t n = (int64_t)N * C * OH * OW;
  if (n < ((int64_t)1 << 31) - ((int64_t)2048 * 256) && (int64_t)N * C * H * W < ((int64_t)1 << 31))
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<uint32_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       mean, inv_std, gamma, beta, y, argmax, C, H, W, k, s, p, OH, OW, n);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<int64_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       mean, inv_std, gamma, beta, y, argmax, C, H, W, k, s, p, OH, OW, n);
  return launch_status();
}

// 1 when srgan_bn_relu_maxpool_bwd has this geometry (3 / 2 / 1 windows on rows of whole float4s)
int srgan_bn_relu_maxpool_bwd_supported(int32_t N, int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p) {
  return (k == 3 && s == 2 && p == 1 && W % 4 == 0 && (int64_t)N * C <= 65535 &&
          (int64_t)N * C * H * W < ((int64_t)1 << 31)) ? 1 : 0;
}

int srgan_bn_relu_maxpool_bwd(const float* gy, const int32_t* argmax, const float* x, const float* mean, const float* inv_std,
                              const float* gamma, const float* beta, float* gx, float* g_gamma, float* g_beta, int32_t N,
                              int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p, int32_t OH, int32_t OW,
                              void* stream) {
  SRGAN_REQUIRE(gy && argmax && x && mean && inv_std && gamma && beta && gx && OH > 0 && OW > 0, SRGAN_EINVAL,
                "srgan_bn_relu_maxpool_bwd arguments");
  SRGAN_REQUIRE((g_gamma == nullptr) == (g_beta == nullptr), SRGAN_EINVAL, "srgan_bn_relu_maxpool_bwd parameter outputs");
  SRGAN_REQUIRE(srgan_bn_relu_maxpool_bwd_supported(N, C, H, W, k, s, p) && ((((uintptr_t)x | (uintptr_t)gx) & 15) == 0),
                SRGAN_EUNSUPPORTED, "srgan_bn_relu_maxpool_bwd geometry");
  const uint32_t plane_quads = (uint32_t)H * ((uint32_t)W >> 2);
  dim3 grid((plane_quads + POOL_BWD_QUADS - 1) / POOL_BWD_QUADS, (unsigned)(N * C), 1);
  unsigned int* tickets = nullptr;
  float* partial = g_gamma ? row_finish_workspace(C, (int)(grid.x * N), 2, g_pool_finish_tickets, (hipStream_t)stream, &tickets) : nullptr;
  hipLaunchKernelGGL((bn_relu_maxpool_bwd_kernel<3, 2, 1>), grid, dim3(256), 0, (hipStream_t)stream, gy, argmax, x, mean, inv_std,
                     gamma, beta, gx, g_gamma, g_beta, C, H, W, OH, OW, partial, tickets);
  return launch_status();
}

// 1 when srgan_bn_relu_avgpool2_fwd / _bwd have this geometry (2 x 2 windows on even planes of whole float4 rows)
int srgan_bn_relu_avgpool2_supported(int32_t N, int32_t C, int32_t H, int32_t W) {
  return (N > 0 && C > 0 && H >= 2 && W >= 4 && H % 2 == 0 && W % 4 == 0 && (int64_t)N * C <= 65535 &&
          (int64_t)N * C * H * W < ((int64_t)1 << 31)) ? 1 : 0;
}

int srgan_bn_relu_avgpool2_fwd(const float* x, const float* mean, const float* inv_std, const float* gamma, const float* beta,
                               float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  SRGAN_REQUIRE(x && mean && inv_std && gamma && beta && y, SRGAN_EINVAL, "srgan_bn_relu_avgpool2_fwd arguments");
  SRGAN_REQUIRE(srgan_bn_relu_avgpool2_supported(N, C, H, W) && ((((uintptr_t)x) & 15) | (((uintptr_t)y) & 7)) == 0,
                SRGAN_EUNSUPPORTED, "srgan_bn_relu_avgpool2_fwd geometry");
  const uint32_t items = ((uint32_t)H >> 1) * ((uint32_t)W >> 2);
  dim3 grid((items + POOL_BWD_QUADS - 1) / POOL_BWD_QUADS, (unsigned)(N * C), 1);
  hipLaunchKernelGGL(bn_relu_avgpool2_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, mean, inv_std, gamma, beta, y, C, H, W);
  return launch_status();
}

int srgan_bn_relu_avgpool2_bwd(const float* gy, const float* x, const float* mean, const float* inv_std, const float* gamma,
                               const float* beta, float* gx, float* g_gamma, float* g_beta, int32_t N, int32_t C, int32_t H,
                               int32_t W, void* stream) {
  SRGAN_REQUIRE(gy && x && mean && inv_std && gamma && beta && gx, SRGAN_EINVAL, "srgan_bn_relu_avgpool2_bwd arguments");
  SRGAN_REQUIRE((g_gamma == nullptr) == (g_beta == nullptr), SRGAN_EINVAL, "srgan_bn_relu_avgpool2_bwd parameter outputs");
  SRGAN_REQUIRE(srgan_bn_relu_avgpool2_supported(N, C, H, W) &&
                ((((uintptr_t)x | (uintptr_t)gx) & 15) | (((uintptr_t)gy) & 7)) == 0, SRGAN_EUNSUPPORTED,
                "srgan_bn_relu_avgpool2_bwd geometry");
  const uint32_t items = ((uint32_t)H >> 1) * ((uint32_t)W >> 2);
  dim3 grid((items + POOL_BWD_QUADS - 1) / POOL_BWD_QUADS, (unsigned)(N * C), 1);
  unsigned int* tickets = nullptr;
  float* partial = g_gamma ? row_finish_workspace(C, (int)(grid.x * N), 2, g_pool_finish_tickets, (hipStream_t)stream, &tickets) : nullptr;
  hipLaunchKernelGGL(bn_relu_avgpool2_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, x, mean, inv_std, gamma, beta, gx,
                     g_gamma, g_beta, C, H, W, partial, tickets);
  return launch_status();
}

int srgan_maxpool2d_bwd(const float* g, const int32_t* argmax, float* gx, int32_t planes, int32_t H, int32_t W, int32_t k,
                        int32_t s, int32_t p, int32_t OH, int32_t OW, void* stream) {
  SRGAN_REQUIRE(g && argmax && gx && planes > 0 && H > 0 && W > 0 && k > 0 && s > 0 && p >= 0 && OH > 0 && OW > 0,
                SRGAN_EINVAL, "srgan_maxpool2d_bwd arguments");
  const int64_t n = (int64_t)planes * H * W;
  if (W % 4 == 0 && ((uintptr_t)gx & 15) == 0 && n < ((int64_t)1 << 31)) {
    const uint32_t quads = (uint32_t)(n / 4);
    const dim3 grid(stream_grid(quads, 256));
    if (k == 3 && s == 2 && p == 1) {
      hipLaunchKernelGGL((maxpool_bwd_quad_kernel<3, 2, 1>), grid, dim3(256), 0, (hipStream_t)stream, g, argmax, gx, H, W, OH,
                         OW, quads);
      return launch_status();
    }
    if (k == 2 && s == 2 && p == 0) {
      hipLaunchKernelGGL((maxpool_bwd_quad_kernel<2, 2, 0>), grid, dim3(256), 0, (hipStream_t)stream, g, argmax, gx, H, W, OH,
                         OW, quads);
      return launch_status();
    }
  }
  if (n < ((int64_t)1 << 31) - ((int64_t)2048 * 256))
    hipLaunchKernelGGL(maxpool_bwd_kernel<uint32_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       argmax, gx, H, W, k, s, p, OH, OW, n);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<int64_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g, argmax,
                       gx, H, W, k, s, p, OH, OW, n);
  return launch_status();
}

int srgan_pool_scatter(const float* g, const int32_t* argmax, float* out, int32_t planes, int64_t in_plane,
                       int64_t out_plane, void* stream) {
  SRGAN_REQUIRE(g && argmax && out && planes > 0 && in_plane > 0 && out_plane > 0, SRGAN_EINVAL,
                "srgan_pool_scatter arguments");
  hipStream_t s = (hipStream_t)stream;
  if (const int status = zero_floats(out, (int64_t)planes * in_plane, s)) return status;
  const int64_t n = (int64_t)planes * out_plane;
  hipLaunchKernelGGL(pool_scatter_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, argmax, out, in_plane,
                     out_plane, n);
  return launch_status();
}

int srgan_pool_gather(const float* src, const int32_t* argmax, float* out, int32_t planes, int64_t in_plane,
                      int64_t out_plane, void* stream) {
  SRGAN_REQUIRE(src && argmax && out && planes > 0 && in_plane > 0 && out_plane > 0, SRGAN_EINVAL,
                "srgan_pool_gather arguments");
  const int64_t n = (int64_t)planes * out_plane;
  hipLaunchKernelGGL(pool_gather_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src, argmax, out,
                     in_plane, out_plane, n);
  return launch_status();
}

int srgan_avgpool2d_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t s,
                        int32_t OH, int32_t OW, void* stream) {
  SRGAN_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && k > 0 && s > 0 && OH > 0 && OW > 0, SRGAN_EINVAL,
                "srgan_avgpool2d_fwd arguments");
  SRGAN_REQUIRE((OH - 1) * s + k <= H && (OW - 1) * s + k <= W, SRGAN_EINVAL, "srgan_avgpool2d_fwd geometry");
  const int64_t n = (int64_t)planes * OH * OW;
  if (n < ((int64_t)1 << 31) - ((int64_t)2048 * 256))
    hipLaunchKernelGGL(avgpool_fwd_kernel<uint32_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H,
                       W, k, s, OH, OW, n);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel<int64_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W,
                       k, s, OH, OW, n);
  return launch_status();
}

int srgan_avgpool2d_bwd(const float* g, float* gx, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t s,
                        int32_t OH, int32_t OW, void* stream) {
  SRGAN_REQUIRE(g && gx && planes > 0 && H > 0 && W > 0 && k > 0 && s > 0 && OH > 0 && OW > 0, SRGAN_EINVAL,
                "srgan_avgpool2d_bwd arguments");
  SRGAN_REQUIRE((OH - 1) * s + k <= H && (OW - 1) * s + k <= W, SRGAN_EINVAL, "srgan_avgpool2d_bwd geometry");
  const int64_t n = (int64_t)planes * H * W;
  if (W % 4 == 0 && ((uintptr_t)gx & 15) == 0 && n < ((int64_t)1 << 31)) {
    hipLaunchKernelGGL(avgpool_bwd_vec4_kernel<uint32_t>, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, g, gx,
                       H, W, k, s, OH, OW, n / 4);
    return launch_status();
  }
  if (n < ((int64_t)1 << 31) - ((int64_t)2048 * 256))
    hipLaunchKernelGGL(avgpool_bwd_kernel<uint32_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g, gx, H,
                       W, k, s, OH, OW, n);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<int64_t>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g, gx, H, W,
                       k, s, OH, OW, n);
  return launch_status();
}

int srgan_version(void) { return 110; }

}  // extern "C"
