// blocked16.hip -- the 16-bit data path: layout conversions, streaming kernels, the 3x3 convolution family.
//
// Reference graphs served: VGG-16 (age/vgg.py:28-53,70-84: conv3x3 + bias -> ReLU, max-pool 2x2, two Linear + ReLU)
// with activations, gradients and a shadow copy of the weights stored as bf16 / fp16 in the blocked layout of blocked16.h;
// master weights, weight gradients, Adam and every loss stay fp32.  The activation is FUSED: a convolution's epilogue adds
// the bias and applies relu / leaky_relu before the 16-bit store, and every kernel that produces a gradient with respect
// to an activated tensor multiplies by the activation's derivative on the way out (`mask by reference`: the sign pattern
// of the activated tensor itself), so neither the pre-activation tensor nor the un-masked gradient ever exists in HBM.
//
//   hconv3x3_kernel      out = epi(conv3x3_s1_p1(in, w) [+ bias])      forward, data gradient (flipped packed weights) and
//                                                                      the linearised forward of the penalty's double backward
//   hwgrad3x3_kernel     gw += gy (x) x over pixels, all nine taps     ds_read_b64_tr_b16 operands, partial blocks +
//                                                                      an ordered finish (bit-reproducible)
//   pool / pack / unpack / add / channel sums                          HBM-bound streaming kernels on 16-byte slots
#include <type_traits>
#include <string.h>
#include "blocked16.h"
#include "split_finish.h"
#include <stdlib.h>
#include <atomic>

namespace srgan {

// ---------------------------------------------------------------------------------------------------- streaming kernels
// fp32 NCHW -> blocked.  One thread per slot (n, group, pixel): eight loads, each coalesced across the lanes along the
// plane.  Optional mask by reference (the gradient w.r.t. an activated tensor arrives here in fp32 from the loss side).
template <int PREC>
__global__ __launch_bounds__(256) void h_pack_kernel(const float* __restrict__ x, Slot* __restrict__ out,
                                                     const Slot* __restrict__ ref, float slope, int64_t slots, int32_t C,
                                                     int32_t CG, int32_t HW) {
  constexpr int G = HGroup<PREC>::N;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < slots; s += (int64_t)gridDim.x * 256) {
    const int pix = (int)(s % HW);
    const int64_t ng = s / HW;
    const int g = (int)(ng % CG);
    const int64_t n = ng / CG;
    float v[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = G * g + j;
      v[j] = c < C ? x[(n * C + c) * HW + pix] : 0.f;
    }
    if (ref) {
      const Slot r = ref[s];
#pragma unroll
      for (int j = 0; j < G; ++j) v[j] *= h_slot_positive<PREC>(r, j) ? 1.f : slope;
    }
    out[s] = h_pack<PREC>(v);
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void h_unpack_kernel(const Slot* __restrict__ x, float* __restrict__ out, int64_t slots,
                                                       int32_t C, int32_t CG, int32_t HW) {
  constexpr int G = HGroup<PREC>::N;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < slots; s += (int64_t)gridDim.x * 256) {
    const int pix = (int)(s % HW);
    const int64_t ng = s / HW;
    const int g = (int)(ng % CG);
    const int64_t n = ng / CG;
    float v[G];
    h_unpack<PREC>(x[s], v);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = G * g + j;
      if (c < C) out[(n * C + c) * HW + pix] = v[j];
    }
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void h_add_kernel(const Slot* __restrict__ a, const Slot* __restrict__ b,
                                                    Slot* __restrict__ out, int64_t slots) {
  constexpr int G = HGroup<PREC>::N;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < slots; s += (int64_t)gridDim.x * 256) {
    float x[G], y[G];
    h_unpack<PREC>(a[s], x);
    h_unpack<PREC>(b[s], y);
#pragma unroll
    for (int j = 0; j < G; ++j) x[j] += y[j];
    out[s] = h_pack<PREC>(x);
  }
}

// Channel sums (a bias gradient): part[(g * parts + part) * 8 + j] = sum over this block's share of (n, pixel) of channel
// 8g + j; h_channel_sums_finish_kernel adds the parts of a channel in part order into the fp32 gradient (bit-reproducible).
// `tickets` != NULL: ONE launch -- the group's last workgroup (a ticket per group and stream, split_finish.h) adds the parts in a
// fixed order and accumulates into `out` itself (the second launch was 47 launches of ~7 us per step of age-vgg-bf16).
__device__ unsigned int g_h_sum_tickets[SPLIT_TICKET_SETS * ROW_FINISH_ROWS];

template <int PREC>
__global__ __launch_bounds__(256) void h_channel_sums_kernel(const Slot* __restrict__ x, float* __restrict__ part, int32_t N,
                                                             int32_t CG, int32_t HW, int32_t parts, unsigned int* tickets,
                                                             float* __restrict__ out, int32_t C) {
  constexpr int G = HGroup<PREC>::N;
  __shared__ float scratch[4 * 8];
  __shared__ float finish_scratch[4];
  const int g = (int)blockIdx.x, part_id = (int)blockIdx.y;
  const int64_t total = (int64_t)N * HW;
  float sum[G];
#pragma unroll
  for (int j = 0; j < G; ++j) sum[j] = 0.f;
  for (int64_t i = (int64_t)part_id * 256 + threadIdx.x; i < total; i += (int64_t)parts * 256) {
    const int64_t n = i / HW;
    const int pix = (int)(i - n * HW);
    float v[G];
    h_unpack<PREC>(x[(n * CG + g) * HW + pix], v);
#pragma unroll
    for (int j = 0; j < G; ++j) sum[j] += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const float w = wave_sum(sum[j]);
    if (lane == 0) scratch[wave * 8 + j] = w;
  }
  __syncthreads();
  if (tickets == nullptr) {
    if (threadIdx.x < G) {
      const int j = threadIdx.x;
      part[((int64_t)g * parts + part_id) * 8 + j] = (scratch[j] + scratch[8 + j]) + (scratch[16 + j] + scratch[24 + j]);
    }
    return;
  }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = j < G ? (scratch[j] + scratch[8 + j]) + (scratch[16 + j] + scratch[24 + j]) : 0.f;
  if (ordered_row_finish<8>(v, part + (int64_t)g * parts * 8, part_id, parts, tickets + g, finish_scratch)) {
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (g * G + j < C) out[g * G + j] += v[j];
  }
}

// `group` = channels per slot (8; 4 for the fp32 blocked form)
__global__ __launch_bounds__(256) void h_channel_sums_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                    int32_t C, int32_t parts, int32_t group) {
  const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (c >= C) return;
  const float* mine = part + (int64_t)(c / group) * parts * 8 + (c % group);
  float total = 0.f;
  for (int s = 0; s < parts; ++s) total += mine[s * 8];
  out[c] += total;
}

// 2x2 / stride 2 max-pool (reference age/vgg.py:76).  The arg-max is not stored: every kernel that needs it finds the FIRST
// window position (scan order (0,0), (0,1), (1,0), (1,1): torch's tie rule) that holds the maximum.
template <int PREC>
__device__ __forceinline__ void h_pool_window(const Slot* __restrict__ x, int64_t base, int32_t W, float (&m)[8], int (&at)[8]) {
  float v[4][8];
  h_unpack8<PREC>(x[base], v[0]);
  h_unpack8<PREC>(x[base + 1], v[1]);
  h_unpack8<PREC>(x[base + W], v[2]);
  h_unpack8<PREC>(x[base + W + 1], v[3]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = v[0][j]; at[j] = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (v[q][j] > m[j]) { m[j] = v[q][j]; at[j] = q; }
  }
}

// mode 0: out = pool(x)             [pooled shape]
// mode 2: out = gather of `g` (input shape) at the arg-max of x      [pooled shape]   (the tangent of the double backward)
template <int PREC, int MODE>
__global__ __launch_bounds__(256) void h_maxpool_kernel(const Slot* __restrict__ x, const Slot* __restrict__ g,
                                                        Slot* __restrict__ out, int64_t planes, int32_t H, int32_t W) {
  const int OH = H / 2, OW = W / 2;
  const int64_t total = planes * OH * OW;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (int64_t)gridDim.x * 256) {
    const int ox = (int)(s % OW);
    const int64_t rest = s / OW;
    const int oy = (int)(rest % OH);
    const int64_t plane = rest / OH;
    const int64_t base = (plane * H + 2 * oy) * W + 2 * ox;
    float m[8]; int at[8];
    h_pool_window<PREC>(x, base, W, m, at);
    if (MODE == 0) {
      out[s] = h_pack8<PREC>(m);
    } else {
      float v[4][8], r[8];
      h_unpack8<PREC>(g[base], v[0]);
      h_unpack8<PREC>(g[base + 1], v[1]);
      h_unpack8<PREC>(g[base + W], v[2]);
      h_unpack8<PREC>(g[base + W + 1], v[3]);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = at[j] == 0 ? v[0][j] : (at[j] == 1 ? v[1][j] : (at[j] == 2 ? v[2][j] : v[3][j]));
      out[s] = h_pack8<PREC>(r);
    }
  }
}

// Backward: gx (input shape) = the pooled gradient placed at the arg-max, times the derivative of the activation that
// produced x (mask by x itself when `masked`): the gradient w.r.t. the PRE-activation tensor in one pass.
template <int PREC>
__global__ __launch_bounds__(256) void h_maxpool_bwd_kernel(const Slot* __restrict__ x, const Slot* __restrict__ gp,
                                                            Slot* __restrict__ gx, int64_t planes, int32_t H, int32_t W,
                                                            int masked, float slope) {
  const int OH = H / 2, OW = W / 2;
  const int64_t total = planes * OH * OW;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (int64_t)gridDim.x * 256) {
    const int ox = (int)(s % OW);
    const int64_t rest = s / OW;
    const int oy = (int)(rest % OH);
    const int64_t plane = rest / OH;
    const int64_t base = (plane * H + 2 * oy) * W + 2 * ox;
    float m[8]; int at[8];
    h_pool_window<PREC>(x, base, W, m, at);
    float g[8];
    h_unpack8<PREC>(gp[s], g);
    if (masked) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] *= m[j] > 0.f ? 1.f : slope;
    }
    float r[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[q][j] = at[j] == q ? g[j] : 0.f;
    gx[base] = h_pack8<PREC>(r[0]);
    gx[base + 1] = h_pack8<PREC>(r[1]);
    gx[base + W] = h_pack8<PREC>(r[2]);
    gx[base + W + 1] = h_pack8<PREC>(r[3]);
  }
}

// ---------------------------------------------------------------------------------------------------- packed weights
// A convolution's weights in operand slots, rounded once per optimizer step (the "16-bit shadow" of the fp32 masters):
//   packed[((chunk * T + tap) * 2 + group) * CO + o] = the 8 channels chunk * 16 + group * 8 ... + 7 of tap `tap` of output
//   channel o, element (o, c, kh, kw) at w[base + o * so + c * si + kh * skh + kw * skw]   (zero beyond CI)
// T = R * S taps.  The data gradient's shadow is the same call with the channel roles swapped and mirrored taps (base at the
// last tap, negative tap strides).
template <int PREC>
__global__ __launch_bounds__(256) void h_pack_conv_weights_kernel(const float* __restrict__ w, Slot* __restrict__ packed,
                                                                  int64_t slots, int32_t CO, int32_t CI, int32_t R, int32_t S,
                                                                  int32_t base, int32_t so, int32_t si, int32_t skh, int32_t skw) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= slots) return;
  h_pack_conv_weights_slot<PREC>(w, packed, slot, CO, CI, R, S, base, so, si, skh, skw);
}

// Every convolution shadow of one network in ONE launch: a workgroup finds its job by bisection over the jobs' first blocks
// (wave-uniform), then runs that job's body.  A step of the driving configuration re-rounds 80 operands, of the VGG
// configuration 52: as single launches they were 0.45 / 0.34 ms of 5 us kernels with the launch gaps on top.
__global__ __launch_bounds__(256) void h_pack_batched_kernel(const HPackJob* __restrict__ jobs, int32_t count) {
  int lo = 0, hi = count - 1;
  const int64_t block = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= block) lo = mid; else hi = mid - 1;
  }
  const HPackJob job = jobs[lo];
  const int64_t slot = (block - job.first_block) * 256 + threadIdx.x;
  if (slot >= job.slots) return;
  const int32_t* q = job.p;
  if (job.kind == 0) {
    if (job.prec == 1) h_pack_conv_weights_slot<1>(job.w, job.packed, slot, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
    else h_pack_conv_weights_slot<2>(job.w, job.packed, slot, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
  } else {
    if (job.prec == 0) h_pack_k4s2_weights_slot<0>(job.w, job.packed, slot, q[0], q[1], q[2], q[3], q[4]);
    else if (job.prec == 1) h_pack_k4s2_weights_slot<1>(job.w, job.packed, slot, q[0], q[1], q[2], q[3], q[4]);
    else h_pack_k4s2_weights_slot<2>(job.w, job.packed, slot, q[0], q[1], q[2], q[3], q[4]);
  }
}

// ---------------------------------------------------------------------------------------------------- 3x3 convolution
struct HConv3Params {
  const Slot* in; const Slot* wp; Slot* out; const float* bias; const Slot* ref;
  float slope;
  int32_t epi;                 // 0 plain, 1 bias (optional) + leaky(slope), 2 multiply by mask(ref, slope)
  int32_t N, CGI, CGO, CO, C_real, H, W;      // CO = rows of the packed weights (>= 8 * CGO is not required: rows beyond are zero)
  int32_t chunks, chunks_per_split;
  int32_t tiles_x, tiles_y, tiles_m;
  float* split_ws; unsigned int* split_tickets;
  int32_t xcd_remap;
  int32_t debug;               // SRGAN_H_EXPERIMENT (timing experiments, wrong results): 1 = no staging after the prologue
};

__device__ unsigned int g_hconv3_split_tickets[SPLIT_TICKET_SETS * SPLIT_TICKET_TILES];

// The BM bias values of a workgroup's rows into LDS, at the kernel's start (the barrier behind it is nowhere near the epilogue's
// stores: a barrier in front of the epilogue would make the four waves' stores wait for the slowest wave's last MFMA).
template <int BM>
__device__ __forceinline__ void hconv3_stage_bias(const float* bias, int epi, int C_real, int m0, float* bias_rows) {
  if (epi == 1 && bias != nullptr) {
    if ((int)threadIdx.x < BM) bias_rows[threadIdx.x] = m0 + (int)threadIdx.x < C_real ? bias[m0 + threadIdx.x] : 0.f;
    __syncthreads();
  }
}

// Epilogue of the 3x3 kernels.  C/D fragment: column = pixel (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): registers
// 4q .. 4q + 3 are four CONSECUTIVE output channels = 8 bytes of the pixel's slot of group (m0 + 32 mi) / 8 + q; the 32 lanes of
// a half cover 32 consecutive slots (with the other half: 512 contiguous bytes per store instruction).
template <int BM, int NI, int TW, int ROWS, int PREC>
__device__ __forceinline__ void hconv3_epilogue(const HConv3Params& p, f32x16 (&acc)[BM / 32][NI], int n0, int y0, int x0, int m0,
                                                int wave, int l31, int lhi, const float* bias_rows) {
  constexpr int MI = BM / 32;
  const int HW = p.H * p.W;
  // bias_rows: the tile's bias values in LDS (hconv3_stage_bias at the kernel's start).  Read per value from global memory
  // inside the loops below they were 4 dependent loads per quad -- 128 VMEM instructions per wave at 64 rows x 512 pixels, 25 us
  // of a 59 us launch (conv1_1, batch 128: scratch/h_conv_bench.py).
  const bool with_bias = p.epi == 1 && p.bias != nullptr;
  float4 bias4[MI][4];              // this lane's rows are the same for every pixel column: read once
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      bias4[mi][qd] = with_bias ? *reinterpret_cast<const float4*>(&bias_rows[mi * 32 + 8 * qd + 4 * lhi]) : make_float4(0.f, 0.f, 0.f, 0.f);
  // epi 2: the mask references of a pixel column are loaded in ONE batch in front of its stores.  On this ISA loads and stores
  // return through one in-order counter (vmcnt): interleaved load - use - store, every load waited for the stores in front of it
  // (32 round trips per wave at 64 rows x 512 pixels: 102 us for the data gradient of a layer whose forward took 69).
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = (wave * NI + ni) * 32 + l31;
    const int n = n0 + q / (ROWS * TW), y = y0 + (q / TW) % ROWS, x = x0 + q % TW;
    if (n >= p.N || y >= p.H || x >= p.W) continue;
    const int64_t pixel = (int64_t)n * p.CGO * HW + y * p.W + x;
    uint2 refs[MI][4];
    if (p.epi == 2) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int group = (m0 + mi * 32) / 8 + qd;
          refs[mi][qd] = group < p.CGO ? *(reinterpret_cast<const uint2*>(p.ref + pixel + (int64_t)group * HW) + lhi) : uint2{0u, 0u};
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        // two register quads = this lane's half (channels 4 lhi .. 4 lhi + 3) of the slots of groups g0 and g0 + 1
        const int g0 = (m0 + mi * 32) / 8 + 2 * qp;
        uint2 packed[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int qd = 2 * qp + h;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * qd + j];
          if (p.epi == 1) {
            const float4 b4 = bias4[mi][qd];
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
          } else if (p.epi == 2) {
            const uint2 r = refs[mi][qd];
            v[0] *= h_mask(r.x & 0xFFFFu, p.slope); v[1] *= h_mask(r.x >> 16, p.slope);
            v[2] *= h_mask(r.y & 0xFFFFu, p.slope); v[3] *= h_mask(r.y >> 16, p.slope);
          }
          packed[h].x = h_pack2<PREC>(v[0], v[1]);
          packed[h].y = h_pack2<PREC>(v[2], v[3]);
        }
        // v_permlane32_swap: the upper half-wave's value of the first operand changes places with the lower half-wave's of the
        // second.  Before: lane l (< 32) holds channels 0-3 of (g0, pixel l) and of (g0 + 1, pixel l), lane l + 32 channels 4-7 of
        // both.  After: lane l holds the WHOLE slot of (g0, pixel l), lane l + 32 the whole slot of (g0 + 1, pixel l): one
        // 16-byte store per lane (2 x 512 contiguous bytes per instruction) instead of two 8-byte stores (0-7 % per launch).
        const auto first = __builtin_amdgcn_permlane32_swap(packed[0].x, packed[1].x, false, false);
        const auto second = __builtin_amdgcn_permlane32_swap(packed[0].y, packed[1].y, false, false);
        const int group = g0 + lhi;
        if (group < p.CGO) {
          uint4 whole;
          whole.x = first[0]; whole.y = second[0]; whole.z = first[1]; whole.w = second[1];
          *reinterpret_cast<uint4*>(p.out + pixel + (int64_t)group * HW) = whole;
        }
      }
    }
  }
}

// One workgroup (4 waves) = BM output channels x P = 128 * NI pixels.  The pixel tile is IMG x ROWS x TW with
// IMG * ROWS * TW = P: a ROWS x 32 band of one image on wide planes, whole small images side by side on the 16 / 8 / 4
// pixel planes of VGG's later stages (a one-image tile would leave most MFMA columns outside the image there).  Per
// 16-channel chunk the halo patch [2 groups][IMG][(ROWS + 2) x (TW + 2)] and the weight slice [9 taps][2 groups][BM] are
// staged in LDS as operand slots (register-staged: the next chunk's 16-byte loads are in flight during the current
// chunk's 9 * MI * NI MFMAs) and every fragment is one ds_read_b128.
// STAGES = 2: two LDS stages, one barrier per chunk; STAGES = 1 (the 512-pixel tiles, whose two stages would not fit two
// workgroups per CU): one stage, the next chunk waits in registers, two barriers per (twice as long) chunk.
template <int BM, int NI, int TW, int ROWS, int PREC, int STAGES = 2>
__global__ __launch_bounds__(256, 2) void hconv3x3_kernel(const HConv3Params p) {
  constexpr int P = 128 * NI, IMG = P / (ROWS * TW), PW = TW + 2, PH = ROWS + 2, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int MI = BM / 32;
  constexpr int PATCH_G = IMG * PLANE, PATCH_Q = 2 * PATCH_G, WT_Q = 18 * BM, STAGE_Q = PATCH_Q + WT_Q;
  constexpr int NP = (PATCH_Q + 255) / 256, NW = (WT_Q + 255) / 256;
  __shared__ Slot lds[STAGES * STAGE_Q];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  int block = blockIdx.x;
  if (p.xcd_remap) block = (block & 7) * ((int)gridDim.x >> 3) + (block >> 3);
  const int logical_block = block;
  const int tm = block % p.tiles_m; block /= p.tiles_m;
  const int tx = block % p.tiles_x; block /= p.tiles_x;
  const int ty = block % p.tiles_y;
  const int n0 = (block / p.tiles_y) * IMG;
  const int m0 = tm * BM, y0 = ty * ROWS, x0 = tx * TW;
  __shared__ __attribute__((aligned(16))) float bias_rows[BM];
  hconv3_stage_bias<BM>(p.bias, p.epi, p.C_real, m0, bias_rows);
  const int cbeg = (int)blockIdx.y * p.chunks_per_split;
  const int cend = min(p.chunks, cbeg + p.chunks_per_split);
  const int HW = p.H * p.W;

  int poff[NP], pgrp[NP], woff[NW];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const int flat = e * 256 + tid;
    const int g = flat / PATCH_G, rest = flat - g * PATCH_G;
    const int img = rest / PLANE, pix = rest - img * PLANE;
    const int py = pix / PW, px = pix - py * PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = flat < PATCH_Q && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && n0 + img < p.N;
    poff[e] = ok ? img * p.CGI * HW + y * p.W + x : -1;
    pgrp[e] = g;
  }
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int flat = e * 256 + tid;
    const int o = flat % BM, tg = flat / BM;
    const bool ok = flat < WT_Q && (m0 + o) < p.CO;
    woff[e] = ok ? tg * p.CO + m0 + o : -1;
  }
  const Slot* in_n = p.in + (int64_t)n0 * p.CGI * HW;

  Slot rp[NP], rw[NW];
  auto fetch = [&](int c) {                                 // raw loads only: validity is applied at stage time
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int g = 2 * c + pgrp[e];
      const bool ok = poff[e] >= 0 && g < p.CGI;
      rp[e] = in_n[ok ? g * HW + poff[e] : 0];
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) rw[e] = p.wp[(int64_t)c * (18 * p.CO) + (woff[e] >= 0 ? woff[e] : 0)];
  };
  auto stage = [&](int c, Slot* stage_base) {
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int flat = e * 256 + tid;
      Slot v = rp[e];
      if (!(poff[e] >= 0 && 2 * c + pgrp[e] < p.CGI)) v = Slot{{0u, 0u, 0u, 0u}};
      if (flat < PATCH_Q) stage_base[flat] = v;
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int flat = e * 256 + tid;
      Slot v = rw[e];
      if (woff[e] < 0) v = Slot{{0u, 0u, 0u, 0u}};
      if (flat < WT_Q) stage_base[PATCH_Q + flat] = v;
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  int b_lane[NI];                                           // this lane's pixel of column block ni in the patch (window origin)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = (wave * NI + ni) * 32 + l31;
    b_lane[ni] = lhi * PATCH_G + (q / (ROWS * TW)) * PLANE + ((q / TW) % ROWS) * PW + q % TW;
  }
  const int a_lane = PATCH_Q + lhi * BM + l31;              // + tap * 2 * BM + mi * 32

  auto multiply = [&](const Slot* st) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
      Slot a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = st[a_lane + tap * 2 * BM + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = st[b_lane[ni] + kh * PW + kw];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = h_mfma<PREC>(a[mi], b[ni], acc[mi][ni]);
    }
  };
  if (cbeg < cend) {
    fetch(cbeg);
    if constexpr (STAGES == 2) {
      stage(cbeg, lds);
      __syncthreads();
      int cur = 0;
      for (int c = cbeg; c < cend; ++c) {
        const bool more = c + 1 < cend;
        if (more) fetch(c + 1);
        multiply(lds + cur * STAGE_Q);
        if (more) stage(c + 1, lds + (cur ^ 1) * STAGE_Q);     // the other stage: everyone left it at the previous barrier
        __syncthreads();
        cur ^= 1;
      }
    } else {
      for (int c = cbeg; c < cend; ++c) {
        if (c > cbeg) __syncthreads();                          // the previous chunk's fragment reads are done
        stage(c, lds);
        __syncthreads();
        if (c + 1 < cend) fetch(c + 1);
        multiply(lds);
      }
    }
  }

  if (gridDim.y > 1) {       // K split, ordered finish: the tile's last workgroup goes on with the sum of all slices, in slice order
    constexpr int COUNT = MI * NI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.split_ws + (int64_t)logical_block * gridDim.y * (COUNT * 256), (int)blockIdx.y,
                                          (int)gridDim.y, p.split_tickets + logical_block,
                                          [&](int i) { return acc[i / (NI * 16)][(i / 16) % NI][i % 16]; },
                                          [&](int i, float v) { acc[i / (NI * 16)][(i / 16) % NI][i % 16] = v; }))
      return;
  }

  hconv3_epilogue<BM, NI, TW, ROWS, PREC>(p, acc, n0, y0, x0, m0, wave, l31, lhi, bias_rows);
}

// ---- the same convolution with LDS-DMA staging (global_load_lds_dwordx4: HBM / L2 -> LDS without passing the registers) -----
// The register-staged kernel above has ONE chunk in flight per workgroup and pays a ds_write phase per chunk; the counters
// (profiles/r06f) show its waves parked on memory 46 % of the time at 1.4x the algorithmic HBM bytes: latency, not bandwidth.
// Here a chunk's patch and weight slots are fetched by DMA into a ring of RING stages, RING - 1 chunks ahead, with no staging
// registers and no LDS-write instructions.  An LDS-DMA writes lane-linearly (wave-uniform LDS base + lane * 16) from PER-LANE
// global addresses, and the stage image [patch slots, rounded up to whole 64-slot instructions][weight slots][padding to a
// multiple of four instructions] is linear in the staging index, so instruction i of a chunk is "slots 64 i .. 64 i + 63" and
// the halo gather lives entirely in the lanes' source addresses; slots that are padding (outside the image, beyond the
// channels) read a 16-byte block of zeros in global memory.  Wave w issues instructions w, w + 4, ...: every wave exactly
// L = TP / 4, so `s_waitcnt vmcnt(L * chunks left in flight)` is exact; one raw s_barrier per chunk (a __syncthreads() would
// drain the DMA queue).  No K split (the small layers keep the register-staged kernel).
__device__ Slot g_h_zero_slots[4];
const Slot* h_zero_slots() { return reinterpret_cast<const Slot*>(device_tickets(g_h_zero_slots)); }

template <int BM, int NI, int TW, int ROWS, int PREC, int RING>
__global__ __launch_bounds__(256, RING == 2 ? 2 : 1) void hconv3x3_dma_kernel(const HConv3Params p, const Slot* zero) {
  constexpr int P = 128 * NI, IMG = P / (ROWS * TW), PW = TW + 2, PH = ROWS + 2, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int MI = BM / 32;
  constexpr int PATCH_G = IMG * PLANE, PATCH_Q = 2 * PATCH_G, WT_Q = 18 * BM;
  constexpr int PATCH_I = (PATCH_Q + 63) / 64, WT_I = WT_Q / 64, T = PATCH_I + WT_I, TP = (T + 3) / 4 * 4, L = TP / 4;
  constexpr int WT_OFF = PATCH_I * 64, STAGE_Q = TP * 64;
  static_assert(WT_Q % 64 == 0, "whole weight instructions");
  extern __shared__ __attribute__((aligned(16))) Slot ring[];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int block = blockIdx.x;
  if (p.xcd_remap) block = (block & 7) * ((int)gridDim.x >> 3) + (block >> 3);
  const int tm = block % p.tiles_m; block /= p.tiles_m;
  const int tx = block % p.tiles_x; block /= p.tiles_x;
  const int ty = block % p.tiles_y;
  const int n0 = (block / p.tiles_y) * IMG;
  const int m0 = tm * BM, y0 = ty * ROWS, x0 = tx * TW;
  __shared__ __attribute__((aligned(16))) float bias_rows[BM];
  const int HW = p.H * p.W;
  const uint32_t lds0 = h_lds_address(ring);

  // this lane's source of instruction wave + 4 e: an offset (slots) from the input of image n0 (patch; + group * HW + the chunk's
  // 2 * HW * c) or from the packed weights (+ 18 * CO * c); kind 0 patch group 0, 1 patch group 1, 2 weights, 3 zeros
  int off[L], kind[L];
#pragma unroll
  for (int e = 0; e < L; ++e) {
    const int i = wave + 4 * e;
    off[e] = 0; kind[e] = 3;
    if (i < PATCH_I) {
      const int flat = i * 64 + lane;
      const int g = flat / PATCH_G, rest = flat - g * PATCH_G;
      const int img = rest / PLANE, pix = rest - img * PLANE;
      const int py = pix / PW, px = pix - py * PW;
      const int y = y0 - 1 + py, x = x0 - 1 + px;
      const bool ok = flat < PATCH_Q && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && n0 + img < p.N;
      if (ok) { off[e] = (img * p.CGI + g) * HW + y * p.W + x; kind[e] = g; }
    } else if (i < T) {
      const int flat = (i - PATCH_I) * 64 + lane;
      const int o = flat % BM, tg = flat / BM;
      if (m0 + o < p.CO) { off[e] = tg * p.CO + m0 + o; kind[e] = 2; }
    }
  }
  const Slot* in_n = p.in + (int64_t)n0 * p.CGI * HW;
  auto issue = [&](int c, int stage) {
#pragma unroll
    for (int e = 0; e < L; ++e) {
      const Slot* src = zero;
      if (kind[e] == 2) src = p.wp + ((int64_t)c * (18 * p.CO) + off[e]);
      else if (kind[e] < 2 && 2 * c + kind[e] < p.CGI) src = in_n + ((int64_t)c * (2 * HW) + off[e]);
      h_glds16(src, lds0 + (uint32_t)((stage * STAGE_Q + (wave + 4 * e) * 64) * 16));
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  int b_lane[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = ((tid >> 6) * NI + ni) * 32 + l31;
    b_lane[ni] = lhi * PATCH_G + (q / (ROWS * TW)) * PLANE + ((q / TW) % ROWS) * PW + q % TW;
  }
  const int a_lane = WT_OFF + lhi * BM + l31;

  const int chunks = p.chunks;
#pragma unroll
  for (int d = 0; d < RING - 1; ++d)
    if (d < chunks) issue(d, d);
  if ((p.debug & 1) && RING - 1 < chunks) issue(RING - 1, RING - 1);       // (experiment: every stage holds finite data)
  hconv3_stage_bias<BM>(p.bias, p.epi, p.C_real, m0, bias_rows);           // behind the first chunk's requests, not in front of them
  int stage = 0;
  for (int c = 0; c < chunks; ++c) {
    // chunk c has landed (this wave's part: vmcnt; everybody's: the barrier), and everybody is done reading the stage that
    // chunk c + RING - 1 goes into (it held chunk c - 1)
    if (RING == 3 && c + 1 < chunks) h_dma_wait_and_barrier<L>();
    else h_dma_wait_and_barrier<0>();
    if (c + RING - 1 < chunks && !(p.debug & 1)) issue(c + RING - 1, (stage + RING - 1) % RING);
    const Slot* st = ring + stage * STAGE_Q;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
      Slot a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = st[a_lane + tap * 2 * BM + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = st[b_lane[ni] + kh * PW + kw];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = h_mfma<PREC>(a[mi], b[ni], acc[mi][ni]);
    }
    stage = stage + 1 == RING ? 0 : stage + 1;
  }
  hconv3_epilogue<BM, NI, TW, ROWS, PREC>(p, acc, n0, y0, x0, m0, tid >> 6, l31, lhi, bias_rows);
}

struct HConv3Plan { int bm, ni, tw, rows, tiles_x, tiles_y, tiles_m, tiles_n, split, chunks_per; int64_t blocks; };

static bool hconv3_plan(int32_t N, int32_t CGI, int32_t CO_rows, int32_t H, int32_t W, HConv3Plan& plan) {
  // tile width and the image rows of a 128 * NI pixel tile (NI = 1, 2, 4; 0 = that tile does not exist for the plane)
  int tw, rows_for_ni[5] = {0, 0, 0, 0, 0};
  if (W > 16) { tw = 32; rows_for_ni[1] = 4; rows_for_ni[2] = 8; rows_for_ni[4] = 16; }
  else if (W == 8 && H == 8) { tw = 8; rows_for_ni[1] = rows_for_ni[2] = rows_for_ni[4] = 8; }   // whole images side by side
  else if (W == 4 && H == 4) { tw = 4; rows_for_ni[1] = 4; }                                      // (32 KB of LDS per stage at NI = 2)
  else { tw = 16; rows_for_ni[1] = 8; rows_for_ni[2] = 16; if (H == 16) rows_for_ni[4] = 16; }    // (NI = 4: two whole images)
  static const bool no_wide = getenv("SRGAN_H_NO_WIDE_TILE") != nullptr;
  if (no_wide) rows_for_ni[4] = 0;
  plan.tw = tw;
  plan.bm = CO_rows > 32 ? 64 : 32;
  plan.tiles_m = (CO_rows + plan.bm - 1) / plan.bm;
  auto count = [&](int ni) {
    const int rows = rows_for_ni[ni];
    const int img = 128 * ni / (rows * tw);
    const int64_t tx = (W + tw - 1) / tw, ty = img > 1 ? 1 : (H + rows - 1) / rows, tn = (N + img - 1) / img;
    return tx * ty * tn * plan.tiles_m;
  };
  // the largest pixel tile that still gives two workgroups per CU: the weight slice of a chunk (18 KB at 64 rows) is staged once
  // per tile, and the L2 -> LDS stream, not the matrix pipe, bounds these kernels
  int ni = 1;
  if (rows_for_ni[4] && plan.bm == 64 && count(4) >= 512) ni = 4;
  else if (rows_for_ni[2] && count(2) >= 512) ni = 2;
  if (const char* forced = getenv("SRGAN_H_CONV_NI")) {          // tests: every tile shape on small tensors
    const int want = atoi(forced);
    if ((want == 1 || want == 2 || want == 4) && rows_for_ni[want] && (want != 4 || plan.bm == 64)) ni = want;
  }
  plan.ni = ni;
  plan.rows = rows_for_ni[ni];
  const int img = 128 * ni / (plan.rows * tw);
  plan.tiles_x = (W + tw - 1) / tw;
  plan.tiles_y = img > 1 ? 1 : (H + plan.rows - 1) / plan.rows;
  plan.tiles_n = (N + img - 1) / img;
  plan.blocks = count(ni);
  if (plan.blocks >= ((int64_t)1 << 31)) return false;
  const int chunks = (CGI + 1) / 2;
  int split = 1;
  if (plan.blocks < 384 && chunks >= 4 && plan.blocks <= SPLIT_TICKET_TILES) {
    split = (int)((512 + plan.blocks - 1) / plan.blocks);
    if (split > chunks / 2) split = chunks / 2;
    if (split > 16) split = 16;
  }
  plan.chunks_per = (chunks + split - 1) / split;
  plan.split = (chunks + plan.chunks_per - 1) / plan.chunks_per;
  return true;
}

template <int BM, int NI, int PREC>
static void hconv3_launch_tiles(const HConv3Params& p, int tw, dim3 grid, hipStream_t stream) {
  if constexpr (NI == 4) {
    if (tw == 32) hipLaunchKernelGGL((hconv3x3_kernel<BM, 4, 32, 16, PREC, 1>), grid, dim3(256), 0, stream, p);
    else if (tw == 16) hipLaunchKernelGGL((hconv3x3_kernel<BM, 4, 16, 16, PREC, 1>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((hconv3x3_kernel<BM, 4, 8, 8, PREC, 1>), grid, dim3(256), 0, stream, p);
  } else {
    if (tw == 32) hipLaunchKernelGGL((hconv3x3_kernel<BM, NI, 32, 4 * NI, PREC>), grid, dim3(256), 0, stream, p);
    else if (tw == 16) hipLaunchKernelGGL((hconv3x3_kernel<BM, NI, 16, 8 * NI, PREC>), grid, dim3(256), 0, stream, p);
    else if (tw == 8) hipLaunchKernelGGL((hconv3x3_kernel<BM, NI, 8, 8, PREC>), grid, dim3(256), 0, stream, p);
    else if constexpr (NI == 1) hipLaunchKernelGGL((hconv3x3_kernel<BM, 1, 4, 4, PREC>), grid, dim3(256), 0, stream, p);
  }
}

template <int PREC>
static void hconv3_launch(const HConv3Params& p, const HConv3Plan& plan, dim3 grid, hipStream_t stream) {
  if (plan.bm == 64) {
    if (plan.ni == 4) hconv3_launch_tiles<64, 4, PREC>(p, plan.tw, grid, stream);
    else if (plan.ni == 2) hconv3_launch_tiles<64, 2, PREC>(p, plan.tw, grid, stream);
    else hconv3_launch_tiles<64, 1, PREC>(p, plan.tw, grid, stream);
  } else {
    if (plan.ni == 2) hconv3_launch_tiles<32, 2, PREC>(p, plan.tw, grid, stream);
    else hconv3_launch_tiles<32, 1, PREC>(p, plan.tw, grid, stream);
  }
}

// ---- launchers of the LDS-DMA ring kernel (64-row tiles, unsplit K) -------------------------------------------------------
template <int NI, int TW, int ROWS, int PREC, int RING>
static int hconv3_dma_launch_one(const HConv3Params& p, dim3 grid, hipStream_t stream, const Slot* zero) {
  constexpr int P = 128 * NI, IMG = P / (ROWS * TW), PLANE = (ROWS + 2) * (TW + 2);
  constexpr int PATCH_I = (2 * IMG * PLANE + 63) / 64, T = PATCH_I + 18, TP = (T + 3) / 4 * 4;
  constexpr int bytes = RING * TP * 64 * 16;
  static_assert(bytes <= 160 * 1024, "the ring fits the CU's LDS");
  auto kernel = hconv3x3_dma_kernel<64, NI, TW, ROWS, PREC, RING>;
  // more than 64 KB of dynamic LDS needs the attribute, once per kernel and device (as pointwise_ring.hip)
  static std::atomic<uint64_t> configured_devices{0};
  int device = 0;
  SRGAN_HIP(hipGetDevice(&device));
  const uint64_t bit = (uint64_t)1 << (device & 63);
  if (!(configured_devices.load(std::memory_order_acquire) & bit)) {
    SRGAN_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    configured_devices.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kernel, grid, dim3(256), bytes, stream, p, zero);
  return SRGAN_OK;
}

template <int PREC, int RING>
static int hconv3_dma_launch(const HConv3Params& p, const HConv3Plan& plan, dim3 grid, hipStream_t stream, const Slot* zero) {
  const int tw = plan.tw;
  if (plan.ni == 4) {
    if (tw == 32) return hconv3_dma_launch_one<4, 32, 16, PREC, RING>(p, grid, stream, zero);
    if (tw == 16) return hconv3_dma_launch_one<4, 16, 16, PREC, RING>(p, grid, stream, zero);
    return hconv3_dma_launch_one<4, 8, 8, PREC, RING>(p, grid, stream, zero);
  }
  if (plan.ni == 2) {
    if (tw == 32) return hconv3_dma_launch_one<2, 32, 8, PREC, RING>(p, grid, stream, zero);
    if (tw == 16) return hconv3_dma_launch_one<2, 16, 16, PREC, RING>(p, grid, stream, zero);
    return hconv3_dma_launch_one<2, 8, 8, PREC, RING>(p, grid, stream, zero);
  }
  if (tw == 32) return hconv3_dma_launch_one<1, 32, 4, PREC, RING>(p, grid, stream, zero);
  if (tw == 16) return hconv3_dma_launch_one<1, 16, 8, PREC, RING>(p, grid, stream, zero);
  if (tw == 8) return hconv3_dma_launch_one<1, 8, 8, PREC, RING>(p, grid, stream, zero);
  return hconv3_dma_launch_one<1, 4, 4, PREC, RING>(p, grid, stream, zero);
}

// ---------------------------------------------------------------------------------------------------- 3x3 weight gradient
//   gw[co, ci, kh, kw] += sum_{n, y, x} gy[n, co, y, x] * x[n, ci, y + kh - 1, x + kw - 1]        (fp32, [CO][CI][3][3])
// M = co, N = ci, K = pixels: both operands are reduced over the pixel index, which the blocked layout does NOT keep
// contiguous per channel -- the fragments come from LDS through ds_read_b64_tr_b16 (blocked16.h).  A workgroup (4 waves) owns
// a 64 x 64 (co x ci) block of gw for ALL nine taps: wave (mi, ni) keeps nine 32 x 32 accumulators (144 registers) while the
// workgroup walks its share of the 64-pixel tiles; per tile the gy slots [8 groups][64 pixels] and the x halo patch
// [8 groups][IMG][(ROWS + 2) x (TW + 2)] are staged once (register-staged, one LDS stage) and all nine taps read them: per
// 16-pixel step one A fragment and nine shifted B fragments, two transpose reads each.  The walkers of a block leave their
// accumulators in the caller's workspace in thread order; hwgrad3x3_finish_kernel adds them in walker order into gw.
struct HWgrad3Params {
  const Slot* x; const Slot* gy; float* partial;
  int32_t N, CGX, CGY, H, W;
  int32_t tiles_ci, tiles_x, tiles_y, tiles_n, pixel_tiles, walkers;
};

constexpr int HWGRAD_P = 64;      // pixels per staged tile (128 needs 44+ staging registers next to the 144 accumulators: spills)
constexpr int h_pad_stride(int slots) { return ((slots + 15) / 16) * 16 + 4; }     // = 4 slots (mod 16): the two groups of a
                                                                                  // transpose read land 64 bytes apart
// MB = 32-row blocks of output channels per workgroup: 2 (64 x 64 block, 4 waves, two workgroups per CU) or 4 (128 x 64 block,
// 8 waves = two per SIMD in one workgroup: the x patch, two thirds of the staged bytes, is then shared by twice the matrix work).
template <int TW, int ROWS, int PREC, int MB>
__global__ __launch_bounds__(128 * MB, 2) void hwgrad3x3_kernel(const HWgrad3Params p) {
  constexpr int THREADS = 128 * MB;
  constexpr int P = HWGRAD_P, IMG = P / (ROWS * TW), PW = TW + 2, PH = ROWS + 2, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int GS = h_pad_stride(P), XS = h_pad_stride(IMG * PLANE);       // group strides of the two LDS images (slots)
  constexpr int GQ = 4 * MB * P, XQ = 8 * IMG * PLANE;                      // slots staged per tile
  constexpr int NG = GQ / THREADS, NX = (XQ + THREADS - 1) / THREADS;
  __shared__ Slot lds[4 * MB * GS + 8 * XS];
  Slot* gs = lds;
  Slot* xs = lds + 4 * MB * GS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave >> 1, ni = wave & 1;
  const int tci = (int)blockIdx.x % p.tiles_ci, tco = (int)blockIdx.x / p.tiles_ci;
  const int walker = (int)blockIdx.y;
  const int HW = p.H * p.W;

  // staging ownership: gy slot e * THREADS + tid -> (group, tile pixel); x slot -> (group, image, patch position); the index
  // arithmetic (compile-time divisors) is redone where it is used instead of being kept in registers
  Slot rg[NG], rx[NX];
  uint32_t okg = 0, okx = 0;
  auto fetch = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int rest_t = tile / p.tiles_x;
    const int ty = rest_t % p.tiles_y;
    const int n0 = (rest_t / p.tiles_y) * IMG;
    const int y0 = ty * ROWS, x0 = tx * TW;
    okg = okx = 0;
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int flat = e * THREADS + tid;
      const int grp = flat / P, q = flat % P;
      const int n = n0 + q / (ROWS * TW), y = y0 + (q / TW) % ROWS, x = x0 + q % TW;
      const int group = tco * (4 * MB) + grp;
      const bool ok = n < p.N && y < p.H && x < p.W && group < p.CGY;
      okg |= (ok ? 1u : 0u) << e;
      rg[e] = p.gy[ok ? ((int64_t)n * p.CGY + group) * HW + y * p.W + x : 0];
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int flat = e * THREADS + tid;
      const int grp = flat / (IMG * PLANE), rest = flat - grp * (IMG * PLANE);
      const int img = rest / PLANE, pix = rest % PLANE;
      const int n = n0 + img, y = y0 - 1 + pix / PW, x = x0 - 1 + pix % PW;
      const int group = tci * 8 + grp;
      const bool ok = flat < XQ && n < p.N && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && group < p.CGX;
      okx |= (ok ? 1u : 0u) << e;
      rx[e] = p.x[ok ? ((int64_t)n * p.CGX + group) * HW + y * p.W + x : 0];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Fragment addressing (blocked16.h, h_tr_read): 16-lane group G = lane >> 4: row block rb = G & 1 (channels 16 rb .. of the
  // wave's 32), k half = G >> 1 (pixels 8 khalf .. of the 16-pixel step); within the group lane s = 4 j + u supplies pixel j
  // of its quad, channels 4 u .. 4 u + 3.
  const int G = lane >> 4, rb = G & 1, khalf = G >> 1, s = lane & 15, j = s >> 2, u = s & 3;
  constexpr int HALF = TW >= 16 ? 8 : (TW == 8 ? PW : 2 * PW);     // patch offset of pixel 8 of a 16-pixel step
  constexpr int QUAD = TW >= 8 ? 4 : PW;                           // ... of the second quad of a half
  const uint32_t a_base = h_lds_address(gs) + (uint32_t)(((4 * mi + 2 * rb + (u >> 1)) * GS + 8 * khalf + j) * 16 + (u & 1) * 8);
  const uint32_t b_base = h_lds_address(xs) + (uint32_t)(((4 * ni + 2 * rb + (u >> 1)) * XS + khalf * HALF + j) * 16 + (u & 1) * 8);

  // a wave whose 32 x 32 block lies entirely beyond the channels that exist (3 input channels: conv1_1) only helps staging
  const bool active = (tco * (32 * MB) + mi * 32) < p.CGY * 8 && (tci * 64 + ni * 32) < p.CGX * 8;
  int tile = walker;
  if (tile < p.pixel_tiles) fetch(tile);
  for (; tile < p.pixel_tiles; tile += p.walkers) {
    __syncthreads();                        // the previous tile's reads are done
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int flat = e * THREADS + tid;
      Slot v = rg[e];
      if (!((okg >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      gs[(flat / P) * GS + flat % P] = v;
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int flat = e * THREADS + tid;
      const int grp = flat / (IMG * PLANE), rest = flat - grp * (IMG * PLANE);
      Slot v = rx[e];
      if (!((okx >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      if (flat < XQ) xs[grp * XS + rest] = v;
    }
    __syncthreads();
    const int next = tile + p.walkers;
    if (next < p.pixel_tiles) fetch(next);
    if (active)
#pragma unroll 2
    for (int t = 0; t < P / 16; ++t) {       // (not fully unrolled: the scheduler otherwise hoists the transpose reads of many steps and spills)
      const int pix = 16 * t;                                                        // first pixel of the step (tile order)
      const int origin = ((pix / (ROWS * TW)) * PH + (pix / TW) % ROWS) * PW + pix % TW;   // its window origin in the patch
      Slot a;
      const uint2 a0 = h_tr_read(a_base + pix * 16), a1 = h_tr_read(a_base + (pix + 4) * 16);
      a.v[0] = a0.x; a.v[1] = a0.y; a.v[2] = a1.x; a.v[3] = a1.y;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int shift = origin + (tap / 3) * PW + tap % 3;
        Slot b;
        const uint2 b0 = h_tr_read(b_base + shift * 16), b1 = h_tr_read(b_base + (shift + QUAD) * 16);
        b.v[0] = b0.x; b.v[1] = b0.y; b.v[2] = b1.x; b.v[3] = b1.y;
        acc[tap] = h_mfma<PREC>(a, b, acc[tap]);
      }
    }
  }

  float* mine = p.partial + ((int64_t)blockIdx.x * p.walkers + walker) * (9 * 16 * THREADS) + tid;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(t * 16 + r) * THREADS] = acc[t][r];
}

// gw[co][ci][tap] += sum over the walkers of the block's partial accumulators, in a FIXED order.  One workgroup per (block,
// output channel): the 64 input channels x 9 taps of that row are 576 consecutive floats of gw.  Lane l of every wave reads
// input channel l of each tap -- two 128-byte runs of the partial block per (walker, tap), whole lines: the first version gave a
// workgroup 64 consecutive gw elements = 7 of a line's 32 floats and fetched every line 4.5 times (18 GB per step, PMC
// profiles/r06f) -- the four waves add the walkers w = wave, wave + 4, ... and the four sums meet as (0 + 1) + (2 + 3); the
// row leaves through LDS in gw's own order.  The read index undoes the accumulator layout: wave = (co' / 32) * 2 + ci' / 32,
// C/D row co' % 32 = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column ci' % 32 = lane & 31.
__global__ __launch_bounds__(256) void hwgrad3x3_finish_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                               int32_t CO, int32_t CI, int32_t tiles_ci, int32_t walkers,
                                                               int32_t block_rows) {
  __shared__ float sums[4][9][64];
  const int lane = (int)threadIdx.x & 63, part = (int)threadIdx.x >> 6;
  const int block = (int)blockIdx.x / block_rows, row = (int)blockIdx.x % block_rows;
  const int tco = block / tiles_ci, tci = block % tiles_ci;
  const int co = tco * block_rows + row;
  if (co >= CO) return;
  const int threads = 4 * block_rows;
  const int64_t per_walker = (int64_t)9 * 16 * threads;              // accumulators of one workgroup
  const int r32 = row & 31, lhi = (r32 >> 2) & 1, r = (r32 & 3) + 4 * (r32 >> 3);
  const int thread = ((row >> 5) * 2 + (lane >> 5)) * 64 + lhi * 32 + (lane & 31);
  const float* base = partial + (int64_t)block * walkers * per_walker + r * threads + thread;
  float total[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) total[t] = 0.f;
  // four walkers' loads in flight per wave, added in walker order (the sum is the one-at-a-time loop's, bit for bit): the
  // single-block layers have 512 walkers = 128 dependent round trips per wave
  for (int w0 = part; w0 < walkers; w0 += 16) {
    float v[4][9];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int w = w0 + 4 * d;
#pragma unroll
      for (int t = 0; t < 9; ++t) v[d][t] = w < walkers ? base[(int64_t)w * per_walker + t * 16 * threads] : 0.f;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (w0 + 4 * d < walkers)
#pragma unroll
        for (int t = 0; t < 9; ++t) total[t] += v[d][t];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) sums[part][t][lane] = total[t];
  __syncthreads();
  float previous[3];                        // read all, then write all (loads wait for the stores in front of them: one vmcnt)
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int i = (int)threadIdx.x + 256 * e, ci_local = i / 9, tap = i - ci_local * 9, ci = tci * 64 + ci_local;
    previous[e] = (i < 576 && ci < CI) ? gw[((int64_t)co * CI + ci) * 9 + tap] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int i = (int)threadIdx.x + 256 * e, ci_local = i / 9, tap = i - ci_local * 9, ci = tci * 64 + ci_local;
    if (i < 576 && ci < CI)
      gw[((int64_t)co * CI + ci) * 9 + tap] = previous[e] + ((sums[0][tap][ci_local] + sums[1][tap][ci_local]) + (sums[2][tap][ci_local] + sums[3][tap][ci_local]));
  }
}

float* partial_workspace(size_t bytes, hipStream_t stream);

static int check_dtype(int dtype) {
  SRGAN_REQUIRE(dtype == 1 || dtype == 2, SRGAN_EINVAL, "blocked 16-bit tensors are bf16 (1) or fp16 (2)");
  return SRGAN_OK;
}
static int check_dtype_or_f32(int dtype) {        // the layout conversions and sums also take dtype 0: fp32, four channels per slot
  SRGAN_REQUIRE(dtype >= 0 && dtype <= 2, SRGAN_EINVAL, "blocked tensors are fp32 (0), bf16 (1) or fp16 (2)");
  return SRGAN_OK;
}

}  // namespace srgan

using namespace srgan;

extern "C" int64_t srgan_h_k4s2_weight_slots(int32_t A, int32_t B, int direction, int dtype);     // (blocked16_k4s2.hip)

namespace srgan {
// the job (= the single-layer kernel's arguments) of one conv-weight operand; transposed: rows = C, taps mirrored
static void conv_weights_job(HPackJob& job, const float* w, void* packed, int32_t K, int32_t C, int32_t R, int32_t S, int transposed,
                             int dtype) {
  memset(&job, 0, sizeof(job));
  const int rows = transposed ? C : K, reduced = transposed ? K : C;
  job.w = w; job.packed = (Slot*)packed; job.kind = 0; job.prec = dtype;
  job.slots = (int64_t)((reduced + 15) / 16) * R * S * 2 * rows;
  job.p[0] = rows; job.p[1] = reduced; job.p[2] = R; job.p[3] = S;
  job.p[4] = transposed ? R * S - 1 : 0;                    // base
  job.p[5] = transposed ? R * S : C * R * S;                // row stride
  job.p[6] = transposed ? C * R * S : R * S;                // reduced-channel stride
  job.p[7] = transposed ? -S : S; job.p[8] = transposed ? -1 : 1;
}
}  // namespace srgan

extern "C" {

int srgan_h_pack(const float* x, void* out, const void* mask_ref, float slope, int32_t N, int32_t C, int64_t HW, int dtype,
                 hipStream_t stream) {
  if (const int status = check_dtype_or_f32(dtype)) return status;
  SRGAN_REQUIRE(x && out && N >= 0 && C > 0 && HW > 0 && HW < ((int64_t)1 << 31), SRGAN_EINVAL, "srgan_h_pack arguments");
  const int group = dtype == 0 ? 4 : 8, CG = (C + group - 1) / group;
  const int64_t slots = (int64_t)N * CG * HW;
  if (slots == 0) return SRGAN_OK;
  const dim3 grid(stream_grid(slots, 256));
  if (dtype == 0) hipLaunchKernelGGL(h_pack_kernel<0>, grid, dim3(256), 0, stream, x, (Slot*)out, (const Slot*)mask_ref, slope, slots, C, CG, (int32_t)HW);
  else if (dtype == 1) hipLaunchKernelGGL(h_pack_kernel<1>, grid, dim3(256), 0, stream, x, (Slot*)out, (const Slot*)mask_ref, slope, slots, C, CG, (int32_t)HW);
  else hipLaunchKernelGGL(h_pack_kernel<2>, grid, dim3(256), 0, stream, x, (Slot*)out, (const Slot*)mask_ref, slope, slots, C, CG, (int32_t)HW);
  return launch_status();
}

int srgan_h_unpack(const void* x, float* out, int32_t N, int32_t C, int64_t HW, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_or_f32(dtype)) return status;
  SRGAN_REQUIRE(x && out && N >= 0 && C > 0 && HW > 0 && HW < ((int64_t)1 << 31), SRGAN_EINVAL, "srgan_h_unpack arguments");
  const int group = dtype == 0 ? 4 : 8, CG = (C + group - 1) / group;
  const int64_t slots = (int64_t)N * CG * HW;
  if (slots == 0) return SRGAN_OK;
  const dim3 grid(stream_grid(slots, 256));
  if (dtype == 0) hipLaunchKernelGGL(h_unpack_kernel<0>, grid, dim3(256), 0, stream, (const Slot*)x, out, slots, C, CG, (int32_t)HW);
  else if (dtype == 1) hipLaunchKernelGGL(h_unpack_kernel<1>, grid, dim3(256), 0, stream, (const Slot*)x, out, slots, C, CG, (int32_t)HW);
  else hipLaunchKernelGGL(h_unpack_kernel<2>, grid, dim3(256), 0, stream, (const Slot*)x, out, slots, C, CG, (int32_t)HW);
  return launch_status();
}

int srgan_h_add(const void* a, const void* b, void* out, int64_t slots, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_or_f32(dtype)) return status;
  SRGAN_REQUIRE(a && b && out && slots >= 0, SRGAN_EINVAL, "srgan_h_add arguments");
  if (slots == 0) return SRGAN_OK;
  const dim3 grid(stream_grid(slots, 256));
  if (dtype == 0) hipLaunchKernelGGL(h_add_kernel<0>, grid, dim3(256), 0, stream, (const Slot*)a, (const Slot*)b, (Slot*)out, slots);
  else if (dtype == 1) hipLaunchKernelGGL(h_add_kernel<1>, grid, dim3(256), 0, stream, (const Slot*)a, (const Slot*)b, (Slot*)out, slots);
  else hipLaunchKernelGGL(h_add_kernel<2>, grid, dim3(256), 0, stream, (const Slot*)a, (const Slot*)b, (Slot*)out, slots);
  return launch_status();
}

// out[c] += sum over n and pixels of x[n, c, pixel]   (c < C; fp32; the bias gradient of a fused convolution / linear layer)
int srgan_h_channel_sums(const void* x, float* out, int32_t N, int32_t C, int64_t HW, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_or_f32(dtype)) return status;
  SRGAN_REQUIRE(x && out && N > 0 && C > 0 && HW > 0 && HW < ((int64_t)1 << 31), SRGAN_EINVAL, "srgan_h_channel_sums arguments");
  const int group = dtype == 0 ? 4 : 8, CG = (C + group - 1) / group;
  const int64_t per_group = (int64_t)N * HW;
  int parts = (int)((per_group + 4095) / 4096);
  if (parts > 256) parts = 256;
  if (parts < 1) parts = 1;
  while (parts > 1 && (int64_t)parts * CG > 4096) parts >>= 1;
  unsigned int* tickets = nullptr;
  static const bool two_launches = getenv("SRGAN_H_SUMS_TWO_LAUNCHES") != nullptr;
  float* part = two_launches ? nullptr : row_finish_workspace(CG, parts, 8, g_h_sum_tickets, stream, &tickets);
  if (!part) {
    tickets = nullptr;
    part = partial_workspace((size_t)CG * parts * 8 * sizeof(float), stream);
  }
  SRGAN_REQUIRE(part, SRGAN_EINVAL, "srgan_h_channel_sums: register a workspace for this stream first (srgan_set_workspace)");
  const dim3 grid((unsigned)CG, (unsigned)parts);
  if (dtype == 0) hipLaunchKernelGGL(h_channel_sums_kernel<0>, grid, dim3(256), 0, stream, (const Slot*)x, part, N, CG, (int32_t)HW, parts, tickets, out, C);
  else if (dtype == 1) hipLaunchKernelGGL(h_channel_sums_kernel<1>, grid, dim3(256), 0, stream, (const Slot*)x, part, N, CG, (int32_t)HW, parts, tickets, out, C);
  else hipLaunchKernelGGL(h_channel_sums_kernel<2>, grid, dim3(256), 0, stream, (const Slot*)x, part, N, CG, (int32_t)HW, parts, tickets, out, C);
  if (!tickets) hipLaunchKernelGGL(h_channel_sums_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, out, C, parts, group);
  return launch_status();
}

// mode 0: out = maxpool2x2(x); mode 2: out = `g` gathered at the arg-max of x (both [planes][H/2][W/2] slots, planes = N * groups)
int srgan_h_maxpool2(const void* x, const void* g, void* out, int64_t planes, int32_t H, int32_t W, int mode, int dtype,
                     hipStream_t stream) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(x && out && planes >= 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && (mode == 0 || (mode == 2 && g)),
                SRGAN_EINVAL, "srgan_h_maxpool2 arguments (even planes; mode 0 or 2)");
  const int64_t total = planes * (H / 2) * (W / 2);
  if (total == 0) return SRGAN_OK;
  const dim3 grid(stream_grid(total, 256));
  if (dtype == 1) {
    if (mode == 0) hipLaunchKernelGGL((h_maxpool_kernel<1, 0>), grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)g, (Slot*)out, planes, H, W);
    else hipLaunchKernelGGL((h_maxpool_kernel<1, 2>), grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)g, (Slot*)out, planes, H, W);
  } else {
    if (mode == 0) hipLaunchKernelGGL((h_maxpool_kernel<2, 0>), grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)g, (Slot*)out, planes, H, W);
    else hipLaunchKernelGGL((h_maxpool_kernel<2, 2>), grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)g, (Slot*)out, planes, H, W);
  }
  return launch_status();
}

// gx (shape of x) = gp placed at the arg-max of x, times mask(x, slope) when `masked`
int srgan_h_maxpool2_bwd(const void* x, const void* gp, void* gx, int64_t planes, int32_t H, int32_t W, int masked, float slope,
                         int dtype, hipStream_t stream) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(x && gp && gx && planes >= 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SRGAN_EINVAL,
                "srgan_h_maxpool2_bwd arguments");
  const int64_t total = planes * (H / 2) * (W / 2);
  if (total == 0) return SRGAN_OK;
  const dim3 grid(stream_grid(total, 256));
  if (dtype == 1) hipLaunchKernelGGL(h_maxpool_bwd_kernel<1>, grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)gp, (Slot*)gx, planes, H, W, masked, slope);
  else hipLaunchKernelGGL(h_maxpool_bwd_kernel<2>, grid, dim3(256), 0, stream, (const Slot*)x, (const Slot*)gp, (Slot*)gx, planes, H, W, masked, slope);
  return launch_status();
}

// Slots of a packed convolution weight tensor of CO rows, CI reduced channels and R x S taps.
int64_t srgan_h_conv_weight_slots(int32_t CO, int32_t CI, int32_t R, int32_t S) {
  return (int64_t)((CI + 15) / 16) * R * S * 2 * CO;
}

// transposed = 0: the forward operand of conv2d weights w[K][C][R][S] (rows = K, reduced = C).
// transposed = 1: the data-gradient operand (rows = C, reduced = K, taps mirrored).
int srgan_h_pack_conv_weights(const float* w, void* packed, int32_t K, int32_t C, int32_t R, int32_t S, int transposed, int dtype,
                              hipStream_t stream) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(w && packed && K > 0 && C > 0 && R > 0 && S > 0, SRGAN_EINVAL, "srgan_h_pack_conv_weights arguments");
  HPackJob job;
  conv_weights_job(job, w, packed, K, C, R, S, transposed, dtype);
  const dim3 grid((unsigned)((job.slots + 255) / 256));
  const int32_t* q = job.p;
  if (dtype == 1) hipLaunchKernelGGL(h_pack_conv_weights_kernel<1>, grid, dim3(256), 0, stream, w, (Slot*)packed, job.slots, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
  else hipLaunchKernelGGL(h_pack_conv_weights_kernel<2>, grid, dim3(256), 0, stream, w, (Slot*)packed, job.slots, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
  return launch_status();
}

// ---- the batched form: the caller keeps a device array of jobs (srgan_h_pack_job_bytes() each), filled on the host by the two
// functions below with the arguments of the single-layer calls, and launches them all at once.  A job function writes its jobs
// at `jobs` (host memory) with workgroups from `first_block` on and returns the number of workgroups they take (< 0: error).
int32_t srgan_h_pack_job_bytes(void) { return (int32_t)sizeof(HPackJob); }

int64_t srgan_h_pack_job_conv_weights(void* jobs, int64_t first_block, const float* w, void* packed, int32_t K, int32_t C, int32_t R,
                                      int32_t S, int transposed, int dtype) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(jobs && first_block >= 0 && w && packed && K > 0 && C > 0 && R > 0 && S > 0, SRGAN_EINVAL,
                "srgan_h_pack_job_conv_weights arguments");
  HPackJob job;
  conv_weights_job(job, w, packed, K, C, R, S, transposed, dtype);
  job.first_block = first_block;
  memcpy(jobs, &job, sizeof(job));
  return (job.slots + 255) / 256;
}

int64_t srgan_h_pack_job_k4s2_weights(void* jobs, int64_t first_block, const float* w, void* packed, int32_t A, int32_t B,
                                      int direction, int dtype, int32_t* jobs_written) {
  SRGAN_REQUIRE(dtype >= 0 && dtype <= 2, SRGAN_EINVAL, "srgan_h_pack_job_k4s2_weights dtype");
  SRGAN_REQUIRE(jobs && first_block >= 0 && w && packed && A > 0 && B > 0 && (direction == 0 || direction == 1) && jobs_written,
                SRGAN_EINVAL, "srgan_h_pack_job_k4s2_weights arguments");
  const int64_t slots = srgan_h_k4s2_weight_slots(A, B, direction, dtype) / (direction ? 4 : 1);
  const int64_t blocks = (slots + 255) / 256;
  const int count = direction ? 4 : 1;
  for (int cls = 0; cls < count; ++cls) {
    HPackJob job;
    memset(&job, 0, sizeof(job));
    job.w = w; job.packed = (Slot*)packed + cls * slots; job.slots = slots; job.first_block = first_block + cls * blocks;
    job.kind = 1; job.prec = dtype;
    // the body's (rows, reduced, row_stride, reduced_stride, mode): "down" reads w[A][B] as rows A, "up" as rows B
    job.p[0] = direction ? B : A; job.p[1] = direction ? A : B;
    job.p[2] = direction ? 16 : B * 16; job.p[3] = direction ? B * 16 : 16; job.p[4] = direction ? 1 + cls : 0;
    memcpy((char*)jobs + cls * sizeof(job), &job, sizeof(job));
  }
  *jobs_written = count;
  return count * blocks;
}

int srgan_h_pack_batched(const void* jobs_device, int32_t count, int64_t blocks, hipStream_t stream) {
  SRGAN_REQUIRE(jobs_device && count > 0 && blocks > 0 && blocks < ((int64_t)1 << 31), SRGAN_EINVAL, "srgan_h_pack_batched arguments");
  hipLaunchKernelGGL(h_pack_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const HPackJob*)jobs_device, count);
  return launch_status();
}

// out[N, C_out, H, W] = epi(conv3x3 / stride 1 / pad 1 of x[N, C_in, H, W] with packed weights of `rows` rows [+ bias]).
// epi 0: plain; 1: + bias (may be NULL), then leaky_relu(slope) (slope 0 = relu, 1 = identity); 2: times mask(ref, slope), ref
// of the output's shape.  C_out = the channels the output tensor stores (rows of the packed weights may be fewer or more).
int srgan_h_conv3x3(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi, void* out,
                    int32_t N, int32_t C_in, int32_t C_out, int32_t rows, int32_t H, int32_t W, int dtype, hipStream_t stream) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(x && packed && out && N > 0 && C_in > 0 && C_out > 0 && rows > 0 && H > 0 && W > 0 && epi >= 0 && epi <= 2 &&
                (epi != 2 || ref), SRGAN_EINVAL, "srgan_h_conv3x3 arguments");
  HConv3Params p;
  p.in = (const Slot*)x; p.wp = (const Slot*)packed; p.out = (Slot*)out; p.bias = bias; p.ref = (const Slot*)ref;
  p.slope = slope; p.epi = epi;
  p.N = N; p.CGI = (C_in + 7) / 8; p.CGO = (C_out + 7) / 8; p.CO = rows; p.C_real = C_out; p.H = H; p.W = W;
  p.chunks = (p.CGI + 1) / 2;
  SRGAN_REQUIRE((int64_t)N * (p.CGI > p.CGO ? p.CGI : p.CGO) * H * W < ((int64_t)1 << 31), SRGAN_ERANGE, "srgan_h_conv3x3 tensor size");
  HConv3Plan plan;
  const int rows_needed = p.CGO * 8 < rows ? p.CGO * 8 : rows;       // output rows that are stored
  SRGAN_REQUIRE(hconv3_plan(N, p.CGI, rows_needed, H, W, plan), SRGAN_EUNSUPPORTED, "srgan_h_conv3x3 plane size");
  p.tiles_x = plan.tiles_x; p.tiles_y = plan.tiles_y; p.tiles_m = plan.tiles_m;
  p.chunks_per_split = plan.chunks_per;
  p.split_ws = nullptr; p.split_tickets = nullptr;
  int split = plan.split;
  if (split > 1) {
    int ticket_set = -1;
    const int64_t accumulators = (int64_t)(plan.bm / 32) * plan.ni * 16 * 256;
    float* ws = split_workspace(plan.blocks, split, accumulators, 0, stream, &ticket_set);
    unsigned int* tickets = ws ? device_tickets(g_hconv3_split_tickets) : nullptr;
    if (ws && tickets) {
      p.split_ws = ws;
      p.split_tickets = tickets + (size_t)ticket_set * SPLIT_TICKET_TILES;
    } else {
      split = 1;
      p.chunks_per_split = p.chunks;
    }
  }
  p.xcd_remap = (plan.blocks % 8 == 0 && plan.blocks >= 64) ? 1 : 0;
  static const int experiment = getenv("SRGAN_H_EXPERIMENT") ? atoi(getenv("SRGAN_H_EXPERIMENT")) : 0;
  p.debug = experiment;
  const dim3 grid((unsigned)plan.blocks, (unsigned)split, 1);
  // LDS-DMA ring for 64-row tiles with an unsplit K.  SRGAN_H_DMA_RING: 0 = off (register staging), 2 / 3 = that many stages;
  // default two stages = two workgroups per CU.  Measured (profiles/r06h_*, scratch/h_conv_bench.py): three stages -- two chunks
  // in flight but ONE workgroup per CU -- lose 25-40 %; two stages equal the register-staged kernel on the wide planes and gain
  // 15-20 % on the 16 x 16 / 8 x 8 planes of the stacked pass.  With the staging switched off altogether the loop reaches
  // 950-1160 TF/s: fragment reads and matrix work, not the operand stream, are the larger part of what is left.
  static const int ring_env = getenv("SRGAN_H_DMA_RING") ? atoi(getenv("SRGAN_H_DMA_RING")) : -1;
  const int ring = ring_env >= 0 ? ring_env : 2;
  const Slot* zero = (ring == 2 || ring == 3) && split == 1 && plan.bm == 64
                         ? h_zero_slots() : nullptr;
  const int slot = profile_bracket_begin(stream);
  if (zero) {
    int launched;
    if (ring == 3) launched = dtype == 1 ? hconv3_dma_launch<1, 3>(p, plan, grid, stream, zero) : hconv3_dma_launch<2, 3>(p, plan, grid, stream, zero);
    else launched = dtype == 1 ? hconv3_dma_launch<1, 2>(p, plan, grid, stream, zero) : hconv3_dma_launch<2, 2>(p, plan, grid, stream, zero);
    if (launched != SRGAN_OK) return launched;
  }
  else if (dtype == 1) hconv3_launch<1>(p, plan, grid, stream);
  else hconv3_launch<2>(p, plan, grid, stream);
  const int status = launch_status();
  const double pixels = (double)N * H * W;
  profile_bracket_end_bytes(slot, stream, rows_needed, (int64_t)pixels, (int64_t)p.CGI * 8 * 9, 14, plan.bm, plan.ni * 128, split,
                            2.0 * (pixels * p.CGI * 8 + pixels * p.CGO * 8 * (epi == 2 ? 2 : 1) + (double)rows * p.CGI * 8 * 9), dtype);
  return status;
}

// gw[C_out][C_in][3][3] (fp32) += the weight gradient of a 3x3 / s1 / p1 convolution from x[N, C_in, H, W] and gy[N, C_out, H, W]
int srgan_h_conv3x3_wgrad(const void* x, const void* gy, float* gw, int32_t N, int32_t C_in, int32_t C_out, int32_t H, int32_t W,
                          int dtype, hipStream_t stream) {
  if (const int status = check_dtype(dtype)) return status;
  SRGAN_REQUIRE(x && gy && gw && N > 0 && C_in > 0 && C_out > 0 && H > 0 && W > 0, SRGAN_EINVAL, "srgan_h_conv3x3_wgrad arguments");
  HWgrad3Params p;
  p.x = (const Slot*)x; p.gy = (const Slot*)gy;
  p.N = N; p.CGX = (C_in + 7) / 8; p.CGY = (C_out + 7) / 8; p.H = H; p.W = W;
  int tw, rows;
  if (W > 16) { tw = 32; rows = 2; }
  else if (W == 8 && H == 8) { tw = 8; rows = 8; }
  else if (W == 4 && H == 4) { tw = 4; rows = 4; }
  else { tw = 16; rows = 4; }
  const int img = HWGRAD_P / (rows * tw);
  p.tiles_x = (W + tw - 1) / tw;
  p.tiles_y = img > 1 ? 1 : (H + rows - 1) / rows;
  p.tiles_n = (N + img - 1) / img;
  p.pixel_tiles = p.tiles_x * p.tiles_y * p.tiles_n;
  p.tiles_ci = (C_in + 63) / 64;
  static const bool no_tall = getenv("SRGAN_H_WGRAD_64") != nullptr;
  const int mb = (C_out >= 128 && !no_tall) ? 4 : 2;                 // 128- or 64-row blocks of gw
  const int tiles_co = (C_out + 32 * mb - 1) / (32 * mb);
  const int blocks = p.tiles_ci * tiles_co;
  // Walkers per block: enough workgroups to occupy the chip, but every walker leaves its accumulators (147 / 295 KB) as a partial
  // block that the finish reads back -- 1024 / blocks walkers made that traffic (and the finish's serial walker loop) the larger
  // part of the launch on both ends of VGG (1 block x 1024 walkers; 64 blocks x 16 walkers: 151 MB each, profiles/r06b_*).
  // (64-row blocks run two workgroups per CU: where a walker still gets a long run of tiles -- the 64-channel layers on 64 x 64
  // planes: one block, 8192+ tiles -- all 512 slots are filled; elsewhere 320, for the partial traffic)
  static const char* forced_walkers = getenv("SRGAN_H_WGRAD_WALKERS");
  int target = mb == 4 ? 256 : ((int64_t)p.pixel_tiles >= (int64_t)16 * 512 * blocks ? 512 : 320);
  if (forced_walkers) target = atoi(forced_walkers);
  int walkers = (target + blocks - 1) / blocks;
  if (walkers > p.pixel_tiles) walkers = p.pixel_tiles;
  if (walkers < 1) walkers = 1;
  p.walkers = walkers;
  const int threads = 128 * mb;
  p.partial = partial_workspace((size_t)blocks * walkers * 9 * 16 * threads * sizeof(float), stream);
  SRGAN_REQUIRE(p.partial, SRGAN_EINVAL, "srgan_h_conv3x3_wgrad: register a workspace for this stream first (srgan_set_workspace)");
  const dim3 grid((unsigned)blocks, (unsigned)walkers);
  const int slot = profile_bracket_begin(stream);
#define HWGRAD_LAUNCH(TWv, ROWSv)                                                                                       \
  do {                                                                                                                  \
    if (mb == 4) {                                                                                                      \
      if (dtype == 1) hipLaunchKernelGGL((hwgrad3x3_kernel<TWv, ROWSv, 1, 4>), grid, dim3(512), 0, stream, p);         \
      else hipLaunchKernelGGL((hwgrad3x3_kernel<TWv, ROWSv, 2, 4>), grid, dim3(512), 0, stream, p);                    \
    } else {                                                                                                            \
      if (dtype == 1) hipLaunchKernelGGL((hwgrad3x3_kernel<TWv, ROWSv, 1, 2>), grid, dim3(256), 0, stream, p);         \
      else hipLaunchKernelGGL((hwgrad3x3_kernel<TWv, ROWSv, 2, 2>), grid, dim3(256), 0, stream, p);                    \
    }                                                                                                                   \
  } while (0)
  if (tw == 32) HWGRAD_LAUNCH(32, 2);
  else if (tw == 16) HWGRAD_LAUNCH(16, 4);
  else if (tw == 8) HWGRAD_LAUNCH(8, 8);
  else HWGRAD_LAUNCH(4, 4);
#undef HWGRAD_LAUNCH
  const int64_t elements = (int64_t)C_out * C_in * 9;
  hipLaunchKernelGGL(hwgrad3x3_finish_kernel, dim3((unsigned)(blocks * 32 * mb)), dim3(256), 0, stream, p.partial, gw,
                     C_out, C_in, p.tiles_ci, walkers, 32 * mb);
  const int status = launch_status();
  const double pixels = (double)N * H * W;
  profile_bracket_end_bytes(slot, stream, C_out, (int64_t)C_in * 9, (int64_t)pixels, 15, 32 * mb, 64, walkers,
                            2.0 * pixels * (p.CGX + p.CGY) * 8 + 8.0 * (double)elements, dtype);
  return status;
}

}  // extern "C"
