// conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution with an LDS-resident input halo tile.
//
// The DenseNet growth convolutions (128 -> 32, reference crowd/models.py:344-345), their data gradients
// (32 -> 128, same kernel with flipped taps) and every VGG convolution (age/vgg.py:78) are 3x3/s1/p1.  In the
// generic gather-GEMM each B element is gathered once per tap with its own address decode; at 32 output channels
// that gather, not the matrix pipe, is the bottleneck.  Here a workgroup stages, per chunk of CI_T input channels,
//   * the input patch  [CI_T][TH + 2][TW + 2]  (each input element read from HBM/L2 ONCE for all 9 taps), and
//   * the weight slice [CI_T * 9][BM]
// into LDS with coalesced loads whose per-thread offsets are computed once per workgroup, and the inner loop is
// nothing but ds_read_b32 at per-lane base + compile-time immediate offsets feeding v_mfma_f32_32x32x2_f32:
// the two k-values of one MFMA are the SAME tap of two consecutive input channels, so the lane-half (k parity)
// contributes a constant LDS offset.  MFMA rows = output channels, columns = 32 consecutive pixels of one image row
// (lanes -> pixels: conflict-free LDS reads, 128-byte global stores).  The next chunk's global loads are issued
// before the current chunk's 9 * CI_T / 2 * MI * NI MFMAs, so HBM latency hides under the matrix pipe.
// Roofline: fp32 MFMA (157.3 TF/s); algorithmic work 2 * 9 * CI * CO FLOP per output pixel.
#include "common.h"
#include "split_finish.h"
#include <stdlib.h>
#include <type_traits>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Conv3Params {
  const float* in;      // [N, CI, H, W] (batch stride in_bs)
  const float* w;       // element (o, i, kh, kw) at w[w_base + o*w_so + i*w_si + kh*w_skh + kw*w_skw]
  float* out;           // [N, CO, H, W] (batch stride out_bs)
  const float* bias;    // per output channel, optional
  int32_t N, CI, CO, H, W;
  int64_t in_bs, out_bs;
  int32_t w_so, w_si, w_skh, w_skw, w_base;
  int32_t tiles_x, tiles_y, tiles_m;
  int32_t ci_per_split;
  int32_t mode;         // 0 store, 1 accumulate, 2 atomic; K split with the ORDERED finish (split_finish.h): 3 store, 4 accumulate
  float* split_ws;      // modes 3 / 4: [tiles][splits][accumulators of a workgroup] partials in the caller's workspace
  unsigned int* split_tickets;   // one ticket per output tile (device global, this stream's row)
  // PRO: the input is relu(batch_norm_eval(in)) computed on the fly (per input channel; NULL otherwise)
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
  // EPI: the output goes through the backward of relu(batch_norm_eval(epi_x)) on its way out (bn_* then describe the
  // OUTPUT channels, see BnBackwardEpilogue); epi_partial[q][workgroup tile][CO] receives the parameter-gradient sums
  const float* epi_x; int64_t epi_x_bs;
  float* epi_partial; int32_t epi_tiles;
  const void* w_packed;   // mixed precision: the weights repacked by conv3x3_pack_weights_kernel (16-byte operand slots)
  // Output placement: element (o, y, x) of the H x W grid the kernel walks goes to out[o * out_plane + y * out_sy +
  // x * out_sx + out_off] (a plain convolution: H * W, W, 1, 0; a stride-2 class of a transposed convolution writes every
  // other pixel of a 2H x 2W plane) and `taps` is the 9-bit mask of the window positions (kh * 3 + kw) that exist.
  int32_t out_plane, out_sy, out_sx, out_off;
  int32_t taps;
  // XCD-aware order: hardware workgroup b runs on XCD b % 8, so logical tile = (b % 8) * (grid / 8) + b / 8 gives every
  // XCD one contiguous band of tiles -- neighbouring tiles (which share halo rows / columns and the 128-byte lines of a
  // 32-pixel tile row) then read them through ONE L2 instead of two (round 2 PMC: 1.9x the algorithmic bytes).
  int32_t xcd_remap;
};

constexpr int CONV3_PRO_MAX_CI = 512;

__device__ unsigned int g_conv3_split_tickets[SPLIT_TICKET_SETS * SPLIT_TICKET_TILES];

// A stride-2 class of a k4 / s2 / p1 transposed convolution as a 2x2 sub-window of the 3x3 kernel (see conv3x3_run).
struct Conv3Placement { int32_t taps, out_plane, out_sy, out_sx, out_off; };   // channels of one workgroup's K range whose (a, b) fit the LDS table

// PRO = frozen batch-norm + ReLU fused into the patch staging (reference crowd/models.py:342-345: norm2, relu2,
// conv2): the (a, b) of the workgroup's input channels sit in a small LDS table and every patch element goes through
// max(fma(x, a, b), 0) when it is written to LDS; padding stays exactly 0 (the reference pads the activated tensor).
// TW = tile width: 32, or 16 for 16-pixel-wide images, where a 32-lane column block is two image rows of 16 pixels
// (a 32-wide tile would leave half of every MFMA's columns outside the image).
// EPI = the backward of a frozen batch-norm + ReLU fused into the epilogue of the data gradient (reference
// crowd/models.py:342-345 backwards: conv2 -> relu2 -> norm2): out = acc * [fma(x, a, b) > 0] * a, plus the workgroup's
// two parameter-gradient row sums (store mode only, no split over input channels).
// TAPS: compile-time mask of the window positions the inner loop multiplies (0x1FF = the full 3x3 window; the four
// stride-2 classes of a k4 / s2 / p1 transposed convolution are 2x2 sub-windows: 0x01B, 0x036, 0x0D8, 0x1B0).
template <int BM, int TH, int CI_T, bool PRO, int TW, bool EPI = false, int TAPS = 0x1FF>
__global__ __launch_bounds__(256, 2) void conv3x3_lds_kernel(const Conv3Params p) {
  static_assert(!EPI || !PRO, "the batch-norm backward epilogue pairs with the plain kernel");
  static_assert(TAPS == 0x1FF || (!EPI && !PRO), "tap subsets pair with the plain kernel");
  constexpr int RPB = 32 / TW;                    // image rows per 32-lane column block
  constexpr int ROWS = TH * RPB;                  // image rows of the workgroup's tile
  constexpr int PH = ROWS + 2, PW = TW + 2, PHPW = PH * PW;
  constexpr int MI = BM / 32, NI = TH / 4;        // each of the 4 waves owns NI column blocks
  constexpr int LDW = BM + 1;
  constexpr int PATCH = CI_T * PHPW, WTS = CI_T * 9 * BM;
  constexpr int NP = (PATCH + 255) / 256, NW = (WTS + 255) / 256;
  static_assert(CI_T % 2 == 0 && NI >= 1 && MI >= 1, "bad tile");
  __shared__ float lds[PATCH + CI_T * 9 * LDW];
  __shared__ float2 coef[PRO ? CONV3_PRO_MAX_CI : 1];
  float* patch = lds;
  float* wt = lds + PATCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  int block = blockIdx.x;
  if (p.xcd_remap) block = (block & 7) * ((int)gridDim.x >> 3) + (block >> 3);
  const int logical_block = block;
  const int tm = block % p.tiles_m; block /= p.tiles_m;
  const int tx = block % p.tiles_x; block /= p.tiles_x;
  const int ty = block % p.tiles_y;
  const int n = block / p.tiles_y;
  const int m0 = tm * BM, y0 = ty * ROWS, x0 = tx * TW;
  const int cbeg = (int)blockIdx.y * p.ci_per_split;
  const int cend = min(p.CI, cbeg + p.ci_per_split);
  const int HW = p.H * p.W;

  // Per-thread staging offsets (relative to the chunk's first channel), computed once.
  int poff[NP], woff[NW];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const int flat = e * 256 + tid;
    const int c = flat / PHPW, rem = flat - c * PHPW;
    const int py = rem / PW, px = rem - py * PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = flat < PATCH && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    poff[e] = ok ? c * HW + y * p.W + x : -1;
  }
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int flat = e * 256 + tid;
    const int o = flat / (CI_T * 9), r = flat - o * (CI_T * 9);     // r = ci_local * 9 + tap : lanes walk r
    const int ci = r / 9, tap = r - ci * 9;
    const int kh = tap / 3, kw = tap - kh * 3;
    const bool ok = flat < WTS && (m0 + o) < p.CO && ((p.taps >> tap) & 1);     // (a missing tap stages zeros)
    woff[e] = ok ? p.w_base + (m0 + o) * p.w_so + ci * p.w_si + kh * p.w_skh + kw * p.w_skw : -1;
  }
  const float* in_n = p.in + (int64_t)n * p.in_bs;

  // Raw loads only in fetch(); validity is applied when the registers are written to LDS (any use of a loaded value
  // before the MFMA loop would make the compiler wait for the loads there instead of overlapping them).
  float rp[NP], rw[NW];
  auto fetch = [&](int c0) {
    const int room = cend - c0;                         // channels of this chunk that exist
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int flat = e * 256 + tid;
      const bool ok = poff[e] >= 0 && flat / PHPW < room;
      rp[e] = in_n[(int64_t)c0 * HW + (ok ? poff[e] : 0)];
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int flat = e * 256 + tid;
      const int ci = (flat % (CI_T * 9)) / 9;
      const bool ok = woff[e] >= 0 && ci < room;
      rw[e] = p.w[ok ? woff[e] + c0 * p.w_si : 0];
    }
  };
  auto stage = [&](int c0) {
    const int room = cend - c0;
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int flat = e * 256 + tid;
      const bool ok = poff[e] >= 0 && flat / PHPW < room;
      float v = rp[e];
      if (PRO) {
        const float2 cf = coef[min(c0 - cbeg + flat / PHPW, CONV3_PRO_MAX_CI - 1)];
        v = fmaxf(fmaf(v, cf.x, cf.y), 0.f);
      }
      if (flat < PATCH) patch[flat] = ok ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int flat = e * 256 + tid;
      const int o = flat / (CI_T * 9), r = flat - o * (CI_T * 9);
      const bool ok = woff[e] >= 0 && r / 9 < room;
      if (flat < WTS) wt[r * LDW + o] = ok ? rw[e] : 0.f;
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // Per-lane LDS bases: the lane half selects the odd channel of a pair (constant offset), lanes 0..31 walk pixels
  // (B) or output channels (A); everything else below is a compile-time immediate.
  const float* a_base = wt + lhi * 9 * LDW + l31;
  const float* b_base = patch + lhi * PHPW + ((wave * NI) * RPB + l31 / TW) * PW + l31 % TW;

  if (cbeg < cend) {
    fetch(cbeg);
    if (PRO) {
      for (int c = tid; c < cend - cbeg; c += 256) {
        float a, b;
        bn_coefficients(p.bn_mean[cbeg + c], p.bn_inv[cbeg + c], p.bn_gamma[cbeg + c], p.bn_beta[cbeg + c], a, b);
        coef[c] = make_float2(a, b);
      }
      __syncthreads();
    }
    stage(cbeg);
    __syncthreads();
    for (int c0 = cbeg; c0 < cend; c0 += CI_T) {
      const bool more = c0 + CI_T < cend;
      if (more) fetch(c0 + CI_T);
      // (the channel-pair loop is kept rolled: full unrolling makes the scheduler hoist hundreds of LDS reads and
      // spill; the 9 taps x MI x NI MFMAs inside are plenty of straight-line work)
#pragma unroll 1
      for (int cp = 0; cp < CI_T / 2; ++cp) {
        const float* a_cp = a_base + cp * (2 * 9 * LDW);
        const float* b_cp = b_base + cp * (2 * PHPW);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (!((TAPS >> tap) & 1)) continue;
          const int kh = tap / 3, kw = tap % 3;
          float a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = a_cp[tap * LDW + mi * 32];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b[ni] = b_cp[(ni * RPB + kh) * PW + kw];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      }
      __syncthreads();
      if (more) {
        stage(c0 + CI_T);
        __syncthreads();
      }
    }
  }

  const int x = x0 + l31 % TW;
  float* out_n = p.out + (int64_t)n * p.out_bs;
  if (EPI) {
    // (the K loop ends with a barrier: the LDS is free)
    float* table = lds;                            // [BM][4]: a, b, mean of output row m0 + i
    float* sums = lds + BM * 4;                    // [4 waves][2][BM]
    if (tid < BM) {
      const int o = min(m0 + tid, p.CO - 1);
      const float mu = p.bn_mean[o];
      float a, b;
      bn_coefficients(mu, p.bn_inv[o], p.bn_gamma[o], p.bn_beta[o], a, b);   // the forward's own (a, b): same mask
      table[tid * 4 + 0] = a; table[tid * 4 + 1] = b; table[tid * 4 + 2] = mu;
    }
    __syncthreads();
    const float* x_n = p.epi_x + (int64_t)n * p.epi_x_bs;
    const bool sums_wanted = p.epi_partial != nullptr;
    int pix[NI];
    bool inside[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int y = y0 + (wave * NI + ni) * RPB + l31 / TW;
      inside[ni] = y < p.H && x < p.W;
      pix[ni] = inside[ni] ? y * p.W + x : 0;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float xs[NI][16];                            // loads first, stores last (the stores may alias for all the compiler knows)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = min(m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, p.CO - 1);
          xs[ni][r] = x_n[(int64_t)o * HW + pix[ni]];
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int o = m0 + row;
        const float4 t = *reinterpret_cast<const float4*>(table + row * 4);
        float plain = 0.f, centred = 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const bool ok = inside[ni] && o < p.CO;
          const float v = (ok && fmaf(xs[ni][r], t.x, t.y) > 0.f) ? acc[mi][ni][r] : 0.f;
          if (ok) __builtin_nontemporal_store(v * t.x, out_n + (int64_t)o * HW + pix[ni]);
          plain += v;
          centred += v * (xs[ni][r] - t.z);
        }
        if (sums_wanted) {
          plain = half_wave_sum(plain);
          centred = half_wave_sum(centred);
          if (l31 == 31) {
            sums[(wave * 2 + 0) * BM + row] = plain;
            sums[(wave * 2 + 1) * BM + row] = centred;
          }
        }
      }
    }
    if (sums_wanted) {
      __syncthreads();
      if (tid < 2 * BM) {
        const int q = tid / BM, row = tid - q * BM;
        const float total = (sums[(0 * 2 + q) * BM + row] + sums[(1 * 2 + q) * BM + row]) +
                            (sums[(2 * 2 + q) * BM + row] + sums[(3 * 2 + q) * BM + row]);
        const int o = m0 + row;
        if (o < p.CO) p.epi_partial[((int64_t)q * p.epi_tiles + logical_block / p.tiles_m) * p.CO + o] = total;
      }
    }
    return;
  }
  int mode = p.mode;
  if (mode >= 3) {       // K split, ordered finish: the tile's last workgroup goes on with the sum of all slices, in slice order
    constexpr int COUNT = MI * NI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.split_ws + (int64_t)logical_block * gridDim.y * (COUNT * 256), (int)blockIdx.y,
                                          (int)gridDim.y, p.split_tickets + logical_block,
                                          [&](int i) { return acc[i / (NI * 16)][(i / 16) % NI][i % 16]; },
                                          [&](int i, float v) { acc[i / (NI * 16)][(i / 16) % NI][i % 16] = v; }))
      return;
    mode -= 3;
  }
  const bool add_bias = p.bias != nullptr && (blockIdx.y == 0 || p.mode >= 3);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int y = y0 + (wave * NI + ni) * RPB + l31 / TW;
    if (y >= p.H || x >= p.W) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // accumulate: the sixteen old values in one batch in front of the stores (loads and stores return through one in-order
      // counter, vmcnt: `*dst += v` per element made every load wait for the store in front of it)
      float previous[16];
      if (mode == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          previous[r] = o < p.CO ? out_n[(int64_t)o * p.out_plane + y * p.out_sy + x * p.out_sx + p.out_off] : 0.f;
        }
      }
      float bias_values[16];                 // (in front of the stores, like the old values)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        bias_values[r] = (add_bias && o < p.CO) ? p.bias[o] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (o >= p.CO) continue;
        const float v = acc[mi][ni][r] + bias_values[r];
        float* dst = out_n + (int64_t)o * p.out_plane + y * p.out_sy + x * p.out_sx + p.out_off;
        if (mode == 0) __builtin_nontemporal_store(v, dst);      // consumed by a later kernel, not by this one
        else if (mode == 1) *dst = previous[r] + v;
        else unsafeAtomicAdd(dst, v);
      }
    }
  }
}

// ---- mixed precision (BASELINE.json configs 2 / 5: bf16, fp16) ------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 / _f16 take 8 consecutive-k values per lane (16 B).  Here k = 16 input channels at one tap and
// the LDS tiles are stored IN THE OPERAND TYPE, grouped so that a fragment is ONE ds_read_b128:
//   patch  [2 channel groups of 8][(TH*RPB + 2) x (TW + 2) pixels]   16 B per (group, pixel)
//   weights[9 taps][2 channel groups][BM output channels]            16 B per (tap, group, output channel)
// Lanes 0-31 (pixels / output channels) read consecutive 16-byte slots and the lane half selects the channel group:
// conflict-free by construction.  Staging: a thread gathers the 8 channels of one (group, pixel) slot -- eight loads,
// each coalesced across the lanes along the image row -- rounds them (v_cvt_pk_*) and writes one 16-byte slot; the data
// in HBM stays fp32 and accumulation is fp32 (the C/D fragment, bias and store modes are those of the fp32 kernel).
// Two LDS stages: the next 16-channel chunk is converted and written while the current one is in the matrix pipe.
struct alignas(16) Half8 { uint32_t v[4]; };

template <int PREC>
__device__ __forceinline__ Half8 pack8(const float (&x)[8]) {
  Half8 out;
  if constexpr (PREC == 1) {
    bf16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (__bf16)x[j];
    out = *reinterpret_cast<Half8*>(&h);
  } else {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)x[j];
    out = *reinterpret_cast<Half8*>(&h);
  }
  return out;
}

// Weights in operand slots: packed[((chunk * 9 + tap) * 2 + group) * CO + o] = the 8 channels chunk * 16 + group * 8 ... + 7 of
// tap `tap` of output channel o, rounded to the operand type (zero beyond CI).  One thread per slot: the gather over the
// strided fp32 weight tensor happens ONCE per convolution call here (a few microseconds) instead of in every workgroup of
// the convolution, whose weight staging then is a handful of coalesced 16-byte loads per chunk.
template <int PREC>
__global__ __launch_bounds__(256) void conv3x3_pack_weights_kernel(const Conv3Params p, Half8* __restrict__ packed, int slots) {
  const int slot = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (slot >= slots) return;
  const int o = slot % p.CO, rest = slot / p.CO;
  const int g = rest & 1, ct = rest >> 1;
  const int tap = ct % 9, chunk = ct / 9;
  const int kh = tap / 3, kw = tap - kh * 3;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = chunk * 16 + g * 8 + j;
    v[j] = (c < p.CI && ((p.taps >> tap) & 1)) ? p.w[p.w_base + o * p.w_so + c * p.w_si + kh * p.w_skh + kw * p.w_skw] : 0.f;
  }
  packed[slot] = pack8<PREC>(v);
}

template <int BM, int TH, int TW, int PREC, int TAPS = 0x1FF>
__global__ __launch_bounds__(256, 2) void conv3x3_mixed_kernel(const Conv3Params p) {
  constexpr int RPB = 32 / TW, ROWS = TH * RPB, PH = ROWS + 2, PW = TW + 2, PHPW = PH * PW;
  constexpr int MI = BM / 32, NI = TH / 4;
  constexpr int PATCH_Q = 2 * PHPW, WT_Q = 9 * 2 * BM, STAGE_Q = PATCH_Q + WT_Q;       // in 16-byte slots
  constexpr int NP = (PATCH_Q + 255) / 256, NW = (WT_Q + 255) / 256;
  using frag = typename std::conditional<PREC == 1, bf16x8, f16x8>::type;
  __shared__ Half8 lds[2 * STAGE_Q];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  int block = blockIdx.x;
  if (p.xcd_remap) block = (block & 7) * ((int)gridDim.x >> 3) + (block >> 3);
  const int logical_block = block;
  const int tm = block % p.tiles_m; block /= p.tiles_m;
  const int tx = block % p.tiles_x; block /= p.tiles_x;
  const int ty = block % p.tiles_y;
  const int n = block / p.tiles_y;
  const int m0 = tm * BM, y0 = ty * ROWS, x0 = tx * TW;
  const int cbeg = (int)blockIdx.y * p.ci_per_split;
  const int cend = min(p.CI, cbeg + p.ci_per_split);
  const int HW = p.H * p.W;

  // staging slots of this thread: patch (group, pixel) and weight (tap, group, output channel)
  int poff[NP], pgrp[NP], woff[NW];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const int flat = e * 256 + tid;
    const int g = flat / PHPW, pix = flat - g * PHPW;
    const int py = pix / PW, px = pix - py * PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = flat < PATCH_Q && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    poff[e] = ok ? y * p.W + x : -1;
    pgrp[e] = g;
  }
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int flat = e * 256 + tid;
    const int o = flat % BM, tg = flat / BM;                 // lanes walk the output channels: consecutive slots in LDS
    const bool ok = flat < WT_Q && (m0 + o) < p.CO;          // and in the packed tensor ((tap * 2 + group) * CO + o)
    woff[e] = ok ? tg * p.CO + m0 + o : -1;
  }
  const float* in_n = p.in + (int64_t)n * p.in_bs;
  const Half8* packed = reinterpret_cast<const Half8*>(p.w_packed);

  float rp[NP][8];
  Half8 rw[NW];
  auto fetch = [&](int c0) {                                // raw loads only: validity is applied at stage time
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + 8 * pgrp[e] + j;
        const bool ok = poff[e] >= 0 && c < cend;
        rp[e][j] = in_n[ok ? (int64_t)c * HW + poff[e] : 0];
      }
#pragma unroll
    for (int e = 0; e < NW; ++e) rw[e] = packed[(int64_t)(c0 / 16) * (18 * p.CO) + (woff[e] >= 0 ? woff[e] : 0)];
  };
  auto stage = [&](int c0, Half8* stage_base) {
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int flat = e * 256 + tid;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (poff[e] >= 0 && c0 + 8 * pgrp[e] + j < cend) ? rp[e][j] : 0.f;
      if (flat < PATCH_Q) stage_base[flat] = pack8<PREC>(v);
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int flat = e * 256 + tid;                      // slot (tap, group, o) with o fastest: [(tap * 2 + g) * BM + o]
      Half8 v = rw[e];
      if (woff[e] < 0) v = Half8{{0u, 0u, 0u, 0u}};
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

  const int b_lane = lhi * PHPW + ((wave * NI) * RPB + l31 / TW) * PW + l31 % TW;      // + (ni * RPB + kh) * PW + kw
  const int a_lane = PATCH_Q + lhi * BM + l31;                                            // + tap * 2 * BM + mi * 32

  if (cbeg < cend) {
    fetch(cbeg);
    stage(cbeg, lds);
    __syncthreads();
    int cur = 0;
    for (int c0 = cbeg; c0 < cend; c0 += 16) {
      const bool more = c0 + 16 < cend;
      if (more) fetch(c0 + 16);
      const Half8* st = lds + cur * STAGE_Q;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (!((TAPS >> tap) & 1)) continue;
        const int kh = tap / 3, kw = tap % 3;
        frag a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag*>(&st[a_lane + tap * 2 * BM + mi * 32]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const frag*>(&st[b_lane + (ni * RPB + kh) * PW + kw]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (PREC == 1) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
          }
      }
      if (more) stage(c0 + 16, lds + (cur ^ 1) * STAGE_Q);     // the other stage: everyone left it at the previous barrier
      __syncthreads();
      cur ^= 1;
    }
  }

  const int x = x0 + l31 % TW;
  float* out_n = p.out + (int64_t)n * p.out_bs;
  int mode = p.mode;
  if (mode >= 3) {       // K split, ordered finish: the tile's last workgroup goes on with the sum of all slices, in slice order
    constexpr int COUNT = MI * NI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.split_ws + (int64_t)logical_block * gridDim.y * (COUNT * 256), (int)blockIdx.y,
                                          (int)gridDim.y, p.split_tickets + logical_block,
                                          [&](int i) { return acc[i / (NI * 16)][(i / 16) % NI][i % 16]; },
                                          [&](int i, float v) { acc[i / (NI * 16)][(i / 16) % NI][i % 16] = v; }))
      return;
    mode -= 3;
  }
  const bool add_bias = p.bias != nullptr && (blockIdx.y == 0 || p.mode >= 3);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int y = y0 + (wave * NI + ni) * RPB + l31 / TW;
    if (y >= p.H || x >= p.W) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (o >= p.CO) continue;
        float v = acc[mi][ni][r];
        if (add_bias) v += p.bias[o];
        float* dst = out_n + (int64_t)o * p.out_plane + y * p.out_sy + x * p.out_sx + p.out_off;
        if (mode == 0) __builtin_nontemporal_store(v, dst);
        else if (mode == 1) *dst += v;
        else unsafeAtomicAdd(dst, v);
      }
    }
  }
}

// Small square planes in mixed precision (P x P, P = 4 or 8: the last stages of VGG-16 on 64 x 64 faces, reference
// age/vgg.py:70-92, 512 -> 512 channels at batch 128).  A tile of one image has 16 / 64 pixels: the kernel above leaves
// 3 / 4 (1 / 2) of every MFMA's columns outside the image on such planes, and the generic gather-GEMM that took the 4 x 4
// ones ran at 80 TF/s in bf16.  Here a workgroup's 128 pixels are 128 / (P * P) WHOLE images: the patch in LDS is
// [2 channel groups of 8][IMG images][(P + 2) x (P + 2) padded plane] in operand slots, a lane's pixel q = 32 * column block
// + lane is image q / (P * P), row (q / P) % P, column q % P, and its nine taps are the same immediates as above.  Weights,
// fragments, stages and store modes are those of conv3x3_mixed_kernel.
template <int BM, int P, int PREC>
__global__ __launch_bounds__(256, 2) void conv3x3_mixed_small_kernel(const Conv3Params p) {
  constexpr int PP = P * P, IMG = 128 / PP, PW = P + 2, PHW = PW * PW;
  constexpr int MI = BM / 32;
  constexpr int PATCH_Q = 2 * IMG * PHW, WT_Q = 9 * 2 * BM, STAGE_Q = PATCH_Q + WT_Q;       // in 16-byte slots
  constexpr int NP = (PATCH_Q + 255) / 256, NW = (WT_Q + 255) / 256;
  using frag = typename std::conditional<PREC == 1, bf16x8, f16x8>::type;
  __shared__ Half8 lds[2 * STAGE_Q];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  int block = blockIdx.x;
  if (p.xcd_remap) block = (block & 7) * ((int)gridDim.x >> 3) + (block >> 3);
  const int tm = block % p.tiles_m, n0 = (block / p.tiles_m) * IMG;
  const int m0 = tm * BM;
  const int cbeg = (int)blockIdx.y * p.ci_per_split;
  const int cend = min(p.CI, cbeg + p.ci_per_split);

  // staging slots of this thread: patch (group, image, padded pixel) and weight (tap, group, output channel)
  int64_t poff[NP];
  int pgrp[NP], woff[NW];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const int flat = e * 256 + tid;
    const int g = flat / (IMG * PHW), rest = flat - g * (IMG * PHW);
    const int i = rest / PHW, pix = rest - i * PHW;
    const int y = pix / PW - 1, x = pix % PW - 1;
    const bool ok = flat < PATCH_Q && (unsigned)y < (unsigned)P && (unsigned)x < (unsigned)P && n0 + i < p.N;
    poff[e] = ok ? (int64_t)i * p.in_bs + y * P + x : -1;
    pgrp[e] = g;
  }
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int flat = e * 256 + tid;
    const int o = flat % BM, tg = flat / BM;
    const bool ok = flat < WT_Q && (m0 + o) < p.CO;
    woff[e] = ok ? tg * p.CO + m0 + o : -1;
  }
  const float* in_n = p.in + (int64_t)n0 * p.in_bs;
  const Half8* packed = reinterpret_cast<const Half8*>(p.w_packed);

  float rp[NP][8];
  Half8 rw[NW];
  auto fetch = [&](int c0) {                                // raw loads only: validity is applied at stage time
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + 8 * pgrp[e] + j;
        const bool ok = poff[e] >= 0 && c < cend;
        rp[e][j] = in_n[ok ? (int64_t)c * PP + poff[e] : 0];
      }
#pragma unroll
    for (int e = 0; e < NW; ++e) rw[e] = packed[(int64_t)(c0 / 16) * (18 * p.CO) + (woff[e] >= 0 ? woff[e] : 0)];
  };
  auto stage = [&](int c0, Half8* stage_base) {
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int flat = e * 256 + tid;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (poff[e] >= 0 && c0 + 8 * pgrp[e] + j < cend) ? rp[e][j] : 0.f;
      if (flat < PATCH_Q) stage_base[flat] = pack8<PREC>(v);
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int flat = e * 256 + tid;
      Half8 v = rw[e];
      if (woff[e] < 0) v = Half8{{0u, 0u, 0u, 0u}};
      if (flat < WT_Q) stage_base[PATCH_Q + flat] = v;
    }
  };

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  // this lane's pixel: wave = its column block
  const int q = wave * 32 + l31, qi = q / PP, qy = (q / P) % P, qx = q % P;
  const int b_lane = lhi * (IMG * PHW) + qi * PHW + qy * PW + qx;                      // + kh * PW + kw
  const int a_lane = PATCH_Q + lhi * BM + l31;                                        // + tap * 2 * BM + mi * 32

  if (cbeg < cend) {
    fetch(cbeg);
    stage(cbeg, lds);
    __syncthreads();
    int cur = 0;
    for (int c0 = cbeg; c0 < cend; c0 += 16) {
      const bool more = c0 + 16 < cend;
      if (more) fetch(c0 + 16);
      const Half8* st = lds + cur * STAGE_Q;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap % 3;
        frag a[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag*>(&st[a_lane + tap * 2 * BM + mi * 32]);
        const frag b = *reinterpret_cast<const frag*>(&st[b_lane + kh * PW + kw]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if constexpr (PREC == 1) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b, acc[mi], 0, 0, 0);
          else acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b, acc[mi], 0, 0, 0);
        }
      }
      if (more) stage(c0 + 16, lds + (cur ^ 1) * STAGE_Q);     // the other stage: everyone left it at the previous barrier
      __syncthreads();
      cur ^= 1;
    }
  }

  int mode = p.mode;
  if (mode >= 3) {       // K split, ordered finish (split_finish.h)
    constexpr int COUNT = MI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.split_ws + (int64_t)block * gridDim.y * (COUNT * 256), (int)blockIdx.y, (int)gridDim.y,
                                          p.split_tickets + block, [&](int i) { return acc[i / 16][i % 16]; },
                                          [&](int i, float v) { acc[i / 16][i % 16] = v; }))
      return;
    mode -= 3;
  }
  if (n0 + qi >= p.N) return;
  float* out_n = p.out + (int64_t)(n0 + qi) * p.out_bs;
  const bool add_bias = p.bias != nullptr && (blockIdx.y == 0 || p.mode >= 3);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (o >= p.CO) continue;
      float v = acc[mi][r];
      if (add_bias) v += p.bias[o];
      float* dst = out_n + (int64_t)o * PP + qy * P + qx;
      if (mode == 0) __builtin_nontemporal_store(v, dst);
      else if (mode == 1) *dst += v;
      else unsafeAtomicAdd(dst, v);
    }
  }
}

template <int BM, int CI_T, int TW, int TAPS>
static void launch_conv3_plain(const Conv3Params& p, int th, dim3 grid, hipStream_t stream) {
  if (th == 8) hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 8, CI_T, false, TW, false, TAPS>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 4, CI_T, false, TW, false, TAPS>), grid, dim3(256), 0, stream, p);
}

template <int BM, int CI_T, int TW>
static void launch_conv3_w(const Conv3Params& p, int th, dim3 grid, hipStream_t stream) {
  if (p.epi_x) {                       // (the plan admits the epilogue for 32- and 64-row tiles only)
    if constexpr (BM <= 64) {
      if (th == 8) hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 8, CI_T, false, TW, true>), grid, dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 4, CI_T, false, TW, true>), grid, dim3(256), 0, stream, p);
    }
  } else if (p.bn_mean) {
    if (th == 8) hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 8, CI_T, true, TW>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_lds_kernel<BM, 4, CI_T, true, TW>), grid, dim3(256), 0, stream, p);
  } else {
    switch (p.taps) {            // the full window, or one of the four 2x2 sub-windows of a stride-2 transposed convolution
      case 0x01B: launch_conv3_plain<BM, CI_T, TW, 0x01B>(p, th, grid, stream); break;
      case 0x036: launch_conv3_plain<BM, CI_T, TW, 0x036>(p, th, grid, stream); break;
      case 0x0D8: launch_conv3_plain<BM, CI_T, TW, 0x0D8>(p, th, grid, stream); break;
      case 0x1B0: launch_conv3_plain<BM, CI_T, TW, 0x1B0>(p, th, grid, stream); break;
      default: launch_conv3_plain<BM, CI_T, TW, 0x1FF>(p, th, grid, stream); break;
    }
  }
}

template <int BM, int CI_T>
static void launch_conv3(const Conv3Params& p, int th, int tw, dim3 grid, hipStream_t stream) {
  if (tw == 16) launch_conv3_w<BM, CI_T, 16>(p, 4, grid, stream);      // 16-wide tiles always use 4 column blocks
  else launch_conv3_w<BM, CI_T, 32>(p, th, grid, stream);
}

template <int BM, int PREC, int TAPS>
static void launch_conv3_mixed_taps(const Conv3Params& p, int th, int tw, dim3 grid, hipStream_t stream) {
  if (tw == 16) hipLaunchKernelGGL((conv3x3_mixed_kernel<BM, 4, 16, PREC, TAPS>), grid, dim3(256), 0, stream, p);
  else if (th == 8) hipLaunchKernelGGL((conv3x3_mixed_kernel<BM, 8, 32, PREC, TAPS>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv3x3_mixed_kernel<BM, 4, 32, PREC, TAPS>), grid, dim3(256), 0, stream, p);
}

template <int BM, int PREC>
static void launch_conv3_mixed_bm(const Conv3Params& p, int th, int tw, dim3 grid, hipStream_t stream) {
  switch (p.taps) {
    case 0x01B: launch_conv3_mixed_taps<BM, PREC, 0x01B>(p, th, tw, grid, stream); break;
    case 0x036: launch_conv3_mixed_taps<BM, PREC, 0x036>(p, th, tw, grid, stream); break;
    case 0x0D8: launch_conv3_mixed_taps<BM, PREC, 0x0D8>(p, th, tw, grid, stream); break;
    case 0x1B0: launch_conv3_mixed_taps<BM, PREC, 0x1B0>(p, th, tw, grid, stream); break;
    default: launch_conv3_mixed_taps<BM, PREC, 0x1FF>(p, th, tw, grid, stream); break;
  }
}

template <int P>
static void launch_conv3_mixed_small(const Conv3Params& p, int bm, int precision, dim3 grid, hipStream_t stream) {
  if (precision == 1) {
    if (bm == 64) hipLaunchKernelGGL((conv3x3_mixed_small_kernel<64, P, 1>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_mixed_small_kernel<32, P, 1>), grid, dim3(256), 0, stream, p);
  } else {
    if (bm == 64) hipLaunchKernelGGL((conv3x3_mixed_small_kernel<64, P, 2>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_mixed_small_kernel<32, P, 2>), grid, dim3(256), 0, stream, p);
  }
}

static void launch_conv3_mixed(const Conv3Params& p, int bm, int th, int tw, int precision, dim3 grid, hipStream_t stream) {
  if (precision == 1) {
    if (bm == 64) launch_conv3_mixed_bm<64, 1>(p, th, tw, grid, stream);
    else launch_conv3_mixed_bm<32, 1>(p, th, tw, grid, stream);
  } else {
    if (bm == 64) launch_conv3_mixed_bm<64, 2>(p, th, tw, grid, stream);
    else launch_conv3_mixed_bm<32, 2>(p, th, tw, grid, stream);
  }
}

// Declared in gather_gemm_kernels.hip: records a launch for the bench's live event timing.
int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

bool conv3x3_enabled() {
  static const bool disabled = getenv("SRGAN_NO_CONV3") != nullptr;
  return !disabled;
}

// Tile choice: the widest output-channel tile (fewest re-reads of the input patch) and the 8-row pixel tile, as long
// as that still gives ~2 workgroups per CU; otherwise narrower / shorter tiles; input-channel splitting (fp32 atomics
// into a pre-zeroed output) only as the last resort and never below two chunks per workgroup.
struct Conv3Plan { int bm, th, tw, ci_t, tiles_x, tiles_y, tiles_m, split, chunks_per; int64_t blocks; bool small; };

static Conv3Plan conv3x3_plan(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W, bool allow_split = true,
                              int precision = 0, bool small_ok = true) {
  Conv3Plan plan;
  // mixed precision on 4 x 4 / 8 x 8 planes: whole images side by side in a 128-pixel tile (conv3x3_mixed_small_kernel)
  static const bool no_small = getenv("SRGAN_NO_CONV3_SMALL") != nullptr;
  plan.small = precision != 0 && small_ok && !no_small && H == W && (W == 4 || W == 8);
  if (plan.small) {
    const int images = 128 / (H * W);
    plan.tw = W; plan.th = 4; plan.ci_t = 16; plan.tiles_x = 1;
    plan.bm = CO > 32 ? 64 : 32;
    plan.tiles_m = (CO + plan.bm - 1) / plan.bm;
    plan.tiles_y = (N + images - 1) / images;              // image groups
    plan.blocks = (int64_t)plan.tiles_y * plan.tiles_m;
  } else {
  const int tw = W <= 16 ? 16 : 32;      // 16-wide images: a 32-lane column block = two image rows (no dead columns)
  plan.tw = tw;
  plan.tiles_x = (W + tw - 1) / tw;
  static const int bm_cap = getenv("SRGAN_CONV3_BM") ? atoi(getenv("SRGAN_CONV3_BM")) : 64;   // 64 rows: +0.4 % in-step over 128
  int bm = CO > 64 ? 128 : (CO > 32 ? 64 : 32);
  if (bm > bm_cap) bm = bm_cap;
  if (precision && bm > 64) bm = 64;          // mixed precision: 32- / 64-row tiles, 16-channel chunks
  static const int th_cap = getenv("SRGAN_CONV3_TH") ? atoi(getenv("SRGAN_CONV3_TH")) : 8;
  int th = (bm == 128 || tw == 16 || th_cap < 8) ? 4 : 8;   // 128 rows / 16-wide tiles always use 4 column blocks per workgroup
  auto rows = [&](int th_) { return th_ * (32 / tw); };
  auto count = [&](int bm_, int th_) {
    return (int64_t)N * ((H + rows(th_) - 1) / rows(th_)) * plan.tiles_x * ((CO + bm_ - 1) / bm_);
  };
  while (count(bm, th) < 512) {
    if (th == 8) th = 4;
    else if (bm > 32) { bm >>= 1; th = 4; }
    else break;
  }
  plan.bm = bm; plan.th = th;
  static const int ci_t32 = getenv("SRGAN_CONV3_CIT32") ? atoi(getenv("SRGAN_CONV3_CIT32")) : 8;
  static const int ci_t64 = getenv("SRGAN_CONV3_CIT64") ? atoi(getenv("SRGAN_CONV3_CIT64")) : 4;
  plan.ci_t = bm == 128 ? 4 : (bm == 64 ? ci_t64 : ci_t32);     // keeps the staged registers + accumulators <= 256
  if (precision) plan.ci_t = 16;
  plan.tiles_m = (CO + bm - 1) / bm;
  plan.tiles_y = (H + rows(th) - 1) / rows(th);
  plan.blocks = count(bm, th);
  }
  const int chunks = (CI + plan.ci_t - 1) / plan.ci_t;
  int split = 1;
  static const int split_below = getenv("SRGAN_CONV3_SPLIT_BELOW") ? atoi(getenv("SRGAN_CONV3_SPLIT_BELOW")) : 384;
  if (allow_split && plan.blocks < split_below && chunks >= 4) {
    split = (int)((512 + plan.blocks - 1) / plan.blocks);
    if (split > chunks / 2) split = chunks / 2;
  }
  plan.chunks_per = (chunks + split - 1) / split;
  plan.split = (chunks + plan.chunks_per - 1) / plan.chunks_per;
  return plan;
}

// (of a launch with tap subsets / strided stores -- the k4 / s2 classes --, which the small-plane kernel does not take)
int conv3x3_splits(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W, int precision) {
  return conv3x3_plan(N, CI, CO, H, W, true, precision, false).split;
}

// The batch-norm backward epilogue needs a 32- or 64-row tile (and whole sums in one workgroup: its plan never splits K).
bool conv3x3_epilogue_supported(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W) {
  return conv3x3_plan(N, CI, CO, H, W, false).bm <= 64;      // (with the epilogue the plan does not split K)
}

int64_t conv3x3_epilogue_tiles(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W) {
  const Conv3Plan plan = conv3x3_plan(N, CI, CO, H, W, false);
  return plan.blocks / plan.tiles_m;
}

void bn_partial_reduce_run(const float* partial, int tiles, int CO, const float* inv_std, float* g_gamma, float* g_beta,
                           hipStream_t stream);
float* partial_workspace(size_t bytes, hipStream_t stream);

// out = conv3x3(in, w) (+ bias), generic weight strides (forward and flipped-tap data gradient share the kernel).
// The caller guarantees dense-or-strided NCHW, 3x3 / stride 1 / pad 1.  `accumulate` adds into out.
int conv3x3_run(const float* in, int64_t in_bs, const float* w, int32_t w_base, int32_t w_so, int32_t w_si, int32_t w_skh,
                int32_t w_skw, const float* bias, float* out, int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t H,
                int32_t W, int accumulate, hipStream_t stream, const float* const* bn, const BnBackwardEpilogue* epilogue,
                int precision, const Conv3Placement* placement) {
  Conv3Params p;
  p.taps = placement ? placement->taps : 0x1FF;
  p.out_plane = placement ? placement->out_plane : H * W;
  p.out_sy = placement ? placement->out_sy : W;
  p.out_sx = placement ? placement->out_sx : 1;
  p.out_off = placement ? placement->out_off : 0;
  SRGAN_REQUIRE(placement == nullptr || (bn == nullptr && epilogue == nullptr), SRGAN_EUNSUPPORTED,
                "conv3x3 tap subsets / strided output: plain kernel only");
  SRGAN_REQUIRE(precision == 0 || (bn == nullptr && epilogue == nullptr), SRGAN_EUNSUPPORTED,
                "conv3x3 mixed precision: the fused batch-norm forms are fp32");
  p.in = in; p.w = w; p.out = out; p.bias = bias;
  p.bn_mean = bn ? bn[0] : nullptr; p.bn_inv = bn ? bn[1] : nullptr;
  p.bn_gamma = bn ? bn[2] : nullptr; p.bn_beta = bn ? bn[3] : nullptr;
  p.epi_x = nullptr; p.epi_x_bs = 0; p.epi_partial = nullptr; p.epi_tiles = 0; p.w_packed = nullptr;
  SRGAN_REQUIRE(bn == nullptr || CI <= CONV3_PRO_MAX_CI, SRGAN_EUNSUPPORTED, "conv3x3 fused batch-norm channel count");
  p.N = N; p.CI = CI; p.CO = CO; p.H = H; p.W = W;
  p.in_bs = in_bs; p.out_bs = out_bs;
  p.w_so = w_so; p.w_si = w_si; p.w_skh = w_skh; p.w_skw = w_skw; p.w_base = w_base;
  const Conv3Plan plan = conv3x3_plan(N, CI, CO, H, W, epilogue == nullptr, precision, placement == nullptr);   // (the epilogue needs whole sums per workgroup)
  const int bm = plan.bm, th = plan.th, tw = plan.tw, split = plan.split;
  p.tiles_x = plan.tiles_x; p.tiles_y = plan.tiles_y; p.tiles_m = plan.tiles_m;
  p.ci_per_split = plan.chunks_per * plan.ci_t;
  const int64_t blocks = plan.blocks;
  SRGAN_REQUIRE(blocks < (int64_t)1 << 31 && split <= 65535, SRGAN_ERANGE, "conv3x3 grid");
  p.split_ws = nullptr; p.split_tickets = nullptr;
  const int packed_slots = precision ? ((CI + 15) / 16) * 18 * CO : 0;      // mixed precision: operand slots of the packed weights
  if (split > 1) {                       // accumulate: 0 store, 1 add to out, 2 out is already zero
    // Ordered finish (split_finish.h): partial tiles through the workspace, the tile's last workgroup adds them in slice
    // order and stores -- one launch, no zero-fill, the same bits every run.  Without a workspace: zero-fill + atomics.
    int ticket_set = -1;
    const int64_t accumulators = (plan.small ? bm / 32 : (bm / 32) * (th / 4)) * 16 * 256;
    float* ws = split_workspace(blocks, split, accumulators, (size_t)packed_slots * sizeof(Half8), stream, &ticket_set);
    unsigned int* tickets = ws ? device_tickets(g_conv3_split_tickets) : nullptr;
    if (ws && tickets) {
      p.mode = accumulate == 1 ? 4 : 3;
      p.split_ws = ws;
      p.split_tickets = tickets + (size_t)ticket_set * SPLIT_TICKET_TILES;
    } else {
      SRGAN_REQUIRE(placement == nullptr || accumulate != 0, SRGAN_EUNSUPPORTED,
                    "conv3x3 strided output with an atomic K split needs a pre-zeroed (or accumulated) output");
      if (accumulate == 0)
        if (const int status = zero_rows(out, out_bs, (int64_t)CO * H * W, N, stream)) return status;
      p.mode = 2;
    }
  } else {
    p.mode = accumulate == 1 ? 1 : 0;
  }
  if (epilogue) {
    SRGAN_REQUIRE(!bn && !bias && !accumulate && split == 1 && bm <= 64, SRGAN_EUNSUPPORTED,
                  "conv3x3 batch-norm backward epilogue (store mode, unsplit, <= 64-row tile)");
    p.epi_x = epilogue->x; p.epi_x_bs = epilogue->x_bs;
    p.bn_mean = epilogue->bn[0]; p.bn_inv = epilogue->bn[1]; p.bn_gamma = epilogue->bn[2]; p.bn_beta = epilogue->bn[3];
    p.epi_tiles = (int32_t)(blocks / plan.tiles_m);
    if (epilogue->partial_out) {
      p.epi_partial = epilogue->partial_out;
    } else if (epilogue->g_gamma) {
      p.epi_partial = partial_workspace((size_t)2 * p.epi_tiles * CO * sizeof(float), stream);
      SRGAN_REQUIRE(p.epi_partial, SRGAN_EINVAL, "conv3x3 batch-norm backward epilogue: register a workspace for this "
                    "stream first (srgan_set_workspace, >= srgan_workspace_bytes())");
    }
  }
  static const bool no_xcd = getenv("SRGAN_NO_XCD_ORDER") != nullptr;
  p.xcd_remap = (!no_xcd && blocks % 8 == 0 && blocks >= 64) ? 1 : 0;
  dim3 grid((unsigned)blocks, (unsigned)split, 1);
  const int profile_slot = profile_bracket_begin(stream);
  if (precision) {
    // the weights in operand slots, in the caller's workspace (2 bytes per weight element, padded to 16-channel chunks)
    const int slots = packed_slots;
    Half8* packed = reinterpret_cast<Half8*>(partial_workspace((size_t)slots * sizeof(Half8), stream));
    SRGAN_REQUIRE(packed, SRGAN_EINVAL, "conv3x3 mixed precision: register a workspace for this stream first "
                  "(srgan_set_workspace, >= srgan_workspace_bytes(); the packed weights take 2 bytes per element)");
    if (precision == 1) hipLaunchKernelGGL(conv3x3_pack_weights_kernel<1>, dim3((slots + 255) / 256), dim3(256), 0, stream, p, packed, slots);
    else hipLaunchKernelGGL(conv3x3_pack_weights_kernel<2>, dim3((slots + 255) / 256), dim3(256), 0, stream, p, packed, slots);
    p.w_packed = packed;
    if (plan.small && W == 4) launch_conv3_mixed_small<4>(p, bm, precision, grid, stream);
    else if (plan.small) launch_conv3_mixed_small<8>(p, bm, precision, grid, stream);
    else launch_conv3_mixed(p, bm, th, tw, precision, grid, stream);
  }
  else if (bm == 32 && plan.ci_t == 16) launch_conv3<32, 16>(p, th, tw, grid, stream);
  else if (bm == 32) launch_conv3<32, 8>(p, th, tw, grid, stream);
  else if (bm == 64 && plan.ci_t == 8) launch_conv3<64, 8>(p, th, tw, grid, stream);
  else if (bm == 64) launch_conv3<64, 4>(p, th, tw, grid, stream);
  else launch_conv3<128, 4>(p, 4, tw, grid, stream);
  if (p.epi_partial && !epilogue->partial_out)
    bn_partial_reduce_run(p.epi_partial, p.epi_tiles, CO, p.bn_inv, epilogue->g_gamma, epilogue->g_beta, stream);
  const int status = launch_status();
  const int64_t pixels3 = (int64_t)N * H * W;       // (+ x read by the fused batch-norm backward epilogue)
  profile_bracket_end(profile_slot, stream, CO, pixels3, (int64_t)CI * __builtin_popcount((unsigned)p.taps), 2, bm, th * 32, split, 0, 0,
                      (int64_t)CI * pixels3 + (epilogue ? (int64_t)CO * pixels3 : 0), precision);
  return status;
}

}  // namespace srgan
