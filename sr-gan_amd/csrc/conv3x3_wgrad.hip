// conv3x3_wgrad.hip -- weight gradient of 3x3 / stride 1 / pad 1 convolutions (the DenseNet growth convolutions,
// reference crowd/models.py:344-345; autograd's conv weight gradient behind loss.backward(), srgan.py:264,295,304).
//
//   gw[co, ci, kh, kw] += sum_{n,h,w} gy[n, co, h, w] * x[n, ci, h + kh - 1, w + kw - 1]
//
// As a gather-GEMM this is M = CO (32 for every growth convolution) x N = CI*9 x K = pixels, and the generic kernel
// gathers every x element nine times (once per tap) for a single 32-row MFMA each: it ran at 27-47 TF/s.  Here one
// workgroup stages a TH x 16 pixel tile of gy (32 output channels) and the matching (TH+2) x 18 halo patch of x
// (32 input channels) in LDS ONCE and all nine taps read it.  MFMA roles (v_mfma_f32_32x32x2_f32): rows = co,
// columns = ci, k = two horizontally adjacent pixels.  The workgroup is 3 waves, one per kernel row kh; a wave keeps
// its three 32x32 tap accumulators (48 registers) across all the pixel tiles the workgroup walks and adds them to gw
// with fp32 atomics once at the end.  The next tile is fetched into registers while the current one is in the MFMAs
// (raw loads from clamped addresses; the validity mask is applied when the registers are written to LDS -- a use of
// the loaded value before the MFMA loop would make the compiler wait for the loads there).
#include "common.h"
#include "split_finish.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace srgan {

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Wgrad3Params {
  const float* x; const float* gy; float* gw;
  int64_t x_bs, gy_bs;
  int32_t N, CI, CO, H, W;
  int32_t tiles_x, tiles_y, tiles;
  int32_t debug;                  // tuning experiments (SRGAN_WGRAD3_DEBUG): 1 skip the atomic pass
  float* partial;                 // non-NULL (round 5): walker `w` of channel chunk `c` stores its 32 x 288 block at
                                  // partial[(c * walkers + w) * 9216 ...]; conv3x3_wgrad_finish adds the walkers in order
  // x is relu(batch_norm_eval(x)) computed on the fly (per input channel) when bn_mean != NULL
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
};

constexpr int WG3_TW = 16;
constexpr int WG3_CI = 32;      // input channels per workgroup (one MFMA column block)
constexpr int WG3_CO = 32;
constexpr int WG3_THREADS = 192;

// PREC = 1 / 2: bf16 / fp16 MFMA operands (v_mfma_f32_32x32x16_*): a 16-deep step is one 16-pixel tile row, a lane rounds
// the 8 pixels of its half; the three kernel columns share one window of 10 patch values.
// RAGGED: widths that are not a multiple of 4 or rows that are only 4-byte aligned (the 14- and 7-wide planes at the
// reference's 224 x 224): the staging loads are single floats with per-element validity instead of aligned float4.
template <int TH, int PREC, bool RAGGED>
struct Wgrad3Lds {
  static constexpr int PW = WG3_TW + 2, PH = TH + 2, PATCH = PH * PW, PS = PATCH | 1, GT = TH * WG3_TW, GS = GT | 1;
  static constexpr int OUT_ROW = WG3_CI * 9 + 1, OUT_HALF = WG3_CO / 2;
  static constexpr int SMEM = WG3_CI * PS + WG3_CO * GS > OUT_HALF * OUT_ROW ? WG3_CI * PS + WG3_CO * GS : OUT_HALF * OUT_ROW;
};

// `first_tile` / `tile_stride`: this workgroup's walk over the pixel tiles; `ci_chunk` / `co_chunk`: its channel blocks.
template <int TH, int PREC = 0, bool RAGGED = false>
__device__ __forceinline__ void conv3x3_wgrad_body(const Wgrad3Params& p, const int first_tile, const int tile_stride,
                                                   const int ci_chunk, const int co_chunk, float* smem) {
  constexpr int PW = WG3_TW + 2, PH = TH + 2;
  constexpr int PATCH = PH * PW;            // halo patch of one channel
  constexpr int PS = PATCH | 1;             // odd plane stride: the 32 lanes of an operand read hit 32 banks
  constexpr int GT = TH * WG3_TW;           // gy tile of one channel
  constexpr int GS = GT | 1;
  static_assert(PH * 32 <= WG3_THREADS, "one (patch row, channel) pair per thread");
  constexpr int OUT_ROW = WG3_CI * 9 + 1;   // epilogue transpose buffer: [co][ci*9 + tap], odd row stride; the 32 output
  constexpr int OUT_HALF = WG3_CO / 2;      // rows go through it in two halves so that it fits under the staging tiles
  float* xs = smem;
  float* gs = smem + WG3_CI * PS;

  const int tid = (int)threadIdx.x, lane = tid & 63, kh = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ci0 = ci_chunk * WG3_CI, co0 = co_chunk * WG3_CO;
  const int HW = p.H * p.W;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* a_base = gs + l31 * GS + lhi;
  const float* b_base = xs + l31 * PS + kh * PW + lhi;
  // PREC: the tiles are OPERAND-TYPED in LDS (round 3).  x patch: [channel][patch row][24] (18 used; 48-byte rows, channel
  // stride 6 x 48 + 16 bytes), gy tile: [channel][TH x 16] (+ 8 elements of padding): a lane's fragment -- 8 consecutive pixels
  // of its half of a 16-pixel row -- is ONE ds_read_b128, and the three kernel columns come out of one window of ten patch
  // values (a b128 and a b32) by shifting: it used to be 18 ds_read_b32 of fp32 values and 32 conversions per row step.
  using half_t = typename std::conditional<PREC == 2, _Float16, __bf16>::type;
  constexpr int XW = 24, XC = PH * XW + 8, GC = GT + 8;             // typed row width / channel strides (elements)
  half_t* xs16 = reinterpret_cast<half_t*>(smem);
  half_t* gs16 = xs16 + WG3_CI * XC;
  static_assert(PREC == 0 || (WG3_CI * (PH * 24 + 8) + WG3_CO * (GT + 8)) * 2 <= Wgrad3Lds<TH, PREC, RAGGED>::SMEM * 4,
                "the typed tiles fit the fp32 allocation");

  // Staging ownership: thread -> (row, channel).  x: one patch row of one input channel = the 6 aligned float4 that
  // cover columns [w0 - 4, w0 + 20) (18 of the 24 floats are the halo row); gy: one tile row of one output channel =
  // 4 float4.  W % 4 == 0 (checked by the launcher) makes every float4 entirely inside or entirely outside the row.
  const int srow = tid >> 5, sch = tid & 31;
  const bool x_owner = srow < PH, g_owner = srow < TH;
  const bool x_ch_ok = ci0 + sch < p.CI, g_ch_ok = co0 + sch < p.CO;
  const uint32_t x_ch_off = (uint32_t)(min(ci0 + sch, p.CI - 1) * HW);
  const uint32_t g_ch_off = (uint32_t)(min(co0 + sch, p.CO - 1) * HW);
  float pro_a = 1.f, pro_b = 0.f;            // fused batch-norm + ReLU of this thread's input channel
  const bool pro = p.bn_mean != nullptr;
  if (pro) {
    const int c = min(ci0 + sch, p.CI - 1);
    bn_coefficients(p.bn_mean[c], p.bn_inv[c], p.bn_gamma[c], p.bn_beta[c], pro_a, pro_b);
  }
  float* xs_row = xs + sch * PS + srow * PW;
  float* gs_row = gs + sch * GS + srow * WG3_TW;

  float4 xv[RAGGED ? 1 : 6], gv[RAGGED ? 1 : 4];
  float xr[RAGGED ? PW : 1], gr[RAGGED ? WG3_TW : 1];
  uint32_t okbits = 0, okbits_g = 0;

  // Issues the loads of one tile (no use of the values here) and records which of them are real.
  auto fetch = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int rest = tile / p.tiles_x;
    const int ty = rest % p.tiles_y;
    const int n = rest / p.tiles_y;
    const int h0 = ty * TH, w0 = tx * WG3_TW;
    const float* xn = p.x + (int64_t)n * p.x_bs;
    const float* gn = p.gy + (int64_t)n * p.gy_bs;
    okbits = 0;
    if constexpr (RAGGED) {
      okbits_g = 0;
      if (x_owner) {
        const int h = h0 + srow - 1;
        const bool row_ok = x_ch_ok && (unsigned)h < (unsigned)p.H;
        const uint32_t row_off = x_ch_off + (uint32_t)(min(max(h, 0), p.H - 1) * p.W);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
          const int col = w0 - 1 + i;
          okbits |= (row_ok && (unsigned)col < (unsigned)p.W ? 1u : 0u) << i;
          xr[i] = xn[row_off + (uint32_t)min(max(col, 0), p.W - 1)];
        }
      }
      if (g_owner) {
        const int h = h0 + srow;
        const bool row_ok = g_ch_ok && h < p.H;
        const uint32_t row_off = g_ch_off + (uint32_t)(min(h, p.H - 1) * p.W);
#pragma unroll
        for (int i = 0; i < WG3_TW; ++i) {
          okbits_g |= (row_ok && w0 + i < p.W ? 1u : 0u) << i;
          gr[i] = gn[row_off + (uint32_t)min(w0 + i, p.W - 1)];
        }
      }
      return;
    }
    if (x_owner) {
      const int h = h0 + srow - 1;
      const bool row_ok = x_ch_ok && (unsigned)h < (unsigned)p.H;
      const uint32_t row_off = x_ch_off + (uint32_t)(min(max(h, 0), p.H - 1) * p.W);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int col = w0 - 4 + 4 * q;
        const bool ok = row_ok && (unsigned)col < (unsigned)p.W;
        okbits |= (ok ? 1u : 0u) << q;
        xv[q] = *reinterpret_cast<const float4*>(xn + row_off + (uint32_t)min(max(col, 0), p.W - 4));
      }
    }
    if (g_owner) {
      const int h = h0 + srow;
      const bool row_ok = g_ch_ok && h < p.H;
      const uint32_t row_off = g_ch_off + (uint32_t)(min(h, p.H - 1) * p.W);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = w0 + 4 * q;
        const bool ok = row_ok && col < p.W;
        okbits |= (ok ? 1u : 0u) << (8 + q);
        gv[q] = *reinterpret_cast<const float4*>(gn + row_off + (uint32_t)min(col, p.W - 4));
      }
    }
  };

  int tile = first_tile;
  if (tile < p.tiles) fetch(tile);
  for (; tile < p.tiles; tile += tile_stride) {
    __syncthreads();                        // the previous tile's MFMA reads are done
    if constexpr (RAGGED) {
      if (x_owner) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
          float v = xr[i];
          if (pro) v = fmaxf(fmaf(v, pro_a, pro_b), 0.f);
          xs_row[i] = (okbits >> i) & 1u ? v : 0.f;
        }
      }
      if (g_owner) {
#pragma unroll
        for (int i = 0; i < WG3_TW; ++i) gs_row[i] = (okbits_g >> i) & 1u ? gr[i] : 0.f;
      }
    } else if constexpr (PREC != 0) {
      typedef half_t half2_t __attribute__((ext_vector_type(2)));
      if (x_owner) {
        float v[24];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const bool ok = (okbits >> q) & 1u;
          const float f[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = f[j];
            if (pro) t = fmaxf(fmaf(t, pro_a, pro_b), 0.f);
            v[4 * q + j] = ok ? t : 0.f;
          }
        }
        half2_t* row = reinterpret_cast<half2_t*>(xs16 + sch * XC + srow * XW);
#pragma unroll
        for (int c = 0; c < PW; c += 2) {    // patch column c = float c + 3 (patch column 0 = image column w0 - 1)
          half2_t pair;
          pair[0] = (half_t)v[c + 3]; pair[1] = (half_t)v[c + 4];
          row[c >> 1] = pair;
        }
      }
      if (g_owner) {
        half2_t* row = reinterpret_cast<half2_t*>(gs16 + sch * GC + srow * WG3_TW);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (okbits >> (8 + q)) & 1u;
          half2_t lo, hi;
          lo[0] = (half_t)(ok ? gv[q].x : 0.f); lo[1] = (half_t)(ok ? gv[q].y : 0.f);
          hi[0] = (half_t)(ok ? gv[q].z : 0.f); hi[1] = (half_t)(ok ? gv[q].w : 0.f);
          row[2 * q] = lo; row[2 * q + 1] = hi;
        }
      }
    } else {
    if (x_owner) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const bool ok = (okbits >> q) & 1u;
        float v[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pro) v[j] = fmaxf(fmaf(v[j], pro_a, pro_b), 0.f);
          v[j] = ok ? v[j] : 0.f;            // padding is applied to the activated tensor
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 4 * q + j - 3;      // patch column of float j of quad q (patch column 0 = image column w0 - 1)
          if (c >= 0 && c < PW) xs_row[c] = v[j];
        }
      }
    }
    if (g_owner) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = (okbits >> (8 + q)) & 1u;
        gs_row[4 * q + 0] = ok ? gv[q].x : 0.f;
        gs_row[4 * q + 1] = ok ? gv[q].y : 0.f;
        gs_row[4 * q + 2] = ok ? gv[q].z : 0.f;
        gs_row[4 * q + 3] = ok ? gv[q].w : 0.f;
      }
    }
    }
    __syncthreads();
    const int next = tile + tile_stride;
    if (next < p.tiles) fetch(next);

    if constexpr (PREC != 0) {
      using frag = typename std::conditional<PREC == 1, bf16x8, f16x8>::type;
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const half_t* a16 = gs16 + l31 * GC + 8 * lhi;
      const half_t* b16 = xs16 + l31 * XC + kh * XW + 8 * lhi;
#pragma unroll
      for (int h = 0; h < TH; ++h) {
        const frag a = *reinterpret_cast<const frag*>(a16 + h * WG3_TW);
        // ten patch values from column 8 * lhi on: d[0..3] = columns 0..7 (kw = 0), d[1..4] = columns 2..9 (kw = 2), and
        // kw = 1 is every pair shifted by one element
        const u32x4 lo = *reinterpret_cast<const u32x4*>(b16 + h * XW);
        const uint32_t top = *reinterpret_cast<const uint32_t*>(b16 + h * XW + 8);
        const uint32_t d[5] = {lo[0], lo[1], lo[2], lo[3], top};
        u32x4 w0, w1, w2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          w0[i] = d[i];
          w1[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);
          w2[i] = d[i + 1];
        }
        const frag b0 = __builtin_bit_cast(frag, w0), b1 = __builtin_bit_cast(frag, w1), b2 = __builtin_bit_cast(frag, w2);
        if constexpr (PREC == 1) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc[2], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc[2], 0, 0, 0);
        }
      }
    } else
#pragma unroll 1
    for (int h = 0; h < TH; ++h) {          // rolled: bounds the LDS values the scheduler keeps in flight
#pragma unroll
      for (int w = 0; w < WG3_TW; w += 2) {
        const float a = a_base[h * WG3_TW + w];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float b = b_base[h * PW + w + kw];
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kw], 0, 0, 0);
        }
      }
    }
  }

  // ---- one atomic pass per workgroup.  C/D fragment: column = lane & 31 (ci), row = (r & 3) + 8 * (r >> 2) +
  // 4 * (lane >> 5) (co).  In that layout a wave's lanes are 36 bytes apart in gw (one L2 atomic transaction per
  // lane: measured 630 us per launch), so the 32 x 288 block is transposed through LDS and the atomics go out with
  // lanes along the contiguous (ci, tap) run of each output-channel row.
  if (p.debug & 1) return;
  const int run = min(WG3_CI, p.CI - ci0) * 9;       // valid floats of each row
#pragma unroll
  for (int half = 0; half < 2; ++half) {             // rows 16 * half ... : accumulator registers 8 * half ...
    __syncthreads();
#pragma unroll
    for (int r = 8 * half; r < 8 * half + 8; ++r) {
      const int co = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi;          // row within the half
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) smem[co * OUT_ROW + l31 * 9 + kh * 3 + kw] = acc[kw][r];
    }
    __syncthreads();
    if (p.partial) {       // the ordered form: this walker's block, coalesced; a second kernel adds the walkers in order
      float* mine = p.partial + ((int64_t)(co_chunk * ((p.CI + WG3_CI - 1) / WG3_CI) + ci_chunk) * tile_stride + first_tile) *
                                    (WG3_CO * WG3_CI * 9) + half * (OUT_HALF * WG3_CI * 9);
      for (int idx = tid; idx < OUT_HALF * WG3_CI * 9; idx += WG3_THREADS) {
        const int co = idx / (WG3_CI * 9), within = idx - co * (WG3_CI * 9);
        mine[idx] = smem[co * OUT_ROW + within];
      }
      continue;
    }
    for (int idx = tid; idx < OUT_HALF * WG3_CI * 9; idx += WG3_THREADS) {
      const int co = idx / (WG3_CI * 9), within = idx - co * (WG3_CI * 9);
      const int row = co0 + OUT_HALF * half + co;
      if (row < p.CO && within < run)
        unsafeAtomicAdd(p.gw + ((int64_t)row * p.CI + ci0) * 9 + within, smem[co * OUT_ROW + within]);
    }
  }
}

// Second stage of the ordered form: gw[channel chunk] += its walkers' 32 x 288 blocks, added in walker order.  One thread per
// element of the block (`slab` = which 256 of its 9216), the walkers' blocks read with lanes along the elements.
__device__ __forceinline__ void conv3x3_wgrad_finish_chunk(const float* __restrict__ partial, float* __restrict__ gw, int chunk,
                                                           int slab, int walkers, int ci_chunks, int CO, int CI) {
  constexpr int BLOCK = WG3_CO * WG3_CI * 9;
  const int co_chunk = chunk / ci_chunks, ci_chunk = chunk - co_chunk * ci_chunks;
  const int co0 = co_chunk * WG3_CO, ci0 = ci_chunk * WG3_CI;
  const int run = min(WG3_CI, CI - ci0) * 9;
  const int idx = slab * 256 + (int)threadIdx.x;
  const int co = idx / (WG3_CI * 9), within = idx - co * (WG3_CI * 9);
  if (co0 + co >= CO || within >= run) return;
  const float* mine = partial + (int64_t)chunk * walkers * BLOCK + idx;
  float total = mine[0];
  int w = 1;
  for (; w + 3 < walkers; w += 4) {              // four loads in flight, one fixed order
    const float a = mine[(int64_t)w * BLOCK], b = mine[(int64_t)(w + 1) * BLOCK];
    const float c = mine[(int64_t)(w + 2) * BLOCK], d = mine[(int64_t)(w + 3) * BLOCK];
    total = (((total + a) + b) + c) + d;
  }
  for (; w < walkers; ++w) total += mine[(int64_t)w * BLOCK];
  gw[((int64_t)(co0 + co) * CI + ci0) * 9 + within] += total;
}

constexpr int WG3_FINISH_SLABS = WG3_CO * WG3_CI * 9 / 256;       // 36

__global__ __launch_bounds__(256) void conv3x3_wgrad_finish_kernel(const Wgrad3Params p, const int walkers, const int ci_chunks) {
  conv3x3_wgrad_finish_chunk(p.partial, p.gw, (int)blockIdx.y, (int)blockIdx.x, walkers, ci_chunks, p.CO, p.CI);
}

// One-dimensional grid in XCD-aware order (hardware workgroup b runs on XCD b % 8): the walkers go round-robin to the XCDs and
// ALL channel-chunk workgroups of a walker follow each other on its XCD -- they read the same pixel tiles (an x chunk is
// wanted by every output-channel chunk, a gy chunk by every input-channel chunk: 16 + 16 distinct tiles for 256 workgroups at
// 512 -> 512 channels), so each tile comes from HBM once and from that XCD's L2 afterwards.  With (walker, ci chunk, co chunk)
// as a three-dimensional grid the chunks of a walker were spread over all eight L2s and over time: 28.1 GB of fabric traffic
// for 4.8 GB of operands on the VGG-16 workload (5.9x, profiles/r03z_age_vgg64_bf16_pmc_per_kernel.md).
template <int TH, int PREC = 0, bool RAGGED = false>
__global__ __launch_bounds__(WG3_THREADS, 3) void conv3x3_wgrad_kernel(const Wgrad3Params p, const int walkers,
                                                                       const int ci_chunks, const int co_chunks) {
  __shared__ float smem[Wgrad3Lds<TH, PREC, RAGGED>::SMEM];
  const int chunks = ci_chunks * co_chunks;
  const int xcd = (int)blockIdx.x & 7, within_xcd = (int)blockIdx.x >> 3;
  const int walker = (within_xcd / chunks) * 8 + xcd, chunk = within_xcd % chunks;
  if (walker >= walkers) return;                                       // (the last round's padding; workgroup-uniform)
  const int co_chunk = chunk / ci_chunks;
  conv3x3_wgrad_body<TH, PREC, RAGGED>(p, walker, walkers, chunk - co_chunk * ci_chunks, co_chunk, smem);
}

// GROUPED: blockIdx.z selects one of many independent problems (all the growth convolutions of a dense block's backward)
// from a device-resident table; x / gy are offsets from two base pointers.  blockIdx.y = ci chunk + ci_chunks * co chunk.
struct Wgrad3Job {
  int64_t x_off, gy_off, x_bs, gy_bs;
  float* gw;
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
  int32_t N, CI, CO, H, W, tiles_x, tiles_y, tiles, walkers, ci_chunks, co_chunks, pad;
  int64_t partial_off;            // ordered form: this job's partial blocks start here (floats) in the launch's workspace region
};
static_assert(sizeof(Wgrad3Job) == 128, "one 128-byte table slot per job");

// XCD-aware order (as in pointwise_wgrad.hip): the channel chunks of one walker read the same gy tiles, and neighbouring
// walkers visit neighbouring tiles at the same time (they share the 128-byte lines of a tile row and the halo columns).
// With blockIdx.x = walker running fastest those workgroups were spread over the eight XCDs and over time: 52.9 GB of fabric
// traffic per step for 12.6 GB of operands (4.2x, profiles/r02z_pmc_per_kernel.md).  Now the grid is one-dimensional: a UNIT
// is WG3_UNIT neighbouring walkers x all channel chunks of one problem, XCD b % 8 works through the units 8 * j + (b % 8).
constexpr int WG3_UNIT = 8;

template <bool RAGGED>
__global__ __launch_bounds__(WG3_THREADS, 3) void conv3x3_wgrad_grouped_kernel(const Wgrad3Job* __restrict__ jobs,
                                                                               const float* x_base, const float* gy_base,
                                                                               float* gw_base, const int grid_x, const int grid_y,
                                                                               const int count, float* partial_base) {
  __shared__ float smem[Wgrad3Lds<4, 0, RAGGED>::SMEM];
  const int xcd = (int)blockIdx.x & 7, within_xcd = (int)blockIdx.x >> 3;
  const int per_unit = WG3_UNIT * grid_y, units_per_job = (grid_x + WG3_UNIT - 1) / WG3_UNIT;
  const int round = within_xcd / per_unit, within = within_xcd % per_unit;
  const int unit = round * 8 + ((xcd - round) & 7);
  const int z = unit / units_per_job;
  if (z >= count) return;                                                                             // (the last round's padding)
  const int walker = (unit - z * units_per_job) * WG3_UNIT + within / grid_y, chunk = within % grid_y;
  const Wgrad3Job job = jobs[z];
  if (walker >= job.walkers || chunk >= job.ci_chunks * job.co_chunks) return;                        // (workgroup-uniform)
  Wgrad3Params p;
  p.x = x_base + job.x_off; p.gy = gy_base + job.gy_off; p.x_bs = job.x_bs; p.gy_bs = job.gy_bs;
  p.gw = gw_base ? gw_base + (int64_t)(intptr_t)job.gw : job.gw;      // (an element offset into the per-step buffer)
  p.N = job.N; p.CI = job.CI; p.CO = job.CO; p.H = job.H; p.W = job.W;
  p.tiles_x = job.tiles_x; p.tiles_y = job.tiles_y; p.tiles = job.tiles; p.debug = 0;
  p.bn_mean = job.bn_mean; p.bn_inv = job.bn_inv; p.bn_gamma = job.bn_gamma; p.bn_beta = job.bn_beta;
  p.partial = partial_base ? partial_base + job.partial_off : nullptr;
  const int co_chunk = chunk / job.ci_chunks;
  conv3x3_wgrad_body<4, 0, RAGGED>(p, walker, job.walkers, chunk - co_chunk * job.ci_chunks, co_chunk, smem);
}

// blockIdx.x = 256-element slab of a block, blockIdx.y = channel chunk, blockIdx.z = job: the second stage of a grouped launch.
__global__ __launch_bounds__(256) void conv3x3_wgrad_grouped_finish_kernel(const Wgrad3Job* __restrict__ jobs, float* gw_base,
                                                                           const float* __restrict__ partial_base) {
  const Wgrad3Job job = jobs[blockIdx.z];
  if ((int)blockIdx.y >= job.ci_chunks * job.co_chunks) return;
  float* gw = gw_base ? gw_base + (int64_t)(intptr_t)job.gw : job.gw;
  conv3x3_wgrad_finish_chunk(partial_base + job.partial_off, gw, (int)blockIdx.y, (int)blockIdx.x, job.walkers, job.ci_chunks, job.CO,
                             job.CI);
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

bool conv3x3_wgrad_enabled() {
  static const bool disabled = getenv("SRGAN_NO_WGRAD3") != nullptr;
  return !disabled;
}

// Walkers over the pixel tiles: five resident 3-wave workgroups per CU (120 VGPRs, 22 KB of LDS) over the whole grid, and
// at least `depth` tiles per walker so that the atomic pass (32 x 288 floats per workgroup) is amortised.
static int conv3x3_wgrad_walkers(int tiles, int ci_chunks, int co_chunks, int group = 1) {
  static const int resident = getenv("SRGAN_WGRAD3_WGS") ? atoi(getenv("SRGAN_WGRAD3_WGS")) : 1280;
  static const int depth_override = getenv("SRGAN_WGRAD3_DEPTH") ? atoi(getenv("SRGAN_WGRAD3_DEPTH")) : 0;
  // measured on 128 -> 32 channels, batch 16: 64x64 images best at 8 tiles per walker, 32x32 at 4, 16x16 at 1 (batch
  // 48 at 16x16 = 192 tiles: 48 us at depth 1 -- 768 workgroups x 9216 atomics -- so 4 from 128 tiles up)
  const int depth = depth_override > 0 ? depth_override : (tiles >= 1024 ? 8 : (tiles >= 128 ? 4 : 1));
  static const int oversubscription = getenv("SRGAN_GROUP_OVERSUB") ? atoi(getenv("SRGAN_GROUP_OVERSUB")) : 4;
  const int wanted = group > 1 ? (resident * oversubscription + group - 1) / group : resident;
  int walkers = wanted / (ci_chunks * co_chunks);
  if (walkers > (tiles + depth - 1) / depth) walkers = (tiles + depth - 1) / depth;
  if (group > 1 && walkers > 8) walkers -= walkers % 8;     // whole units of the grouped kernel's XCD-aware order
  if (group == 1) {                                         // single problem: walkers go round-robin to the eight XCDs
    walkers = (walkers + 7) / 8 * 8;
    if (walkers > tiles) walkers = tiles;
  }
  return walkers < 1 ? 1 : walkers;
}

// gw (=,+=) the weight gradient; x / gy may be channel-slice views (batch strides in elements).
int conv3x3_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t CI,
                      int32_t CO, int32_t H, int32_t W, int accumulate, hipStream_t stream, const float* const* bn,
                      int precision) {
  static const int th_override = getenv("SRGAN_WGRAD3_TH") ? atoi(getenv("SRGAN_WGRAD3_TH")) : 0;
  Wgrad3Params p;
  p.x = x; p.gy = gy; p.gw = gw; p.x_bs = x_bs; p.gy_bs = gy_bs;
  p.N = N; p.CI = CI; p.CO = CO; p.H = H; p.W = W;
  p.debug = getenv("SRGAN_WGRAD3_DEBUG") ? atoi(getenv("SRGAN_WGRAD3_DEBUG")) : 0;
  p.bn_mean = bn ? bn[0] : nullptr; p.bn_inv = bn ? bn[1] : nullptr;
  p.bn_gamma = bn ? bn[2] : nullptr; p.bn_beta = bn ? bn[3] : nullptr;
  const int ci_chunks = (CI + WG3_CI - 1) / WG3_CI, co_chunks = (CO + WG3_CO - 1) / WG3_CO;
  int th = 4;                                  // 4-row tiles: fewer halo rows per pixel (2-row tiles never measured better)
  if (th_override == 2 || th_override == 4) th = th_override;
  p.tiles_x = (W + WG3_TW - 1) / WG3_TW;
  p.tiles_y = (H + th - 1) / th;
  const int64_t tiles = (int64_t)N * p.tiles_y * p.tiles_x;
  const bool ragged = W % 4 != 0 || (((uintptr_t)x | (uintptr_t)gy) & 15) != 0 || x_bs % 4 != 0 || gy_bs % 4 != 0;
  SRGAN_REQUIRE(!(ragged && precision), SRGAN_EUNSUPPORTED, "conv3x3 wgrad: mixed precision needs 16-byte rows");
  SRGAN_REQUIRE(tiles < (int64_t)1 << 31 && ci_chunks <= 65535 && co_chunks <= 65535, SRGAN_ERANGE, "conv3x3 wgrad grid");
  p.tiles = (int)tiles;
  const int walkers = conv3x3_wgrad_walkers(p.tiles, ci_chunks, co_chunks);
  if (!accumulate) if (const int status = zero_floats(gw, (int64_t)CO * CI * 9, stream)) return status;
  // several walkers per channel chunk: each keeps its 32 x 288 block in the workspace, a second kernel adds them in order
  p.partial = nullptr;
  if (walkers > 1 && !split_atomics_forced() && !(p.debug & 1))
    p.partial = partial_workspace((size_t)ci_chunks * co_chunks * walkers * (WG3_CO * WG3_CI * 9) * sizeof(float), stream);
  const int64_t blocks = (int64_t)((walkers + 7) / 8) * 8 * ci_chunks * co_chunks;
  SRGAN_REQUIRE(blocks < ((int64_t)1 << 31), SRGAN_ERANGE, "conv3x3 wgrad grid");
  dim3 grid((unsigned)blocks, 1, 1);
  const int profile_slot = profile_bracket_begin(stream);
#define SRGAN_WG3_LAUNCH(...) hipLaunchKernelGGL((conv3x3_wgrad_kernel<__VA_ARGS__>), grid, dim3(WG3_THREADS), 0, stream, p, walkers, \
                                                 ci_chunks, co_chunks)
  if (precision == 1) SRGAN_WG3_LAUNCH(4, 1);
  else if (precision == 2) SRGAN_WG3_LAUNCH(4, 2);
  else if (ragged) SRGAN_WG3_LAUNCH(4, 0, true);
  else if (th == 4) SRGAN_WG3_LAUNCH(4);
  else SRGAN_WG3_LAUNCH(2);
#undef SRGAN_WG3_LAUNCH
  if (p.partial)
    hipLaunchKernelGGL(conv3x3_wgrad_finish_kernel, dim3(WG3_FINISH_SLABS, (unsigned)(ci_chunks * co_chunks)), dim3(256), 0, stream, p,
                       walkers, ci_chunks);
  const int status = launch_status();
  profile_bracket_end(profile_slot, stream, CO, (int64_t)CI * 9, (int64_t)N * H * W, 4, th, WG3_TW, walkers, 0, 0, 0, precision);
  return status;
}

// One entry of a grouped launch's table (see pointwise_wgrad_group_plan); the weight gradient is ACCUMULATED into gw.
int conv3x3_wgrad_group_plan(int64_t x_off, int64_t x_bs, int64_t gy_off, int64_t gy_bs, float* gw, int64_t gw_off, int32_t N, int32_t CI,
                             int32_t CO, int32_t H, int32_t W, const float* const* bn, int32_t group, int64_t partial_offset, void* job_out,
                             int32_t* grid_x, int32_t* grid_y, int32_t* ragged, int64_t* partial_floats) {
  Wgrad3Job job;
  job.x_off = x_off; job.gy_off = gy_off; job.x_bs = x_bs; job.gy_bs = gy_bs;
  job.gw = gw ? gw : reinterpret_cast<float*>((intptr_t)gw_off);
  job.bn_mean = bn ? bn[0] : nullptr; job.bn_inv = bn ? bn[1] : nullptr;
  job.bn_gamma = bn ? bn[2] : nullptr; job.bn_beta = bn ? bn[3] : nullptr;
  job.N = N; job.CI = CI; job.CO = CO; job.H = H; job.W = W;
  job.ci_chunks = (CI + WG3_CI - 1) / WG3_CI; job.co_chunks = (CO + WG3_CO - 1) / WG3_CO;
  job.tiles_x = (W + WG3_TW - 1) / WG3_TW; job.tiles_y = (H + 3) / 4;
  const int64_t tiles = (int64_t)N * job.tiles_y * job.tiles_x;
  SRGAN_REQUIRE(tiles < (int64_t)1 << 31 && (int64_t)job.ci_chunks * job.co_chunks <= 65535, SRGAN_ERANGE, "conv3x3 wgrad grid");
  job.tiles = (int)tiles;
  job.walkers = conv3x3_wgrad_walkers(job.tiles, job.ci_chunks, job.co_chunks, group);
  job.pad = 0;
  job.partial_off = partial_offset;
  *partial_floats = (int64_t)job.walkers * job.ci_chunks * job.co_chunks * (WG3_CO * WG3_CI * 9);
  static_assert(sizeof(Wgrad3Job) <= 128, "job slot");
  memset(job_out, 0, 128);
  memcpy(job_out, &job, sizeof(job));
  *grid_x = job.walkers; *grid_y = job.ci_chunks * job.co_chunks;
  *ragged = (W % 4 != 0 || x_bs % 4 != 0 || gy_bs % 4 != 0 || x_off % 4 != 0 || gy_off % 4 != 0) ? 1 : 0;
  return SRGAN_OK;
}

int conv3x3_wgrad_group_run(const void* jobs, int32_t count, int32_t grid_x, int32_t grid_y, int32_t ragged,
                            const float* x_base, const float* gy_base, float* gw_base, int64_t flops_mn, int64_t pixels,
                            int64_t elements, int64_t partial_floats, hipStream_t stream) {
  SRGAN_REQUIRE(count >= 1 && count <= 65535 && grid_y <= 65535, SRGAN_ERANGE, "grouped conv3x3 wgrad grid");
  float* partial_base = nullptr;          // the ordered form (see pointwise_wgrad_group_run)
  if (partial_floats > 0 && grid_x > 1 && !split_atomics_forced())
    partial_base = partial_workspace((size_t)partial_floats * sizeof(float), stream);
  const bool rag = ragged || ((((uintptr_t)x_base | (uintptr_t)gy_base) & 15) != 0);
  // one-dimensional: 8 XCDs x (units per XCD, rounded up) x (WG3_UNIT walkers x channel chunks) -- see the kernel
  const int64_t units = (int64_t)((grid_x + WG3_UNIT - 1) / WG3_UNIT) * count, rounds = (units + 7) / 8;
  SRGAN_REQUIRE(rounds * 8 * WG3_UNIT * grid_y < ((int64_t)1 << 31), SRGAN_ERANGE, "grouped conv3x3 wgrad grid");
  dim3 grid((unsigned)(rounds * 8 * WG3_UNIT * grid_y), 1, 1);
  const int profile_slot = profile_bracket_begin(stream);
  if (rag) hipLaunchKernelGGL((conv3x3_wgrad_grouped_kernel<true>), grid, dim3(WG3_THREADS), 0, stream,
                              reinterpret_cast<const Wgrad3Job*>(jobs), x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base);
  else hipLaunchKernelGGL((conv3x3_wgrad_grouped_kernel<false>), grid, dim3(WG3_THREADS), 0, stream,
                          reinterpret_cast<const Wgrad3Job*>(jobs), x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base);
  if (partial_base)
    hipLaunchKernelGGL(conv3x3_wgrad_grouped_finish_kernel, dim3(WG3_FINISH_SLABS, (unsigned)grid_y, (unsigned)count), dim3(256), 0, stream,
                       reinterpret_cast<const Wgrad3Job*>(jobs), gw_base, partial_base);
  const int status = launch_status();
  profile_bracket_end(profile_slot, stream, 1, flops_mn, pixels, 4, 4, WG3_TW, grid_x, 0, 0, elements > pixels ? elements - pixels : 0);
  return status;
}

}  // namespace srgan
