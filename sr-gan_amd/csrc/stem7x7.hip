// stem7x7.hip -- the DenseNet stem convolution (3 -> 64 channels, 7x7 / stride 2 / pad 3, reference
// crowd/models.py:1072 `conv0`) forward and its weight gradient.
//
// On the generic gather-GEMM these two passes ran at 44 TF/s (K = 3 * 49 = 147: every B element is gathered with its own
// address decode and bounds test, 49 times per input pixel): 3.6 + 2.7 ms per training step at 512 x 512.  Here a
// workgroup walks output tiles of 4 rows x 32 columns persistently; per tile it stages the (2 * 4 + 5) x (2 * 32 + 6)
// x 3 input patch in LDS once (each input pixel is read from HBM / L2 once for all 49 taps and 64 output channels) and
// the inner loops are ds_read_b32 at per-lane base + immediate offsets feeding v_mfma_f32_32x32x2_f32:
//   forward:  rows = output channels (weights [k][co] resident in LDS for the whole walk), columns = 32 output pixels of
//             one row, k-pair = kernel columns (kw, kw + 1) of one (channel, kernel row): 84 steps (kw = 7 is a zero pad);
//   weight gradient: rows = output channels (the gy tile staged in LDS), columns = the 147 (+13 pad) weight positions
//             (each lane owns one (c, kh, kw) = one patch offset), k-pair = two adjacent output pixels; every wave keeps
//             its 2 x 5 accumulator tiles over the whole walk, the four waves are summed through LDS and leave with one
//             atomic pass per workgroup.
// Roofline: fp32 MFMA (157.3 TF/s): 2 * 147 * 64 FLOP per output pixel, against 4 * (3 * 4 + 64) B of HBM traffic.
#include "common.h"
#include "split_finish.h"
#include <stdlib.h>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StemParams {
  const float* x;      // [N, 3, H, W], batch stride x_bs
  const float* w;      // [K, 3, 7, 7]
  float* y;            // forward: [N, K, OH, OW] (batch stride y_bs); weight gradient: gy, read
  float* gw;           // weight gradient: [K, 3, 7, 7], accumulated with atomics
  int64_t x_bs, y_bs;
  int32_t N, H, W, K, OH, OW;
  int32_t tiles_x, tiles_y, tiles;
};

constexpr int ST_TH = 4, ST_TW = 32;                 // output tile
constexpr int ST_PH = 2 * ST_TH + 5, ST_PW = 2 * ST_TW + 6;     // input patch 13 x 70 (column 69 only meets the zero pad tap)
constexpr int ST_PATCH = 3 * ST_PH * ST_PW;          // 2730 floats
constexpr int ST_NP = (ST_PATCH + 255) / 256;        // 11 per thread
constexpr int ST_KSTEPS = 3 * 7 * 4;                 // (c, kh, kw pair)

__device__ __forceinline__ void stem_tile_origin(const StemParams& p, int tile, int& n, int& oh0, int& ow0) {
  const int tx = tile % p.tiles_x;
  const int rest = tile / p.tiles_x;
  const int ty = rest % p.tiles_y;
  n = rest / p.tiles_y;
  oh0 = ty * ST_TH; ow0 = tx * ST_TW;
}

// Raw loads of one patch into registers (clamped addresses; validity bits applied when LDS is written).
__device__ __forceinline__ uint32_t stem_fetch_patch(const StemParams& p, int tile, int tid, float (&r)[ST_NP]) {
  int n, oh0, ow0;
  stem_tile_origin(p, tile, n, oh0, ow0);
  const float* xn = p.x + (int64_t)n * p.x_bs;
  uint32_t ok_bits = 0;
#pragma unroll
  for (int e = 0; e < ST_NP; ++e) {
    const int flat = e * 256 + tid;
    const int c = flat / (ST_PH * ST_PW), rem = flat - c * (ST_PH * ST_PW);
    const int py = rem / ST_PW, px = rem - py * ST_PW;
    const int ih = 2 * oh0 - 3 + py, iw = 2 * ow0 - 3 + px;
    const bool ok = flat < ST_PATCH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    ok_bits |= (ok ? 1u : 0u) << e;
    r[e] = xn[ok ? (int64_t)c * p.H * p.W + (int64_t)ih * p.W + iw : 0];
  }
  return ok_bits;
}

__device__ __forceinline__ void stem_store_patch(float* patch, int tid, const float (&r)[ST_NP], uint32_t ok_bits) {
#pragma unroll
  for (int e = 0; e < ST_NP; ++e) {
    const int flat = e * 256 + tid;
    if (flat < ST_PATCH) patch[flat] = ((ok_bits >> e) & 1u) ? r[e] : 0.f;
  }
}

template <int MI>
__global__ __launch_bounds__(256, 2) void stem7x7_fwd_kernel(const StemParams p) {
  constexpr int BM = MI * 32, LDA = BM + 1;
  __shared__ float wt[2 * ST_KSTEPS * LDA];          // [k = (c, kh, kw)][co], kw = 7 rows are zero
  __shared__ float patches[2 * ST_PATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;

  for (int idx = tid; idx < 2 * ST_KSTEPS * BM; idx += 256) {         // weights once per workgroup
    const int co = idx % BM, k = idx / BM;                            // k = ((c * 7 + kh) * 8 + kw)
    const int kw = k & 7, ckh = k >> 3;
    const bool ok = kw < 7 && co < p.K;
    wt[k * LDA + co] = ok ? p.w[(int64_t)co * 147 + ckh * 7 + kw] : 0.f;
  }
  float r[ST_NP];
  int tile = (int)blockIdx.x;
  uint32_t ok_bits = 0;
  if (tile < p.tiles) ok_bits = stem_fetch_patch(p, tile, tid, r);
  int cur = 0;
  if (tile < p.tiles) stem_store_patch(patches, tid, r, ok_bits);
  __syncthreads();
  for (; tile < p.tiles; tile += (int)gridDim.x) {
    const int next = tile + (int)gridDim.x;
    if (next < p.tiles) ok_bits = stem_fetch_patch(p, next, tid, r);
    const float* patch = patches + cur * ST_PATCH;
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][q] = 0.f;
    const float* a_base = wt + lhi * LDA + l31;
    const float* b_base = patch + (2 * wave) * ST_PW + 2 * l31 + lhi;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
          const int k = ((c * 7 + kh) * 8 + 2 * kp);
          const float b = b_base[c * (ST_PH * ST_PW) + kh * ST_PW + 2 * kp];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_base[k * LDA + mi * 32], b, acc[mi], 0, 0, 0);
        }
      }
    }
    int n, oh0, ow0;
    stem_tile_origin(p, tile, n, oh0, ow0);
    const int oh = oh0 + wave, ow = ow0 + l31;
    if (oh < p.OH && ow < p.OW) {
      float* yn = p.y + (int64_t)n * p.y_bs + (int64_t)oh * p.OW + ow;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int co = mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
          if (co < p.K) __builtin_nontemporal_store(acc[mi][q], yn + (int64_t)co * p.OH * p.OW);
        }
    }
    if (next < p.tiles) stem_store_patch(patches + (cur ^ 1) * ST_PATCH, tid, r, ok_bits);   // nobody reads the other buffer now
    __syncthreads();
    cur ^= 1;
  }
}

// gw[co, c, kh, kw] += sum_{n, oh, ow} gy[n, co, oh, ow] * x[n, c, 2 oh - 3 + kh, 2 ow - 3 + kw]
constexpr int ST_NJ = 5;                             // 5 column blocks of 32 cover the 147 weight positions
// (32 output channels per workgroup, blockIdx.y selects the 32-channel block: with 64 the 2 x 5 accumulator tiles plus the
// staged registers spill)
__global__ __launch_bounds__(256, 2) void stem7x7_wgrad_kernel(const StemParams p, float* __restrict__ partial) {
  constexpr int MI = 1;
  constexpr int BM = MI * 32, LDG = ST_TH * ST_TW + 1;
  const int co0 = (int)blockIdx.y * BM;
  constexpr int NG = BM * ST_TH * ST_TW / 256;       // gy elements staged per thread
  constexpr int LDR = ST_NJ * 32 + 1;                // row stride of the final [co][weight position] reduction tile
  constexpr int SMEM = BM * LDG + ST_PATCH > BM * LDR ? BM * LDG + ST_PATCH : BM * LDR;
  __shared__ float smem[SMEM];
  float* gs = smem;                                  // gy tile [co][4 x 32 pixels]
  float* patch = smem + BM * LDG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;

  int joff[ST_NJ];                                   // patch offset of this lane's weight position in every column block
  bool jok[ST_NJ];
#pragma unroll
  for (int nj = 0; nj < ST_NJ; ++nj) {
    const int j = nj * 32 + l31;
    jok[nj] = j < 147;
    const int c = j / 49, rem = j - c * 49, kh = rem / 7, kw = rem - kh * 7;
    joff[nj] = jok[nj] ? c * (ST_PH * ST_PW) + kh * ST_PW + kw : 0;
  }
  f32x16 acc[MI][ST_NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < ST_NJ; ++nj)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][nj][q] = 0.f;

  float r[ST_NP], g[NG];
  uint32_t ok_bits = 0, g_ok = 0;
  auto fetch_gy = [&](int tile) {
    int n, oh0, ow0;
    stem_tile_origin(p, tile, n, oh0, ow0);
    const float* gn = p.y + (int64_t)n * p.y_bs;
    g_ok = 0;
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int flat = e * 256 + tid;
      const int co = flat / (ST_TH * ST_TW), rem = flat - co * (ST_TH * ST_TW);
      const int oh = oh0 + rem / ST_TW, ow = ow0 + rem % ST_TW;
      const bool ok = co0 + co < p.K && oh < p.OH && ow < p.OW;
      g_ok |= (ok ? 1u : 0u) << e;
      g[e] = gn[ok ? (int64_t)(co0 + co) * p.OH * p.OW + (int64_t)oh * p.OW + ow : 0];
    }
  };
  int tile = (int)blockIdx.x;
  if (tile < p.tiles) { ok_bits = stem_fetch_patch(p, tile, tid, r); fetch_gy(tile); }
  for (; tile < p.tiles; tile += (int)gridDim.x) {
    __syncthreads();                                 // the previous tile's reads are done
    stem_store_patch(patch, tid, r, ok_bits);
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int flat = e * 256 + tid;
      const int co = flat / (ST_TH * ST_TW), rem = flat - co * (ST_TH * ST_TW);
      gs[co * LDG + rem] = ((g_ok >> e) & 1u) ? g[e] : 0.f;
    }
    __syncthreads();
    const int next = tile + (int)gridDim.x;
    if (next < p.tiles) { ok_bits = stem_fetch_patch(p, next, tid, r); fetch_gy(next); }
    // this wave's output row: 16 pixel pairs
    const float* a_base = gs + l31 * LDG + wave * ST_TW + lhi;
    const float* b_base = patch + (2 * wave) * ST_PW + 2 * lhi;
#pragma unroll 2
    for (int s = 0; s < ST_TW / 2; ++s) {
      float a[MI], b[ST_NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = a_base[mi * 32 * LDG + 2 * s];
#pragma unroll
      for (int nj = 0; nj < ST_NJ; ++nj) {
        const float v = b_base[joff[nj] + 4 * s];
        b[nj] = jok[nj] ? v : 0.f;
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < ST_NJ; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
  }
  // The four waves hold partial sums over different pixel rows of the same [co][weight position] tile: they are summed
  // through LDS (one wave at a time into the same slots), then all 256 threads add the tile to gw with lanes along the
  // contiguous weight positions -- one atomic per output element and workgroup.
  float* red = smem;
  for (int turn = 0; turn < 4; ++turn) {
    __syncthreads();
    if (wave == turn) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < ST_NJ; ++nj)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* slot = red + (mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi) * LDR + nj * 32 + l31;
            *slot = turn == 0 ? acc[mi][nj][q] : *slot + acc[mi][nj][q];
          }
    }
  }
  __syncthreads();
  if (partial) {        // round 5, ordered: this walker's [32][147] block goes to the workspace; stem7x7_wgrad_finish_kernel adds
    float* mine = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (BM * 147);      // the walkers in walker order
    for (int idx = tid; idx < BM * 147; idx += 256) mine[idx] = red[(idx / 147) * LDR + idx % 147];
    return;
  }
  for (int idx = tid; idx < BM * 147; idx += 256) {
    const int co = idx / 147, j = idx - co * 147;
    if (co0 + co < p.K) unsafeAtomicAdd(p.gw + (int64_t)(co0 + co) * 147 + j, red[co * LDR + j]);
  }
}

// gw[32-channel block] += its walkers' blocks, added in walker order (blockIdx.y = channel block)
__global__ __launch_bounds__(256) void stem7x7_wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ gw, int K,
                                                                   int walkers) {
  constexpr int BLOCK = 32 * 147;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (idx >= BLOCK) return;
  const int co = (int)blockIdx.y * 32 + idx / 147;
  if (co >= K) return;
  const float* mine = partial + (int64_t)blockIdx.y * walkers * BLOCK + idx;
  float total = mine[0];
  for (int w = 1; w < walkers; ++w) total += mine[(int64_t)w * BLOCK];
  gw[(int64_t)co * 147 + idx % 147] += total;
}

// gx[n, c, ih, iw] = sum_{co, kh, kw} gy[n, co, (ih + 3 - kh) / 2, (iw + 3 - kw) / 2] * w[co, c, kh, kw]   (exact divisions only)
// Three output channels: the matrix pipe would be 3/32 used (the few-rows VALU kernel ran this at 16 TF/s, 1.2 ms per
// pass).  Here a lane owns one 2x2 block of input pixels (all four stride-2 classes) x 3 channels = 12 accumulators and
// the work is arranged so that it is VALU-bound: per output channel co the 16 shifted gy values a block needs
// ((dh, dw) in [-1, 2]^2) are read from an LDS patch -- lanes along the row: conflict-free -- and feed all 49 x 3 taps,
// whose weights are WAVE-UNIFORM (every lane handles the same classes) and therefore scalar operands: 16 LDS reads per
// 147 FMAs.  A workgroup covers 4 block rows x 64 block columns (one row per wave) and walks the output channels in chunks
// of 16 with the next chunk's patch in flight.
constexpr int SB_CC = 16, SB_RQ = 4, SB_CQ = 64;
constexpr int SB_PH = SB_RQ + 3, SB_PW = SB_CQ + 3, SB_PWS = SB_PW + 1;
constexpr int SB_PATCH = SB_CC * SB_PH * SB_PWS;
constexpr int SB_NP = (SB_CC * SB_PH * SB_PW + 255) / 256;

__global__ __launch_bounds__(256, 2) void stem7x7_bwd_data_kernel(const StemParams p) {
  __shared__ float gys[2 * SB_PATCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int block = blockIdx.x;
  const int tx = block % p.tiles_x; block /= p.tiles_x;
  const int ty = block % p.tiles_y;
  const int n = block / p.tiles_y;
  const int qh0 = ty * SB_RQ, qw0 = tx * SB_CQ;
  const float* gn = p.y + (int64_t)n * p.y_bs;
  const int64_t plane = (int64_t)p.OH * p.OW;

  // staging coordinates of this thread inside a chunk: (channel, patch row, patch column), lanes along the row
  int soff[SB_NP], sdst[SB_NP];
#pragma unroll
  for (int e = 0; e < SB_NP; ++e) {
    const int flat = e * 256 + tid;
    const int c = flat / (SB_PH * SB_PW), rem = flat - c * (SB_PH * SB_PW);
    const int ry = rem / SB_PW, rx = rem - ry * SB_PW;
    const int oh = qh0 - 1 + ry, ow = qw0 - 1 + rx;
    const bool ok = flat < SB_CC * SB_PH * SB_PW && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
    soff[e] = ok ? (int)(c * plane + (int64_t)oh * p.OW + ow) : -1;
    sdst[e] = flat < SB_CC * SB_PH * SB_PW ? (c * SB_PH + ry) * SB_PWS + rx : -1;
  }
  float r[SB_NP];
  auto fetch = [&](int co0) {
#pragma unroll
    for (int e = 0; e < SB_NP; ++e) {
      const int c = (e * 256 + tid) / (SB_PH * SB_PW);
      const bool ok = soff[e] >= 0 && co0 + c < p.K;
      r[e] = gn[ok ? (int64_t)co0 * plane + soff[e] : 0];
    }
  };
  auto stage = [&](int co0, float* dst) {
#pragma unroll
    for (int e = 0; e < SB_NP; ++e) {
      const int c = (e * 256 + tid) / (SB_PH * SB_PW);
      if (sdst[e] >= 0) dst[sdst[e]] = (soff[e] >= 0 && co0 + c < p.K) ? r[e] : 0.f;
    }
  };

  float acc[2][2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[a][b][c] = 0.f;

  fetch(0);
  stage(0, gys);
  __syncthreads();
  int cur = 0;
  for (int co0 = 0; co0 < p.K; co0 += SB_CC) {
    const bool more = co0 + SB_CC < p.K;
    if (more) fetch(co0 + SB_CC);
    const float* patch = gys + cur * SB_PATCH + wave * SB_PWS + lane;
#pragma unroll 2
    for (int cl = 0; cl < SB_CC; ++cl) {
      if (co0 + cl >= p.K) break;
      const float* __restrict__ wc = p.w + (int64_t)(co0 + cl) * 147;      // wave-uniform: scalar loads
      float g[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = patch[(cl * SB_PH + i) * SB_PWS + j];
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
        const int a = (kh + 1) & 1;                  // ih + 3 - kh even  <=>  ih parity = (kh + 1) & 1
        const int dh = (a + 3 - kh) / 2;             // oh = qh + dh, dh in [-1, 2]
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          const int b = (kw + 1) & 1;
          const int dw = (b + 3 - kw) / 2;
          const float v = g[dh + 1][dw + 1];
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[a][b][c] = fmaf(v, wc[c * 49 + kh * 7 + kw], acc[a][b][c]);
        }
      }
    }
    if (more) stage(co0 + SB_CC, gys + (cur ^ 1) * SB_PATCH);
    __syncthreads();
    cur ^= 1;
  }
  const int qh = qh0 + wave, qw = qw0 + lane;
  float* xn = p.gw + (int64_t)n * p.x_bs;            // (gw carries the gx pointer in this pass)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ih = 2 * qh + a, iw = 2 * qw;
      if (ih >= p.H || iw >= p.W) continue;
      float* dst = xn + (int64_t)c * p.H * p.W + (int64_t)ih * p.W + iw;
      if (iw + 1 < p.W && (p.W & 1) == 0) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f pair; pair.x = acc[a][0][c]; pair.y = acc[a][1][c];
        __builtin_nontemporal_store(pair, reinterpret_cast<v2f*>(dst));
      } else {
        dst[0] = acc[a][0][c];
        if (iw + 1 < p.W) dst[1] = acc[a][1][c];
      }
    }
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

bool stem7x7_enabled() {
  static const bool disabled = getenv("SRGAN_NO_STEM") != nullptr;
  return !disabled;
}

// 7x7 / stride 2 / pad 3 on 3 input channels with at most 64 output channels (the DenseNet stem).
bool stem7x7_geometry(int32_t C, int32_t K, int32_t R, int32_t S, int32_t sh, int32_t sw, int32_t ph, int32_t pw) {
  return C == 3 && K >= 8 && K <= 64 && R == 7 && S == 7 && sh == 2 && sw == 2 && ph == 3 && pw == 3;
}

static void stem_fill(StemParams& p, const float* x, int64_t x_bs, const float* w, float* y, int64_t y_bs, float* gw, int32_t N,
                      int32_t H, int32_t W, int32_t K, int32_t OH, int32_t OW) {
  p.x = x; p.w = w; p.y = y; p.gw = gw; p.x_bs = x_bs; p.y_bs = y_bs;
  p.N = N; p.H = H; p.W = W; p.K = K; p.OH = OH; p.OW = OW;
  p.tiles_x = (OW + ST_TW - 1) / ST_TW; p.tiles_y = (OH + ST_TH - 1) / ST_TH;
  p.tiles = N * p.tiles_y * p.tiles_x;
}

int stem7x7_fwd_run(const float* x, int64_t x_bs, const float* w, float* y, int64_t y_bs, int32_t N, int32_t H, int32_t W,
                    int32_t K, int32_t OH, int32_t OW, hipStream_t stream) {
  StemParams p;
  stem_fill(p, x, x_bs, w, y, y_bs, nullptr, N, H, W, K, OH, OW);
  SRGAN_REQUIRE((int64_t)N * p.tiles_y * p.tiles_x < ((int64_t)1 << 31), SRGAN_ERANGE, "stem grid");
  static const int resident = getenv("SRGAN_STEM_WGS") ? atoi(getenv("SRGAN_STEM_WGS")) : 1024;
  const int grid = p.tiles < resident ? p.tiles : resident;
  const int slot = profile_bracket_begin(stream);
  if (K > 32) hipLaunchKernelGGL(stem7x7_fwd_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(stem7x7_fwd_kernel<1>, dim3(grid), dim3(256), 0, stream, p);
  const int status = launch_status();
  profile_bracket_end(slot, stream, K, (int64_t)N * OH * OW, 147, 10, K > 32 ? 64 : 32, 128, 1, 0, 0, (int64_t)N * 3 * H * W);
  return status;
}

int stem7x7_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t H, int32_t W,
                      int32_t K, int32_t OH, int32_t OW, int accumulate, hipStream_t stream) {
  StemParams p;
  stem_fill(p, x, x_bs, nullptr, const_cast<float*>(gy), gy_bs, gw, N, H, W, K, OH, OW);
  SRGAN_REQUIRE((int64_t)N * p.tiles_y * p.tiles_x < ((int64_t)1 << 31), SRGAN_ERANGE, "stem grid");
  if (!accumulate) if (const int status = zero_floats(gw, (int64_t)K * 147, stream)) return status;
  static const int resident = getenv("SRGAN_STEM_WGRAD_WGS") ? atoi(getenv("SRGAN_STEM_WGRAD_WGS")) : 512;
  int grid = p.tiles < resident ? p.tiles : resident;
  const int slot = profile_bracket_begin(stream);
  const int co_blocks = (K + 31) / 32;
  // ordered form: half the walkers (each keeps a 32 x 147 block in the workspace that the finish kernel re-reads)
  const bool ordered = grid > 1 && !split_atomics_forced() && partial_workspace(1, stream) != nullptr;
  if (ordered && grid > 256) grid = 256;
  float* partial = ordered ? partial_workspace((size_t)co_blocks * grid * 32 * 147 * sizeof(float), stream) : nullptr;
  hipLaunchKernelGGL(stem7x7_wgrad_kernel, dim3(grid, co_blocks), dim3(256), 0, stream, p, partial);
  if (partial)
    hipLaunchKernelGGL(stem7x7_wgrad_finish_kernel, dim3((32 * 147 + 255) / 256, co_blocks), dim3(256), 0, stream, partial, gw, K, grid);
  const int status = launch_status();
  profile_bracket_end(slot, stream, K, 147, (int64_t)N * OH * OW, 11, 32, 160, grid, 0, 0, (int64_t)N * 3 * H * W);
  return status;
}

int stem7x7_bwd_data_run(const float* gy, int64_t gy_bs, const float* w, float* gx, int64_t gx_bs, int32_t N, int32_t H,
                         int32_t W, int32_t K, int32_t OH, int32_t OW, hipStream_t stream) {
  StemParams p;
  stem_fill(p, nullptr, gx_bs, w, const_cast<float*>(gy), gy_bs, gx, N, H, W, K, OH, OW);
  SRGAN_REQUIRE((gx_bs & 1) == 0 && (((uintptr_t)gx) & 7) == 0, SRGAN_EINVAL, "stem data gradient: 8-byte aligned rows");
  p.tiles_x = ((W + 1) / 2 + SB_CQ - 1) / SB_CQ;
  p.tiles_y = ((H + 1) / 2 + SB_RQ - 1) / SB_RQ;
  const int64_t blocks = (int64_t)N * p.tiles_y * p.tiles_x;
  SRGAN_REQUIRE(blocks < ((int64_t)1 << 31), SRGAN_ERANGE, "stem grid");
  const int slot = profile_bracket_begin(stream);
  hipLaunchKernelGGL(stem7x7_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  const int status = launch_status();
  // logical work: 3 x (N * H * W) outputs, K * 49 / 4 taps each on average
  profile_bracket_end(slot, stream, 3, (int64_t)N * H * W, (int64_t)K * 49 / 4, 12, 4, 256, 1, 0, 0, (int64_t)N * K * OH * OW);
  return status;
}

}  // namespace srgan
