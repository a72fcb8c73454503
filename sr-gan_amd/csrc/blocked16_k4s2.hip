// blocked16_k4s2.hip -- the 4x4 / stride 2 / pad 1 convolution family on the 16-bit data path: the DCGAN discriminator's
// `leaky_relu(conv)` stages and the generator's transposed convolutions (reference age/models.py:37-51,61-73 = driving/models.py,
// crowd/models.py:132-146), all three passes.
//
// A stride-2 4x4 window is a 2x2 window over a space-to-depth view: with in'[(c, qy, qx)][Y][X] = in[c][2Y - 1 + qy][2X - 1 + qx],
//   out[o][y][x] = sum_{a, b in {0, 1}} sum_{c, qy, qx} w[o][c][2a + qy][2b + qx] * in'[(c, qy, qx)][y + a][x + b]
// ("down": the forward convolution; also the data gradient of a transposed convolution), and the transposed direction splits by
// the parity (ry, rx) of the output pixel into four 2x2-window convolutions over the small plane
//   out[c][2t + ry][2s + rx] = sum_{a, b} sum_k w[k][c][kh(ry, a)][kw(rx, b)] * in[k][t + a + oy(ry)][s + b + ox(rx)]
// with kh(0, a) = 3 - 2a, oy(0) = -1; kh(1, a) = 2 - 2a, oy(1) = 0 ("up": the generator's forward; the data gradient of the
// strided convolution).  No multiply-accumulate is spent on taps that cannot align.  Both are ONE kernel: a 2x2-tap implicit
// GEMM in the blocked layout whose halo gather -- stride 2 with the parity offset for "down", stride 1 for "up" -- lives entirely
// in the per-lane source addresses of an LDS-DMA ring (see hconv3x3_dma_kernel in blocked16.hip, whose structure this shares), and
// whose stores go to (y * dsy + doy, x * dsx + dox).  A chunk is CK = 4 k-slots (32 reduced channels: one channel group with
// its four parities for "down", four groups for "up") x 4 taps = 32 MFMAs per wave at a 64 x 256 tile.
//
//   hconv2x2_kernel        both directions, epilogues as the 3x3 kernel (bias + leaky, mask by reference)
//   hwgrad4x4s2_kernel     gw[k][c][4][4] += gy (x) in over pixels: wave = one parity (qy, qx) = four of the sixteen taps,
//                          transpose-read fragments, partial blocks + ordered finish
//   h_pack_k4s2_weights    the operand shadows: "down" [chunk = group][tap][parity][rows], "up" [class][chunk][tap][group][rows]
#include <type_traits>
#include <atomic>
#include <stdlib.h>
#include "blocked16.h"
#include "split_finish.h"

namespace srgan {

float* partial_workspace(size_t bytes, hipStream_t stream);


// (G = channels per slot: 8, or 4 for the fp32 form)
// mode 0 ("down"): slot (chunk, tap = 2a + b, j = 2 qy + qx, o) = G reduced channels G chunk .. of w[o][.][2a + qy][2b + qx]
// mode 1 + 2 ry + rx ("up", one output parity class): slot (chunk, tap, j, o) = 8 reduced channels 8 (4 chunk + j) .. of
//   w[.][o][kh(ry, a)][kw(rx, b)]
// element (row o, reduced r, kh, kw) at w[o * row_stride + r * reduced_stride + kh * 4 + kw]
template <int PREC>
__global__ __launch_bounds__(256) void h_pack_k4s2_weights_kernel(const float* __restrict__ w, Slot* __restrict__ packed,
                                                                  int64_t slots, int32_t rows, int32_t reduced, int64_t row_stride,
                                                                  int64_t reduced_stride, int32_t mode) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= slots) return;
  h_pack_k4s2_weights_slot<PREC>(w, packed, slot, rows, reduced, row_stride, reduced_stride, mode);
}

struct HConv2Params {
  const Slot* in; const Slot* wp; Slot* out; const float* bias; const Slot* ref;
  float slope; int32_t epi;
  int32_t N, CGI, IH, IW;             // input: channel groups, plane
  int32_t down;                       // 1: space-to-depth gather (stride 2, parity offsets); 0: plain 2x2 window
  int32_t oy, ox;                     // window origin (down: -1, -1; up: -1 or 0 per class)
  int32_t GH, GW;                     // the grid of output positions the tiles walk
  int32_t CGO, OH, OW, dsy, dsx, doy, dox;   // output: groups, plane, placement of position (y, x)
  int32_t CO, C_real;                 // rows of the packed operand, output channels that exist (bias entries)
  int32_t chunks;
  int32_t tiles_x, tiles_y, tiles_m;
  int32_t xcd_remap;
  int64_t class_stride;               // "up": blockIdx.y = the output parity class 2 ry + rx; its operand starts class_stride slots further
};

template <int BM, int NI, int TW, int ROWS, int PREC>
__global__ __launch_bounds__(256, 2) void hconv2x2_kernel(const HConv2Params pp, const Slot* zero) {
  // the four output parity classes of the "up" direction are ONE launch (gridDim.y = 4): a class alone is a quarter of the
  // convolution and leaves the chip under-filled on the small planes (192 workgroups on a 4 x 12 plane at batch 128)
  HConv2Params p = pp;
  if (!p.down) {
    const int ry = (int)blockIdx.y >> 1, rx = (int)blockIdx.y & 1;
    p.oy = ry ? 0 : -1; p.ox = rx ? 0 : -1; p.doy = ry; p.dox = rx;
    p.wp += (int64_t)blockIdx.y * p.class_stride;
  }
  constexpr int RING = 2, CK = K4_CK, TAPS = K4_TAPS;
  constexpr int P = 128 * NI, IMG = P / (ROWS * TW), PW = TW + 1, PH = ROWS + 1, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int MI = BM / 32;
  constexpr int PATCH_G = IMG * PLANE, PATCH_Q = CK * PATCH_G, WT_Q = TAPS * CK * BM;
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
  // the tile's bias values in LDS, staged here and read in the epilogue (blocked16.hip, hconv3_stage_bias: per value from global
  // memory they were 4 dependent loads per quad of the epilogue)
  __shared__ __attribute__((aligned(16))) float bias_rows[BM];
  const bool with_bias = p.epi == 1 && p.bias != nullptr;
  if (with_bias) {
    if ((int)threadIdx.x < BM) bias_rows[threadIdx.x] = m0 + (int)threadIdx.x < p.C_real ? p.bias[m0 + threadIdx.x] : 0.f;
    __syncthreads();
  }
  const int IHW = p.IH * p.IW;
  const uint32_t lds0 = h_lds_address(ring);

  // this lane's source of instruction wave + 4 e (see hconv3x3_dma_kernel): kind = the k-slot j of a patch slot, 4 weights,
  // 5 zeros; off in slots from the image-n0 input (+ chunk stride) or from the packed operand
  int off[L], kind[L];
#pragma unroll
  for (int e = 0; e < L; ++e) {
    const int i = wave + 4 * e;
    off[e] = 0; kind[e] = 5;
    if (i < PATCH_I) {
      const int flat = i * 64 + lane;
      const int j = flat / PATCH_G, rest = flat - j * PATCH_G;
      const int img = rest / PLANE, pix = rest - img * PLANE;
      const int py = pix / PW, px = pix - py * PW;
      int sy, sx, group;
      if (p.down) { sy = 2 * (y0 + py) + p.oy + (j >> 1); sx = 2 * (x0 + px) + p.ox + (j & 1); group = 0; }
      else { sy = y0 + py + p.oy; sx = x0 + px + p.ox; group = j; }
      const bool ok = flat < PATCH_Q && (unsigned)sy < (unsigned)p.IH && (unsigned)sx < (unsigned)p.IW && n0 + img < p.N;
      if (ok) { off[e] = ((img * p.CGI + group) * p.IH + sy) * p.IW + sx; kind[e] = j; }
    } else if (i < T) {
      const int flat = (i - PATCH_I) * 64 + lane;
      const int o = flat % BM, tj = flat / BM;
      if (m0 + o < p.CO) { off[e] = tj * p.CO + m0 + o; kind[e] = 4; }
    }
  }
  const Slot* in_n = p.in + (int64_t)n0 * p.CGI * IHW;
  const int64_t chunk_stride = (int64_t)(p.down ? 1 : CK) * IHW;        // channel groups per chunk
  auto issue = [&](int c, int stage) {
#pragma unroll
    for (int e = 0; e < L; ++e) {
      const Slot* src = zero;
      if (kind[e] == 4) src = p.wp + ((int64_t)c * (TAPS * CK * p.CO) + off[e]);
      else if (kind[e] < 4 && (p.down ? c : CK * c + kind[e]) < p.CGI) src = in_n + (c * chunk_stride + off[e]);
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
  if (chunks > 0) issue(0, 0);
  int stage = 0;
  for (int c = 0; c < chunks; ++c) {
    h_dma_wait_and_barrier<0>();
    if (c + 1 < chunks) issue(c + 1, stage ^ 1);
    const Slot* st = ring + stage * STAGE_Q;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int s = 0; s < CK / 2; ++s) {             // k-step s: k-slots 2s (lanes 0-31) and 2s + 1 (lanes 32-63)
        Slot a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = st[a_lane + (tap * CK + 2 * s) * BM + mi * 32];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = st[b_lane[ni] + 2 * s * PATCH_G + (tap >> 1) * PW + (tap & 1)];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = h_mfma<PREC>(a[mi], b[ni], acc[mi][ni]);
      }
    }
    stage ^= 1;
  }

  // epilogue (as hconv3_epilogue, with the output placement)
  const int OHW = p.OH * p.OW;
  float4 bias4[MI][4];              // this lane's rows are the same for every pixel column: read once
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      bias4[mi][qd] = with_bias ? *reinterpret_cast<const float4*>(&bias_rows[mi * 32 + 8 * qd + 4 * lhi]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = ((tid >> 6) * NI + ni) * 32 + l31;
    const int n = n0 + q / (ROWS * TW), y = y0 + (q / TW) % ROWS, x = x0 + q % TW;
    if (n >= p.N || y >= p.GH || x >= p.GW) continue;
    const int pixel = (y * p.dsy + p.doy) * p.OW + x * p.dsx + p.dox;
    // epi 2: the mask references of this pixel column in ONE batch in front of its stores (loads and stores share the in-order
    // vmcnt: interleaved, every load waited for the stores in front of it -- blocked16.hip, hconv3_epilogue)
    using RefWord = std::conditional_t<PREC == 0, float4, uint2>;
    RefWord refs[MI][4];
    if (p.epi == 2) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int o = m0 + mi * 32 + 8 * qd + 4 * lhi;
          const int group = PREC == 0 ? o / 4 : o / 8;
          const int64_t slot = ((int64_t)n * p.CGO + (group < p.CGO ? group : 0)) * OHW + pixel;
          if constexpr (PREC == 0) refs[mi][qd] = *reinterpret_cast<const float4*>(p.ref + slot);
          else refs[mi][qd] = *(reinterpret_cast<const uint2*>(p.ref + slot) + lhi);
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if constexpr (PREC == 0) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          // registers 4 qd .. 4 qd + 3 = the four consecutive channels o .. o + 3 = a whole slot (group o / 4) of the fp32 form
          const int o = m0 + mi * 32 + 8 * qd + 4 * lhi;
          const int group = o / 4;
          if (group >= p.CGO) continue;
          const int64_t slot = ((int64_t)n * p.CGO + group) * OHW + pixel;
          float v[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) v[jj] = acc[mi][ni][4 * qd + jj];
          if (p.epi == 1) {
            const float4 b4 = bias4[mi][qd];
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = v[jj] > 0.f ? v[jj] : v[jj] * p.slope;
          } else if (p.epi == 2) {
            const float4 r = refs[mi][qd];
            v[0] *= r.x > 0.f ? 1.f : p.slope; v[1] *= r.y > 0.f ? 1.f : p.slope;
            v[2] *= r.z > 0.f ? 1.f : p.slope; v[3] *= r.w > 0.f ? 1.f : p.slope;
          }
          *reinterpret_cast<float4*>(p.out + slot) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          // two register quads = this lane's half (channels 4 lhi .. 4 lhi + 3) of the slots of groups g0 and g0 + 1; the halves
          // change places between the half-waves (v_permlane32_swap, blocked16.hip hconv3_epilogue) and every lane stores a whole slot
          const int g0 = (m0 + mi * 32) / 8 + 2 * qp;
          uint2 packed[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int qd = 2 * qp + h;
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = acc[mi][ni][4 * qd + jj];
            if (p.epi == 1) {
              const float4 b4 = bias4[mi][qd];
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) v[jj] = v[jj] > 0.f ? v[jj] : v[jj] * p.slope;
            } else if (p.epi == 2) {
              const uint2 r = refs[mi][qd];
              v[0] *= h_mask(r.x & 0xFFFFu, p.slope); v[1] *= h_mask(r.x >> 16, p.slope);
              v[2] *= h_mask(r.y & 0xFFFFu, p.slope); v[3] *= h_mask(r.y >> 16, p.slope);
            }
            packed[h].x = h_pack2<PREC>(v[0], v[1]);
            packed[h].y = h_pack2<PREC>(v[2], v[3]);
          }
          const auto first = __builtin_amdgcn_permlane32_swap(packed[0].x, packed[1].x, false, false);
          const auto second = __builtin_amdgcn_permlane32_swap(packed[0].y, packed[1].y, false, false);
          const int group = g0 + lhi;
          if (group < p.CGO) {
            uint4 whole;
            whole.x = first[0]; whole.y = second[0]; whole.z = first[1]; whole.w = second[1];
            *reinterpret_cast<uint4*>(p.out + ((int64_t)n * p.CGO + group) * OHW + pixel) = whole;
          }
        }
      }
    }
  }
}

template <int BM, int NI, int TW, int ROWS, int PREC>
static int hconv2_launch_one(const HConv2Params& p, dim3 grid, hipStream_t stream) {
  constexpr int P = 128 * NI, IMG = P / (ROWS * TW), PLANE = (ROWS + 1) * (TW + 1);
  constexpr int PATCH_I = (K4_CK * IMG * PLANE + 63) / 64, T = PATCH_I + K4_TAPS * K4_CK * BM / 64, TP = (T + 3) / 4 * 4;
  constexpr int bytes = 2 * TP * 64 * 16;
  static_assert(bytes <= 80 * 1024, "two workgroups per CU");
  auto kernel = hconv2x2_kernel<BM, NI, TW, ROWS, PREC>;
  static std::atomic<uint64_t> configured_devices{0};
  int device = 0;
  SRGAN_HIP(hipGetDevice(&device));
  const uint64_t bit = (uint64_t)1 << (device & 63);
  if (!(configured_devices.load(std::memory_order_acquire) & bit)) {
    SRGAN_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    configured_devices.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kernel, grid, dim3(256), bytes, stream, p, h_zero_slots());
  return SRGAN_OK;
}

template <int BM, int PREC>
static int hconv2_launch(const HConv2Params& p, int ni, int tw, dim3 grid, hipStream_t stream) {
  if (ni == 2) {                                       // (4-wide tiles: 256 pixels would need 90 KB of LDS)
    if (tw == 32) return hconv2_launch_one<BM, 2, 32, 8, PREC>(p, grid, stream);
    if (tw == 16) return hconv2_launch_one<BM, 2, 16, 16, PREC>(p, grid, stream);
    return hconv2_launch_one<BM, 2, 8, 8, PREC>(p, grid, stream);
  }
  if (tw == 32) return hconv2_launch_one<BM, 1, 32, 4, PREC>(p, grid, stream);
  if (tw == 16) return hconv2_launch_one<BM, 1, 16, 8, PREC>(p, grid, stream);
  if (tw == 8) return hconv2_launch_one<BM, 1, 8, 8, PREC>(p, grid, stream);
  return hconv2_launch_one<BM, 1, 4, 4, PREC>(p, grid, stream);
}

// One launch of the 2x2-tap kernel over a GH x GW grid of output positions.
static int hconv2_run(HConv2Params& p, int dtype, int64_t flops_k, hipStream_t stream) {
  // tile width: the one of {32, 16, 8, 4} that covers GW with the fewest dead columns (ties: the wider)
  int tw = 32, best = 1 << 30;
  for (int candidate : {32, 16, 8, 4}) {
    const int covered = (p.GW + candidate - 1) / candidate * candidate;
    if (covered < best) { best = covered; tw = candidate; }
  }
  const int bm = p.CO > 32 ? 64 : 32;
  p.tiles_m = (p.CO + bm - 1) / bm;
  auto rows_of = [&](int ni) { return tw == 32 ? 4 * ni : (tw == 16 ? 8 * ni : tw); };
  auto count = [&](int ni) {
    const int rows = rows_of(ni), img = 128 * ni / (rows * tw);
    return (int64_t)((p.GW + tw - 1) / tw) * ((p.GH + rows - 1) / rows) * ((p.N + img - 1) / img) * p.tiles_m;
  };
  const int classes = p.down ? 1 : 4;
  const int ni = (tw != 4 && count(2) * classes >= 512) ? 2 : 1;
  const int rows = rows_of(ni), img = 128 * ni / (rows * tw);
  p.tiles_x = (p.GW + tw - 1) / tw;
  p.tiles_y = (p.GH + rows - 1) / rows;
  const int64_t blocks = count(ni);
  SRGAN_REQUIRE(blocks < ((int64_t)1 << 31), SRGAN_ERANGE, "k4s2 convolution grid");
  (void)img;
  p.xcd_remap = (blocks % 8 == 0 && blocks >= 64) ? 1 : 0;
  const dim3 grid((unsigned)blocks, p.down ? 1u : 4u);
  const int slot = profile_bracket_begin(stream);
  int launched;
  if (bm == 64) launched = dtype == 0 ? hconv2_launch<64, 0>(p, ni, tw, grid, stream) : (dtype == 1 ? hconv2_launch<64, 1>(p, ni, tw, grid, stream) : hconv2_launch<64, 2>(p, ni, tw, grid, stream));
  else launched = dtype == 0 ? hconv2_launch<32, 0>(p, ni, tw, grid, stream) : (dtype == 1 ? hconv2_launch<32, 1>(p, ni, tw, grid, stream) : hconv2_launch<32, 2>(p, ni, tw, grid, stream));
  if (launched != SRGAN_OK) return launched;
  const int status = launch_status();
  const double positions = (double)p.N * p.GH * p.GW * (p.down ? 1 : 4);       // output pixels
  profile_bracket_end_bytes(slot, stream, p.CO, (int64_t)positions, flops_k, 18, bm, ni * 128, 1,
                            16.0 * ((double)p.N * p.CGI * p.IH * p.IW + positions * p.CGO * (p.epi == 2 ? 2 : 1)), dtype);
  return status;
}

// ---------------------------------------------------------------------------------------------------- weight gradient
//   gw[k][c][kh][kw] (fp32; element at gw[k * sk + c * sc + kh * 4 + kw]) += sum_{n, y, x} small[n][k][y][x] * big[n][c][2y - 1 + kh][2x - 1 + kw]
// small = the tensor on the H/2 x W/2 plane (the strided convolution's output gradient; a transposed convolution's input),
// big = the one on the H x W plane.  M = k, N = c, K = small-plane pixels.  A workgroup (4 waves) owns a 64 (k) x 32 (c) block
// for all sixteen taps: wave w = parity (qy, qx) = the four taps (2a + qy, 2b + qx), eight 32 x 32 accumulators (MI = 2 x 4
// taps).  Per tile of 64 small-plane pixels the small slots [8 groups][64] and the big tensor's space-to-depth patches
// [4 groups][4 parities][IMG][(ROWS + 1) x (TW + 1)] are staged once; fragments by transpose reads (blocked16.h).  Walkers
// over the pixel tiles leave partial blocks; the finish adds them in walker order.
struct HWgrad4Params {
  const Slot* big; const Slot* small; float* partial;
  int32_t N, CGB, CGS, H, W, SH, SW;          // big plane H x W, small plane SH x SW
  int32_t tiles_c, tiles_x, tiles_y, tiles_n, pixel_tiles, walkers;
};

constexpr int h4_pad_stride(int slots) { return ((slots + 15) / 16) * 16 + 4; }

template <int TW, int ROWS, int PREC>
__global__ __launch_bounds__(256, 2) void hwgrad4x4s2_kernel(const HWgrad4Params p) {
  constexpr int P = 64, IMG = P / (ROWS * TW), PW = TW + 1, PH = ROWS + 1, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int SS = h4_pad_stride(P), BS = h4_pad_stride(IMG * PLANE);     // group strides of the two LDS images (slots)
  constexpr int SQ = 8 * P, BQ = 16 * IMG * PLANE;                          // slots staged per tile: 8 small groups; 4 groups x 4 parities
  constexpr int NS = SQ / 256, NB = (BQ + 255) / 256;
  __shared__ Slot lds[8 * SS + 16 * BS];
  Slot* ss = lds;
  Slot* bs = lds + 8 * SS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qy = wave >> 1, qx = wave & 1;
  const int tc = (int)blockIdx.x % p.tiles_c, tk = (int)blockIdx.x / p.tiles_c;
  const int walker = (int)blockIdx.y;
  const int HW = p.H * p.W, SHW = p.SH * p.SW;

  Slot rs_[NS], rb[NB];
  uint32_t oks = 0, okb = 0;
  auto fetch = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int rest_t = tile / p.tiles_x;
    const int ty = rest_t % p.tiles_y;
    const int n0 = (rest_t / p.tiles_y) * IMG;
    const int y0 = ty * ROWS, x0 = tx * TW;
    oks = okb = 0;
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int flat = e * 256 + tid;
      const int grp = flat / P, q = flat % P;
      const int n = n0 + q / (ROWS * TW), y = y0 + (q / TW) % ROWS, x = x0 + q % TW;
      const int group = tk * 8 + grp;
      const bool ok = n < p.N && y < p.SH && x < p.SW && group < p.CGS;
      oks |= (ok ? 1u : 0u) << e;
      rs_[e] = p.small[ok ? ((int64_t)n * p.CGS + group) * SHW + y * p.SW + x : 0];
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int flat = e * 256 + tid;
      const int gq = flat / (IMG * PLANE), rest = flat - gq * (IMG * PLANE);      // gq = group * 4 + parity
      const int img = rest / PLANE, pix = rest % PLANE;
      const int n = n0 + img;
      const int sy = 2 * (y0 + pix / PW) - 1 + ((gq >> 1) & 1), sx = 2 * (x0 + pix % PW) - 1 + (gq & 1);
      const int group = tc * 4 + (gq >> 2);
      const bool ok = flat < BQ && n < p.N && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W && group < p.CGB;
      okb |= (ok ? 1u : 0u) << e;
      rb[e] = p.big[ok ? ((int64_t)n * p.CGB + group) * HW + sy * p.W + sx : 0];
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.f;

  // fragment addressing as hwgrad3x3_kernel: 16-lane group G: row block rb16 = G & 1, k half = G >> 1; lane s = 4 j + u
  const int G = lane >> 4, rb16 = G & 1, khalf = G >> 1, s16 = lane & 15, j = s16 >> 2, u = s16 & 3;
  constexpr int HALF = TW >= 16 ? 8 : (TW == 8 ? PW : 2 * PW);
  constexpr int QUAD = TW >= 8 ? 4 : PW;
  uint32_t a_base[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
    a_base[mi] = h_lds_address(ss) + (uint32_t)(((4 * mi + 2 * rb16 + (u >> 1)) * SS + 8 * khalf + j) * 16 + (u & 1) * 8);
  // B: the 32 channels of the block = groups 0..3, each with four parity images: image index = group * 4 + (2 qy + qx)
  const uint32_t b_base = h_lds_address(bs) + (uint32_t)((((2 * rb16 + (u >> 1)) * 4 + 2 * qy + qx) * BS + khalf * HALF + j) * 16 + (u & 1) * 8);
  const bool active = tk * 64 < p.CGS * 8 && tc * 32 < p.CGB * 8;

  int tile = walker;
  if (tile < p.pixel_tiles) fetch(tile);
  for (; tile < p.pixel_tiles; tile += p.walkers) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int flat = e * 256 + tid;
      Slot v = rs_[e];
      if (!((oks >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      ss[(flat / P) * SS + flat % P] = v;
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int flat = e * 256 + tid;
      const int gq = flat / (IMG * PLANE), rest = flat - gq * (IMG * PLANE);
      Slot v = rb[e];
      if (!((okb >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      if (flat < BQ) bs[gq * BS + rest] = v;
    }
    __syncthreads();
    const int next = tile + p.walkers;
    if (next < p.pixel_tiles) fetch(next);
    if (active)
#pragma unroll 2
    for (int t = 0; t < P / 16; ++t) {
      const int pix = 16 * t;
      const int origin = ((pix / (ROWS * TW)) * PH + (pix / TW) % ROWS) * PW + pix % TW;
      Slot a[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const uint2 a0 = h_tr_read(a_base[mi] + pix * 16), a1 = h_tr_read(a_base[mi] + (pix + 4) * 16);
        a[mi].v[0] = a0.x; a[mi].v[1] = a0.y; a[mi].v[2] = a1.x; a[mi].v[3] = a1.y;
      }
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int shift = origin + (tap >> 1) * PW + (tap & 1);
        Slot b;
        const uint2 b0 = h_tr_read(b_base + shift * 16), b1 = h_tr_read(b_base + (shift + QUAD) * 16);
        b.v[0] = b0.x; b.v[1] = b0.y; b.v[2] = b1.x; b.v[3] = b1.y;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][tap] = h_mfma<PREC>(a[mi], b, acc[mi][tap]);
      }
    }
  }

  float* mine = p.partial + ((int64_t)blockIdx.x * p.walkers + walker) * (8 * 16 * 256) + tid;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((mi * 4 + t) * 16 + r) * 256] = acc[mi][t][r];
}

// The same weight gradient on fp32 blocked tensors (four channels per slot; the exact gradient-penalty chain of the fp16
// configuration, the crowd generator).  v_mfma_f32_32x32x2_f32 takes ONE float per lane -- A[i = lane & 31][k = lane >> 5] -- so a
// fragment is a plain ds_read_b32 of (pixel 2 s + k, channel i) and no transposition is involved; 32 small-plane pixels per staged
// tile = 16 matrix steps x 8 accumulators x 64 cycles of matrix work per wave and barrier.  The LDS images are [16 small groups]
// [33] and [4 parities][8 big groups][IMG x patch, stride = 1 mod 8]: the 32 lanes of a read (8 groups x 4 channels) then fall
// on 32 different banks.  Same tile ownership, partial layout and finish as hwgrad4x4s2_kernel.
constexpr int h4_stride_f32(int slots) { return ((slots + 6) / 8) * 8 + 1; }      // = 1 (mod 8), >= slots

template <int TW, int ROWS>
__global__ __launch_bounds__(256, 2) void hwgrad4x4s2_f32_kernel(const HWgrad4Params p) {
  constexpr int P = 32, IMG = P / (ROWS * TW), PW = TW + 1, PH = ROWS + 1, PLANE = PH * PW;
  static_assert(IMG * ROWS * TW == P && IMG >= 1, "the pixel tile is IMG x ROWS x TW");
  constexpr int SS = h4_stride_f32(P), BS = h4_stride_f32(IMG * PLANE);
  constexpr int SQ = 16 * P, BQ = 32 * IMG * PLANE;         // 16 small groups of 4 channels; 8 big groups x 4 parities
  constexpr int NS = SQ / 256, NB = (BQ + 255) / 256;
  __shared__ Slot lds[16 * SS + 32 * BS];
  Slot* ss = lds;
  Slot* bs = lds + 16 * SS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int qy = wave >> 1, qx = wave & 1;
  const int tc = (int)blockIdx.x % p.tiles_c, tk = (int)blockIdx.x / p.tiles_c;
  const int walker = (int)blockIdx.y;
  const int HW = p.H * p.W, SHW = p.SH * p.SW;

  Slot rs_[NS], rb[NB];
  uint32_t oks = 0, okb = 0;
  auto fetch = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int rest_t = tile / p.tiles_x;
    const int ty = rest_t % p.tiles_y;
    const int n0 = (rest_t / p.tiles_y) * IMG;
    const int y0 = ty * ROWS, x0 = tx * TW;
    oks = okb = 0;
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int flat = e * 256 + tid;
      const int grp = flat / P, q = flat % P;
      const int n = n0 + q / (ROWS * TW), y = y0 + (q / TW) % ROWS, x = x0 + q % TW;
      const int group = tk * 16 + grp;
      const bool ok = n < p.N && y < p.SH && x < p.SW && group < p.CGS;
      oks |= (ok ? 1u : 0u) << e;
      rs_[e] = p.small[ok ? ((int64_t)n * p.CGS + group) * SHW + y * p.SW + x : 0];
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int flat = e * 256 + tid;
      const int qg = flat / (IMG * PLANE), rest = flat - qg * (IMG * PLANE);      // qg = parity * 8 + group
      const int img = rest / PLANE, pix = rest % PLANE;
      const int n = n0 + img;
      const int parity = qg >> 3;
      const int sy = 2 * (y0 + pix / PW) - 1 + (parity >> 1), sx = 2 * (x0 + pix % PW) - 1 + (parity & 1);
      const int group = tc * 8 + (qg & 7);
      const bool ok = flat < BQ && n < p.N && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W && group < p.CGB;
      okb |= (ok ? 1u : 0u) << e;
      rb[e] = p.big[ok ? ((int64_t)n * p.CGB + group) * HW + sy * p.W + sx : 0];
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.f;

  // fragment addresses (floats): A = small[pixel 2 s + lhi][channel 32 mi + l31], B = big-parity image[window of pixel 2 s + lhi][channel l31]
  const float* a_base[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
    a_base[mi] = reinterpret_cast<const float*>(ss) + ((8 * mi + (l31 >> 2)) * SS + lhi) * 4 + (l31 & 3);
  const float* b_base = reinterpret_cast<const float*>(bs) + (((2 * qy + qx) * 8 + (l31 >> 2)) * BS + lhi) * 4 + (l31 & 3);
  const bool active = tk * 64 < p.CGS * 4 && tc * 32 < p.CGB * 4;

  int tile = walker;
  if (tile < p.pixel_tiles) fetch(tile);
  for (; tile < p.pixel_tiles; tile += p.walkers) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int flat = e * 256 + tid;
      Slot v = rs_[e];
      if (!((oks >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      ss[(flat / P) * SS + flat % P] = v;
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int flat = e * 256 + tid;
      const int qg = flat / (IMG * PLANE), rest = flat - qg * (IMG * PLANE);
      Slot v = rb[e];
      if (!((okb >> e) & 1u)) v = Slot{{0u, 0u, 0u, 0u}};
      if (flat < BQ) bs[qg * BS + rest] = v;
    }
    __syncthreads();
    const int next = tile + p.walkers;
    if (next < p.pixel_tiles) fetch(next);
    if (active)
#pragma unroll 4
    for (int st = 0; st < P / 2; ++st) {
      const int pix = 2 * st;                      // (an even pixel and its right neighbour: the same row of the tile)
      const int origin = ((pix / (ROWS * TW)) * PH + (pix / TW) % ROWS) * PW + pix % TW;
      float a[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = a_base[mi][pix * 4];
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const float b = b_base[(origin + (tap >> 1) * PW + (tap & 1)) * 4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b, acc[mi][tap], 0, 0, 0);
      }
    }
  }

  float* mine = p.partial + ((int64_t)blockIdx.x * p.walkers + walker) * (8 * 16 * 256) + tid;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((mi * 4 + t) * 16 + r) * 256] = acc[mi][t][r];
}

// One workgroup per (block, row k of the block): 32 columns c x 16 taps = 512 floats; element (c, kh, kw) of row k at
// gw[k * sk + c * sc + kh * 4 + kw].  Wave = parity (qy, qx) of the producing kernel, so the reader of (kh, kw) looks into
// wave 2 (kh & 1) + (kw & 1), tap 2 (kh >> 1) + (kw >> 1); C/D row k % 32 = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column c = lane & 31.
// The eight 32-lane groups of the workgroup take the walkers w = group, group + 8, ...: a lane adds its column of all sixteen
// taps (sixteen independent 128-byte-run loads per walker), and the eight sums meet in a fixed tree -- the first version walked
// up to 384 walkers serially in every thread (362 us for the three-channel first layer, profiles/r06l_*).
__global__ __launch_bounds__(256) void hwgrad4x4s2_finish_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                 int32_t K, int32_t C, int32_t tiles_c, int32_t walkers, int64_t sk,
                                                                 int64_t sc) {
  __shared__ float sums[8][16][32];
  const int block = (int)blockIdx.x / 64, row = (int)blockIdx.x % 64;
  const int tk = block / tiles_c, tc = block % tiles_c;
  const int k = tk * 64 + row;
  if (k >= K) return;
  const int64_t per_walker = 8 * 16 * 256;
  const int mi = row >> 5, r32 = row & 31, lhi = (r32 >> 2) & 1, r = (r32 & 3) + 4 * (r32 >> 3);
  const int c32 = (int)threadIdx.x & 31, part = (int)threadIdx.x >> 5;
  const float* base = partial + (int64_t)block * walkers * per_walker + lhi * 32 + c32;
  float total[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) total[t] = 0.f;
  // four walkers' loads in flight per lane group, added in walker order (the sum is the one-at-a-time loop's, bit for bit): a
  // single-block layer has 512 walkers = 64 dependent round trips per lane group, ~100 us of the first layer's 208 us launch
  for (int w0 = part; w0 < walkers; w0 += 32) {
    float v[4][16];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int w = w0 + 8 * d;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kh = t >> 2, kw = t & 3;
        v[d][t] = w < walkers ? base[(int64_t)w * per_walker + ((mi * 4 + 2 * (kh >> 1) + (kw >> 1)) * 16 + r) * 256 + (2 * (kh & 1) + (kw & 1)) * 64]
                              : 0.f;
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (w0 + 8 * d < walkers)
#pragma unroll
        for (int t = 0; t < 16; ++t) total[t] += v[d][t];
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) sums[part][t][c32] = total[t];
  __syncthreads();
  float previous[2];                        // read both, then write both (loads wait for the stores in front of them: one vmcnt)
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = (int)threadIdx.x + 256 * e, cc = i >> 4, t = i & 15, c = tc * 32 + cc;
    previous[e] = c < C ? gw[(int64_t)k * sk + (int64_t)c * sc + t] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = (int)threadIdx.x + 256 * e;
    const int cc = i >> 4, t = i & 15;                    // consecutive threads: the 16 taps of one column = 64 contiguous bytes
    const int c = tc * 32 + cc;
    if (c < C)
      gw[(int64_t)k * sk + (int64_t)c * sc + t] = previous[e] + (((sums[0][t][cc] + sums[1][t][cc]) + (sums[2][t][cc] + sums[3][t][cc])) +
                                                                  ((sums[4][t][cc] + sums[5][t][cc]) + (sums[6][t][cc] + sums[7][t][cc])));
  }
}

static int check_dtype_k(int dtype) {
  SRGAN_REQUIRE(dtype >= 0 && dtype <= 2, SRGAN_EINVAL, "blocked tensors are fp32 (0: four channels per slot), bf16 (1) or fp16 (2)");
  return SRGAN_OK;
}
static inline int group_of(int dtype) { return dtype == 0 ? 4 : 8; }

}  // namespace srgan

using namespace srgan;

extern "C" {

// 16-byte slots of one operand of a [A][B][4][4] weight tensor: direction 0 "down" (rows = A: the strided convolution's forward /
// a transposed convolution's data gradient), 1 "up" (rows = B, all four output parity classes, class-major).
int64_t srgan_h_k4s2_weight_slots(int32_t A, int32_t B, int direction, int dtype) {
  const int g = group_of(dtype);
  if (direction == 0) return (int64_t)((B + g - 1) / g) * K4_TAPS * K4_CK * A;
  return (int64_t)4 * (((A + g - 1) / g + K4_CK - 1) / K4_CK) * K4_TAPS * K4_CK * B;
}

int srgan_h_pack_k4s2_weights(const float* w, void* packed, int32_t A, int32_t B, int direction, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_k(dtype)) return status;
  SRGAN_REQUIRE(w && packed && A > 0 && B > 0 && (direction == 0 || direction == 1), SRGAN_EINVAL, "srgan_h_pack_k4s2_weights arguments");
#define K4_PACK(...)                                                                                                     \
  do {                                                                                                                  \
    if (dtype == 0) hipLaunchKernelGGL(h_pack_k4s2_weights_kernel<0>, __VA_ARGS__);                                     \
    else if (dtype == 1) hipLaunchKernelGGL(h_pack_k4s2_weights_kernel<1>, __VA_ARGS__);                                \
    else hipLaunchKernelGGL(h_pack_k4s2_weights_kernel<2>, __VA_ARGS__);                                                \
  } while (0)
  if (direction == 0) {
    const int64_t slots = srgan_h_k4s2_weight_slots(A, B, 0, dtype);
    const dim3 grid((unsigned)((slots + 255) / 256));
    K4_PACK(grid, dim3(256), 0, stream, w, (Slot*)packed, slots, A, B, (int64_t)B * 16, (int64_t)16, 0);
    return launch_status();
  }
  const int64_t per_class = srgan_h_k4s2_weight_slots(A, B, 1, dtype) / 4;
  const dim3 grid((unsigned)((per_class + 255) / 256));
  for (int cls = 0; cls < 4; ++cls) {
    Slot* into = (Slot*)packed + cls * per_class;
    K4_PACK(grid, dim3(256), 0, stream, w, into, per_class, B, A, (int64_t)16, (int64_t)B * 16, 1 + cls);
  }
#undef K4_PACK
  return launch_status();
}

// "down": out[N, rows, H/2, W/2] = epi(conv2d 4x4 / stride 2 / pad 1 of x[N, C_in, H, W]) with the direction-0 operand of
// srgan_h_pack_k4s2_weights (rows = its A).  epi as srgan_h_conv3x3.
int srgan_h_conv4x4s2(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi, void* out,
                      int32_t N, int32_t C_in, int32_t rows, int32_t H, int32_t W, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_k(dtype)) return status;
  SRGAN_REQUIRE(x && packed && out && N > 0 && C_in > 0 && rows > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && epi >= 0 &&
                epi <= 2 && (epi != 2 || ref), SRGAN_EINVAL, "srgan_h_conv4x4s2 arguments (even planes)");
  HConv2Params p;
  p.in = (const Slot*)x; p.wp = (const Slot*)packed; p.out = (Slot*)out; p.bias = bias; p.ref = (const Slot*)ref;
  p.slope = slope; p.epi = epi;
  const int g = group_of(dtype);
  p.N = N; p.CGI = (C_in + g - 1) / g; p.IH = H; p.IW = W;
  p.down = 1; p.oy = -1; p.ox = -1;
  p.GH = H / 2; p.GW = W / 2;
  p.CGO = (rows + g - 1) / g; p.OH = H / 2; p.OW = W / 2; p.dsy = p.dsx = 1; p.doy = p.dox = 0;
  p.CO = rows; p.C_real = rows;
  p.chunks = p.CGI;
  p.class_stride = 0;
  SRGAN_REQUIRE((int64_t)N * p.CGI * H * W < ((int64_t)1 << 31) && (int64_t)N * p.CGO * p.OH * p.OW < ((int64_t)1 << 31), SRGAN_ERANGE,
                "srgan_h_conv4x4s2 tensor size");
  return hconv2_run(p, dtype, (int64_t)p.CGI * g * 16, stream);
}

// "up": out[N, rows, 2h, 2w] = epi(conv_transpose2d 4x4 / stride 2 / pad 1 of x[N, C_in, h, w]) with the direction-1 operand
// (rows = its B, C_in = its A): one launch, gridDim.y = the four output parity classes.
int srgan_h_conv_transpose4x4s2(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi,
                                void* out, int32_t N, int32_t C_in, int32_t rows, int32_t h, int32_t w, int dtype,
                                hipStream_t stream) {
  if (const int status = check_dtype_k(dtype)) return status;
  SRGAN_REQUIRE(x && packed && out && N > 0 && C_in > 0 && rows > 0 && h > 0 && w > 0 && epi >= 0 && epi <= 2 && (epi != 2 || ref),
                SRGAN_EINVAL, "srgan_h_conv_transpose4x4s2 arguments");
  const int g = group_of(dtype);
  HConv2Params p;
  p.in = (const Slot*)x; p.wp = (const Slot*)packed; p.out = (Slot*)out; p.bias = bias; p.ref = (const Slot*)ref;
  p.slope = slope; p.epi = epi;
  p.N = N; p.CGI = (C_in + g - 1) / g; p.IH = h; p.IW = w;
  p.down = 0; p.oy = p.ox = -1;                                  // (set per class in the kernel)
  p.GH = h; p.GW = w;
  p.CGO = (rows + g - 1) / g; p.OH = 2 * h; p.OW = 2 * w; p.dsy = p.dsx = 2; p.doy = p.dox = 0;
  p.CO = rows; p.C_real = rows;
  p.chunks = (p.CGI + K4_CK - 1) / K4_CK;
  p.class_stride = srgan_h_k4s2_weight_slots(C_in, rows, 1, dtype) / 4;
  SRGAN_REQUIRE((int64_t)N * p.CGI * h * w < ((int64_t)1 << 31) && (int64_t)N * p.CGO * p.OH * p.OW < ((int64_t)1 << 31), SRGAN_ERANGE,
                "srgan_h_conv_transpose4x4s2 tensor size");
  return hconv2_run(p, dtype, (int64_t)p.CGI * g * 4, stream);
}

// gw (fp32 [A][B][4][4] in torch's layout) += the weight gradient of the 4x4 / stride 2 / pad 1 pair from `small` [N, C_small, H/2,
// W/2] and `big` [N, C_big, H, W].  small_is_rows = 1: gw is indexed [C_small][C_big] (a strided convolution's weight: small = its
// output gradient); 0: [C_big][C_small] (a transposed convolution's weight: small = its input).
int srgan_h_k4s2_wgrad(const void* big, const void* small, float* gw, int32_t N, int32_t C_big, int32_t C_small, int32_t H, int32_t W,
                       int small_is_rows, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_k(dtype)) return status;
  SRGAN_REQUIRE(big && small && gw && N > 0 && C_big > 0 && C_small > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SRGAN_EINVAL,
                "srgan_h_k4s2_wgrad arguments");
  HWgrad4Params p;
  const int g = group_of(dtype);
  p.big = (const Slot*)big; p.small = (const Slot*)small;
  p.N = N; p.CGB = (C_big + g - 1) / g; p.CGS = (C_small + g - 1) / g; p.H = H; p.W = W; p.SH = H / 2; p.SW = W / 2;
  int tw = 32, best = 1 << 30;
  for (int candidate : {32, 16, 8, 4}) {
    const int covered = (p.SW + candidate - 1) / candidate * candidate;
    if (covered < best) { best = covered; tw = candidate; }
  }
  const int pixels_per_tile = dtype == 0 ? 32 : 64;
  const int rows = pixels_per_tile / tw > 0 ? (tw == 4 ? 4 : pixels_per_tile / tw) : 1;      // 32: 2 (1) rows, 16: 4 (2), 8: 8 (4), 4: 4
  const int img = pixels_per_tile / (rows * tw);
  p.tiles_x = (p.SW + tw - 1) / tw;
  p.tiles_y = (p.SH + rows - 1) / rows;
  p.tiles_n = (N + img - 1) / img;
  p.pixel_tiles = p.tiles_x * p.tiles_y * p.tiles_n;
  p.tiles_c = (C_big + 31) / 32;
  const int tiles_k = (C_small + 63) / 64;
  const int blocks = p.tiles_c * tiles_k;
  static const char* forced_walkers = getenv("SRGAN_H_K4_WALKERS");
  const int target = forced_walkers ? atoi(forced_walkers) : 512;      // two workgroups per CU x 256 CUs (384: -1.4 % on driving-fp16)
  int walkers = (target + blocks - 1) / blocks;
  if (walkers > p.pixel_tiles) walkers = p.pixel_tiles;
  if (walkers < 1) walkers = 1;
  p.walkers = walkers;
  p.partial = partial_workspace((size_t)blocks * walkers * 8 * 16 * 256 * sizeof(float), stream);
  SRGAN_REQUIRE(p.partial, SRGAN_EINVAL, "srgan_h_k4s2_wgrad: register a workspace for this stream first (srgan_set_workspace)");
  const dim3 grid((unsigned)blocks, (unsigned)walkers);
  const int slot = profile_bracket_begin(stream);
#define HWGRAD4_LAUNCH(TWv, ROWSv, ROWSf)                                                                              \
  do {                                                                                                                  \
    if (dtype == 0) hipLaunchKernelGGL((hwgrad4x4s2_f32_kernel<TWv, ROWSf>), grid, dim3(256), 0, stream, p);           \
    else if (dtype == 1) hipLaunchKernelGGL((hwgrad4x4s2_kernel<TWv, ROWSv, 1>), grid, dim3(256), 0, stream, p);      \
    else hipLaunchKernelGGL((hwgrad4x4s2_kernel<TWv, ROWSv, 2>), grid, dim3(256), 0, stream, p);                      \
  } while (0)
  if (tw == 32) HWGRAD4_LAUNCH(32, 2, 1);
  else if (tw == 16) HWGRAD4_LAUNCH(16, 4, 2);
  else if (tw == 8) HWGRAD4_LAUNCH(8, 8, 4);
  else HWGRAD4_LAUNCH(4, 4, 4);
#undef HWGRAD4_LAUNCH
  const int64_t sk = small_is_rows ? (int64_t)C_big * 16 : 16, sc = small_is_rows ? 16 : (int64_t)C_small * 16;
  hipLaunchKernelGGL(hwgrad4x4s2_finish_kernel, dim3((unsigned)(blocks * 64)), dim3(256), 0, stream, p.partial, gw, C_small, C_big,
                     p.tiles_c, walkers, sk, sc);
  const int status = launch_status();
  const double pixels = (double)N * p.SH * p.SW;
  profile_bracket_end_bytes(slot, stream, C_small, (int64_t)C_big * 16, (int64_t)pixels, 19, 64, 32, walkers,
                            2.0 * (pixels * p.CGS * 8 + 4.0 * pixels * p.CGB * 8) + 8.0 * (double)C_small * C_big * 16, dtype);
  return status;
}

}  // extern "C"
