// gather_gemm_kernels.hip -- the MFMA contraction kernel behind every convolution / linear pass.
//
// One 256-thread workgroup (4 wavefronts of 64) owns a BM x BN tile of C.  Per K-step of 16 the tile's A
// (BM x 16) and B (16 x BN) slices are gathered from HBM into registers (next step's loads are issued
// before the current step's MFMAs so HBM latency hides under the matrix pipe), written to LDS as
// [k][m] / [k][n] rows padded by one word (conflict-free ds_read_b32 for the MFMA operand pattern), and
// consumed by v_mfma_f32_32x32x2_f32: exact fp32, 157 TF/s peak on gfx950.  Lanes map to C columns, so
// stores are 128-byte runs along the contiguous (pixel) dimension of NCHW.
//
// Roofline (MI355X_MICROARCH.md): fp32 MFMA 157.3 TF/s; the gather adds ~30 VALU ops per staged element,
// which overlaps with the 64-cycle MFMAs of the other waves on the SIMD.
#include "common.h"
#include "split_finish.h"
#include "gather_gemm.h"
#include "conv_plan.h"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <type_traits>
#include <utility>

namespace srgan {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_error; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GG_BK = 32;   // K-slice alignment of split-K chunks (the largest per-config BK)

// BK is 16 for the 128x128 and 32x256 tiles (keeps VGPR + AGPR <= 256: two workgroups per CU so that one's staging
// overlaps the other's MFMAs) and 32 for the smaller tiles (half the barriers per FLOP).
// The 16-byte staged (VEC) variants need few registers, so they take K-slices of 32: twice the MFMA work per barrier
// and per load round trip.
template <int BM, int BN, bool VEC>
struct TileBK { static constexpr int value = (BM * BN >= 128 * 128 || BN >= 256) ? 16 : 32; };

// VEC: both operands are staged with 16-byte loads along their contiguous direction (groups of 4 consecutive k or
// m/n that share one address decode): the 1x1 convolutions and linear layers, i.e. plain GEMMs on NCHW data.
// The host (vec_eligible) guarantees that every group is 16-byte aligned and entirely valid or entirely invalid.
// PREC: 0 = v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak); 1 / 2 = v_mfma_f32_32x32x16_bf16 / _f16 (2.5 PF/s peak):
// lane l of a fragment holds row / column l & 31 and k = 8 * (l >> 5) .. + 7 of a 16-deep step.  The tiles in LDS are
// OPERAND-TYPED (round 3): stage() rounds the fp32 values it fetched and writes them as [row][k] with k contiguous and a
// row stride of BK + 8 elements (80 / 48 bytes: the sixteen lanes of a ds_read_b128 group hit sixteen different bank
// quads), so a fragment is ONE ds_read_b128 -- it used to be eight ds_read_b32 of fp32 values plus eight conversions per
// fragment, which kept the kernel at 66-79 TF/s in the bf16 mode, below its own fp32 rate.  Accumulation is fp32 and the
// C/D fragment layout is the same, so fetch and epilogue are shared.  The mixed-precision modes of BASELINE.json configs
// 2 and 5.
template <int BM, int BN, int WGM, bool AKF, bool BKF, bool VEC, int PREC = 0>
__global__ __launch_bounds__(256, 2) void gg_mfma_kernel(const GatherGemm p) {
  constexpr int BK = TileBK<BM, BN, VEC>::value;
  constexpr int WGN = 4 / WGM;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int G = VEC ? 4 : 1;                                  // elements per staged group
  constexpr int AG = BM * BK / G, BG = BN * BK / G;               // groups per tile
  constexpr int EA = (AG + 255) / 256, EB = (BG + 255) / 256;     // groups per thread
  constexpr int LDA = BM + (VEC ? 4 : 1), LDB = BN + (VEC ? 4 : 1);
  static_assert(MI >= 1 && NI >= 1, "tile too small for 4 waves");
  __shared__ __attribute__((aligned(16))) float lds[BK * LDA + BK * LDB];
  float* As = lds;
  float* Bs = lds + BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int m0 = (int)(blockIdx.x % tiles_m) * BM, n0 = (int)(blockIdx.x / tiles_m) * BN;
  const int kbeg = (int)blockIdx.y * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  // Per-thread staging coordinates (first element of each group).  With k-fast staging consecutive lanes walk k
  // (the operand's contiguous direction), otherwise they walk m / n.
  int a_kk[EA], a_ml[EA], b_kk[EB], b_nl[EB];
  bool a_on[EA], b_on[EB];
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    const int flat = e * 256 + tid;
    a_on[e] = (AG % 256 == 0) || flat < AG;
    a_kk[e] = AKF ? (flat % (BK / G)) * G : flat / (BM / (AKF ? 1 : G));
    a_ml[e] = AKF ? flat / (BK / G) : (flat % (BM / G)) * G;
  }
#pragma unroll
  for (int e = 0; e < EB; ++e) {
    const int flat = e * 256 + tid;
    b_on[e] = (BG % 256 == 0) || flat < BG;
    b_kk[e] = BKF ? (flat % (BK / G)) * G : flat / (BN / (BKF ? 1 : G));
    b_nl[e] = BKF ? flat / (BK / G) : (flat % (BN / G)) * G;
  }
  // Loop-invariant halves of the address computation.
  Side a_m[AKF ? EA : 1], b_n[BKF ? EB : 1];
#pragma unroll
  for (int e = 0; e < (AKF ? EA : 1); ++e) a_m[e] = decode(p.am, m0 + a_ml[e]);
#pragma unroll
  for (int e = 0; e < (BKF ? EB : 1); ++e) b_n[e] = decode(p.bn, n0 + b_nl[e]);

  float ra[EA][G], rb[EB][G];
  // With m/n-fast scalar staging every lane of a wavefront works on the same k (tile rows are >= 64 wide), so the
  // k-side decode (two constant divisions + affine maps) is hoisted to the scalar unit via readfirstlane; the vector
  // unit only adds the per-lane half and tests the halo.
  constexpr bool A_UNIFORM_K = !VEC && !AKF && BM >= 64;
  constexpr bool B_UNIFORM_K = !VEC && !BKF && BN >= 64;
  // Loads are unconditional (the offset of an out-of-range element is clamped to 0) and their results are NOT touched
  // until stage(): the validity bits travel separately and the zeroing happens at the LDS write.  Any use of a loaded
  // value inside fetch() makes the compiler wait for the whole slice before the MFMAs it is meant to overlap with.
  uint32_t a_ok = 0u, b_ok = 0u;
  auto load_group = [&](const float* base, uint32_t off, bool ok, float (&dst)[G]) {
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(base + (ok ? off : 0u));
      dst[0] = v.x; dst[G > 1 ? 1 : 0] = v.y; dst[G > 2 ? 2 : 0] = v.z; dst[G > 3 ? 3 : 0] = v.w;
      if (G == 1) dst[0] = v.x;
    } else {
      dst[0] = base[ok ? off : 0u];
    }
  };
  auto fetch = [&](int k0) {
    a_ok = 0u; b_ok = 0u;
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const int k = k0 + ((A_UNIFORM_K) ? __builtin_amdgcn_readfirstlane(a_kk[e]) : a_kk[e]);
      const Side sk = decode(p.ak, k);
      const Side sm = a_m[AKF ? e : 0];
      const bool ok = a_on[e] & sk.valid & (k < kend) & sm.valid;
      a_ok |= (ok ? 1u : 0u) << e;
      load_group(p.A, sm.off + sk.off, ok, ra[e]);
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int k = k0 + ((B_UNIFORM_K) ? __builtin_amdgcn_readfirstlane(b_kk[e]) : b_kk[e]);
      const Side sk = decode(p.bk, k);
      const Side sn = b_n[BKF ? e : 0];
      bool ok = b_on[e] & sk.valid & (k < kend) & sn.valid;
      if (!VEC) ok = ok & ((uint32_t)(sk.h + sn.h) < (uint32_t)p.hlim) & ((uint32_t)(sk.w + sn.w) < (uint32_t)p.wlim);
      b_ok |= (ok ? 1u : 0u) << e;
      load_group(p.B, sk.off + sn.off, ok, rb[e]);
    }
  };
  using half_t = typename std::conditional<PREC == 2, _Float16, __bf16>::type;
  typedef half_t half4_t __attribute__((ext_vector_type(4)));
  constexpr int LDK = BK + 8;                                     // PREC: elements per operand-typed LDS row
  half_t* As16 = reinterpret_cast<half_t*>(lds);
  half_t* Bs16 = As16 + BM * LDK;
  static_assert(PREC == 0 || (BM + BN) * (BK + 8) * 2 <= (BK * LDA + BK * LDB) * 4, "the typed tiles fit the fp32 allocation");
  auto stage_typed = [&](half_t* tile, const float (&v)[G], int row, int kk, bool kfast) {
    if (kfast && VEC) {
      half4_t packed;
#pragma unroll
      for (int i = 0; i < 4; ++i) packed[i] = (half_t)v[G > i ? i : 0];
      *reinterpret_cast<half4_t*>(&tile[row * LDK + kk]) = packed;      // four consecutive k of one row: 8 aligned bytes
    } else if (kfast) {
      tile[row * LDK + kk] = (half_t)v[0];
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) tile[(row + i) * LDK + kk] = (half_t)v[i];   // G consecutive rows at one k
    }
  };
  auto stage = [&]() {
    if constexpr (PREC != 0) {
#pragma unroll
      for (int e = 0; e < EA; ++e) {
        if (!a_on[e]) continue;
        const bool ok = (a_ok >> e) & 1u;
        float v[G];
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = ok ? ra[e][i] : 0.f;
        stage_typed(As16, v, a_ml[e], a_kk[e], AKF);
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        if (!b_on[e]) continue;
        const bool ok = (b_ok >> e) & 1u;
        float v[G];
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = ok ? rb[e][i] : 0.f;
        stage_typed(Bs16, v, b_nl[e], b_kk[e], BKF);
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      if (!a_on[e]) continue;
      const bool ok = (a_ok >> e) & 1u;
      float v[G];
#pragma unroll
      for (int i = 0; i < G; ++i) v[i] = ok ? ra[e][i] : 0.f;
      if (VEC && !AKF) {
        *reinterpret_cast<float4*>(&As[a_kk[e] * LDA + a_ml[e]]) = make_float4(v[0], v[G > 1 ? 1 : 0], v[G > 2 ? 2 : 0],
                                                                                v[G > 3 ? 3 : 0]);
      } else {
#pragma unroll
        for (int i = 0; i < G; ++i) As[(a_kk[e] + (AKF ? i : 0)) * LDA + a_ml[e] + (AKF ? 0 : i)] = v[i];
      }
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      if (!b_on[e]) continue;
      const bool ok = (b_ok >> e) & 1u;
      float v[G];
#pragma unroll
      for (int i = 0; i < G; ++i) v[i] = ok ? rb[e][i] : 0.f;
      if (VEC && !BKF) {
        *reinterpret_cast<float4*>(&Bs[b_kk[e] * LDB + b_nl[e]]) = make_float4(v[0], v[G > 1 ? 1 : 0], v[G > 2 ? 2 : 0],
                                                                                v[G > 3 ? 3 : 0]);
      } else {
#pragma unroll
        for (int i = 0; i < G; ++i) Bs[(b_kk[e] + (BKF ? i : 0)) * LDB + b_nl[e] + (BKF ? 0 : i)] = v[i];
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  if (kbeg < kend) {
    fetch(kbeg);
    stage();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = (k0 + BK < kend) && !(p.debug & 1);
      if (more) fetch(k0 + BK);
      if constexpr (PREC != 0) {
        static_assert(PREC == 0 || BK % 16 == 0, "16-deep MFMA steps");
        using frag = typename std::conditional<PREC == 1, bf16x8, f16x8>::type;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
          frag a[MI], b[NI];                   // one ds_read_b128 each: 8 consecutive k of this lane's row / column
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            a[mi] = *reinterpret_cast<const frag*>(&As16[(wm0 + mi * 32 + l31) * LDK + kk + 8 * lhi]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            b[ni] = *reinterpret_cast<const frag*>(&Bs16[(wn0 + ni * 32 + l31) * LDK + kk + 8 * lhi]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              if constexpr (PREC == 1) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
              else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
      } else
#pragma unroll
      for (int kk = 0; kk < ((p.debug & 2) ? 0 : BK); kk += 2) {
        float a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = As[(kk + lhi) * LDA + wm0 + mi * 32 + l31];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = Bs[(kk + lhi) * LDB + wn0 + ni * 32 + l31];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        stage();
        __syncthreads();
      }
    }
  }

  // Epilogue: C/D fragment of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  int mode = p.mode;
  if (mode >= GG_ORDERED_STORE) {     // K split, ordered finish: the tile's last workgroup goes on with the sum of all slices
    constexpr int COUNT = MI * NI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.partial + (int64_t)blockIdx.x * gridDim.y * (COUNT * 256), (int)blockIdx.y,
                                          (int)gridDim.y, p.tickets + blockIdx.x,
                                          [&](int i) { return acc[i / (NI * 16)][(i / 16) % NI][i % 16]; },
                                          [&](int i, float v) { acc[i / (NI * 16)][(i / 16) % NI][i % 16] = v; }))
      return;
    mode -= GG_ORDERED_STORE;
  }
  const bool add_bias = p.bias != nullptr && (blockIdx.y == 0 || p.mode >= GG_ORDERED_STORE);
  Side sn[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) sn[ni] = decode(p.cn, n0 + wn0 + ni * 32 + l31);
  // bias values in front of every store (a load behind a store waits for it: one in-order vmcnt)
  float row_bias_all[MI][16], col_bias[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const Side sm = decode(p.cm, m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi);
      row_bias_all[mi][r] = (add_bias && !p.bias_cols && sm.valid) ? p.bias[sm.c] : 0.f;
    }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) col_bias[ni] = (add_bias && p.bias_cols && sn[ni].valid) ? p.bias[sn[ni].c] : 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      // accumulate (every weight gradient: the arena sums four backward passes): the 4 x NI old values of a register quad in
      // one batch in front of their stores -- loads and stores return through one in-order counter (vmcnt), so `*dst += v`
      // per element made every load wait for the store in front of it
      float previous[4][NI];
      if (mode == GG_ACCUMULATE) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = m0 + wm0 + mi * 32 + rr + 8 * rq + 4 * lhi;
          const Side sm = decode(p.cm, i);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            previous[rr][ni] = (sm.valid && sn[ni].valid) ? p.C[(uint32_t)(sm.off + sn[ni].off)] : 0.f;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * rq + rr;
        const int i = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const Side sm = decode(p.cm, i);
        if (!sm.valid) continue;
        const float row_bias = row_bias_all[mi][r];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (!sn[ni].valid) continue;
          const float v = (acc[mi][ni][r] + row_bias) + col_bias[ni];
          if (mode == GG_PARTIAL) {
            p.partial[((int64_t)blockIdx.y * p.M + i) * p.N + (n0 + wn0 + ni * 32 + l31)] = v;
            continue;
          }
          float* dst = p.C + (uint32_t)(sm.off + sn[ni].off);
          if (mode == GG_STORE) *dst = v;
          else if (mode == GG_ACCUMULATE) *dst = previous[rr][ni] + v;
          else unsafeAtomicAdd(dst, v);
        }
      }
    }
  }
}

// Direct form for very skinny outputs (M <= 2: single-channel map heads, count
// layers): one thread per C element, lanes along columns, A broadcast.  HBM/L2-bound, no matrix core.
__global__ __launch_bounds__(256) void gg_direct_kernel(const GatherGemm p) {
  const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int i = (int)blockIdx.y;
  const int kbeg = (int)blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const Side am = decode(p.am, i), cm = decode(p.cm, i);
  const Side bn = decode(p.bn, j), cn = decode(p.cn, j);
  if (!cn.valid || !cm.valid) return;
  float acc = 0.f;
  for (int k = kbeg; k < kend; ++k) {
    const Side ak = decode(p.ak, k), bk = decode(p.bk, k);
    acc = fmaf(gg_a(p, am, ak), gg_b(p, bk, bn), acc);
  }
  if (p.bias != nullptr && blockIdx.z == 0) acc += p.bias[p.bias_cols ? cn.c : cm.c];
  if (p.mode == GG_PARTIAL) {
    p.partial[((int64_t)blockIdx.z * p.M + i) * p.N + j] = acc;
    return;
  }
  float* dst = p.C + (uint32_t)(cm.off + cn.off);
  if (p.mode == GG_STORE) *dst = acc;
  else if (p.mode == GG_ACCUMULATE) *dst += acc;
  else unsafeAtomicAdd(dst, acc);
}

// A few output rows (M <= MR: RGB image gradients of the stem, the generator's 3-channel output layer) over a very
// wide N: the MFMA tile would spend 29 of its 32 rows on padding (4.5 TF/s measured).  One thread per column keeps
// MR accumulators.  Everything that depends only on k -- the decoded B-side offset / coordinates and the MR values
// of A -- is tabulated in LDS once per 256-k chunk and read back as broadcasts, so each gathered B element costs one
// vector load, two LDS broadcasts and ~12 VALU instructions, branch-free.
template <int MR>
__global__ __launch_bounds__(256) void gg_rows_kernel(const GatherGemm p) {
  constexpr int KC = 256;
  __shared__ int4 ktab[KC];                  // {B offset, h, w, -} of k
  __shared__ float atab[KC * MR];            // A[i, k] for the MR rows
  const int tid = (int)threadIdx.x;
  const int j = (int)blockIdx.x * 256 + tid;
  const bool col_ok = j < p.N;
  const int jc = col_ok ? j : p.N - 1;
  const Side bn = decode(p.bn, jc), cn = decode(p.cn, jc);
  const int kbeg = (int)blockIdx.y * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const bool lane_ok = col_ok && bn.valid;
  float acc[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) acc[i] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += KC) {
    __syncthreads();
    {
      const int k = k0 + tid;
      const bool k_ok = k < kend;
      const Side bk = decode(p.bk, k_ok ? k : kbeg), ak = decode(p.ak, k_ok ? k : kbeg);
      // an invalid k gets a row coordinate that fails every bounds test
      ktab[tid] = make_int4((int)bk.off, (k_ok && bk.valid) ? bk.h : -0x40000000, bk.w, 0);
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const Side am = decode(p.am, i < p.M ? i : 0);
        const bool ok = k_ok && i < p.M && am.valid && ak.valid;
        const float v = p.A[ok ? am.off + ak.off : 0u];
        atab[tid * MR + i] = ok ? v : 0.f;
      }
    }
    __syncthreads();
    const int kc = min(KC, kend - k0);
#pragma unroll 8
    for (int kk = 0; kk < kc; ++kk) {
      const int4 t = ktab[kk];
      // bitwise, not short-circuit: no divergent branches around the table reads
      const bool ok = (int)lane_ok & (int)((uint32_t)(t.y + bn.h) < (uint32_t)p.hlim) &
                      (int)((uint32_t)(t.z + bn.w) < (uint32_t)p.wlim);
      const float v = p.B[ok ? (uint32_t)t.x + bn.off : 0u];
      const float b = ok ? v : 0.f;
#pragma unroll
      for (int i = 0; i < MR; ++i) acc[i] = fmaf(atab[kk * MR + i], b, acc[i]);
    }
  }
  if (!col_ok || !cn.valid) return;
  const bool add_bias = p.bias != nullptr && blockIdx.y == 0;
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    if (i >= p.M) break;
    const Side cm = decode(p.cm, i);
    if (!cm.valid) continue;
    float v = acc[i];
    if (add_bias) v += p.bias[p.bias_cols ? cn.c : cm.c];
    if (p.mode == GG_PARTIAL) {
      p.partial[((int64_t)blockIdx.y * p.M + i) * p.N + j] = v;
      continue;
    }
    float* dst = p.C + (uint32_t)(cm.off + cn.off);
    if (p.mode == GG_STORE) *dst = v;
    else if (p.mode == GG_ACCUMULATE) *dst += v;
    else unsafeAtomicAdd(dst, v);
  }
}

// A tiny output (M, N <= 8) over a very long K (weight gradients of the k = stride map-head transposed convolutions:
// 8 x 4 outputs, K = all pixels of the batch): lanes along K.  Every thread walks its share of k with an M x N block
// of accumulators (M + N loads, M*N FMAs per k), the workgroup reduces the 64 sums once (shuffles, then LDS) and adds
// them with M*N atomics.  The MFMA tile did this at 0.3 TF/s: 32 x 128 tiles that are 97 % padding.
__global__ __launch_bounds__(256) void gg_dot_kernel(const GatherGemm p) {
  __shared__ float red[4][64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  Side am[8], bn[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { am[i] = decode(p.am, i < p.M ? i : 0); am[i].valid = am[i].valid && i < p.M; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { bn[j] = decode(p.bn, j < p.N ? j : 0); bn[j].valid = bn[j].valid && j < p.N; }
  const int stride = (int)gridDim.x * 256;
  for (int k = (int)blockIdx.x * 256 + tid; k < p.K; k += stride) {
    const Side ak = decode(p.ak, k), bk = decode(p.bk, k);
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = am[i].valid && ak.valid;
      const float v = p.A[ok ? am[i].off + ak.off : 0u];
      a[i] = ok ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = (int)bn[j].valid & (int)bk.valid & (int)((uint32_t)(bk.h + bn[j].h) < (uint32_t)p.hlim) &
                      (int)((uint32_t)(bk.w + bn[j].w) < (uint32_t)p.wlim);
      const float v = p.B[ok ? bk.off + bn[j].off : 0u];
      b[j] = ok ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
  }
  // wave reduction of each of the 64 sums, then the four waves through LDS
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[i][j];
#pragma unroll
      for (int offset = 32; offset > 0; offset >>= 1) v += __shfl_xor(v, offset, 64);
      if (lane == 0) red[wave][i * 8 + j] = v;
    }
  __syncthreads();
  if (tid < 64) {
    const int i = tid >> 3, j = tid & 7;
    if (i < p.M && j < p.N) {
      const Side cm = decode(p.cm, i), cn = decode(p.cn, j);
      if (cm.valid && cn.valid) {
        float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (p.bias != nullptr && blockIdx.x == 0) v += p.bias[p.bias_cols ? cn.c : cm.c];
        float* dst = p.C + (uint32_t)(cm.off + cn.off);
        if (p.mode == GG_PARTIAL) p.partial[((int64_t)blockIdx.x * p.M + i) * p.N + j] = v;
        else if (gridDim.x > 1) unsafeAtomicAdd(dst, v);
        else if (p.mode == GG_ACCUMULATE) *dst += v;
        else *dst = v;
      }
    }
  }
}

// Second stage of a split-K launch whose output is small (M*N up to a few thousand): thousands of workgroups adding into
// the same few cache lines serialise in L2 (measured: 310 us for a 20x48 output from 1024 K-slices), so the slices are
// stored to a workspace and summed here, no atomics, no pre-zeroing.  A workgroup owns 16 consecutive outputs; its 16
// slice-lanes per output walk the slices with four independent loads in flight each (a single dependent chain over 128
// slices cost 64 us of pure load latency), then 16-lane shuffle reductions.
__global__ __launch_bounds__(256) void gg_reduce_partials_kernel(const GatherGemm p, const float* __restrict__ ws) {
  const int o = (int)threadIdx.x & 15, lane_z = (int)threadIdx.x >> 4;      // 16 outputs x 16 slice-lanes
  const int64_t mn = (int64_t)p.M * p.N;
  const int64_t idx = (int64_t)blockIdx.x * 16 + o;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (idx < mn) {
    int z = lane_z;
    for (; z + 48 < p.split_k; z += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += ws[(int64_t)(z + 16 * u) * mn + idx];
    }
    for (; z < p.split_k; z += 16) acc[0] += ws[(int64_t)z * mn + idx];
  }
  float total = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  // lanes of one output are 16 apart (lane = lane_z * 16 + o): combine across lane_z within the wave, then across waves
  total += __shfl_xor(total, 16, 64);
  total += __shfl_xor(total, 32, 64);
  __shared__ float scratch[4][16];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  if (lane < 16) scratch[wave][lane] = total;
  __syncthreads();
  if (threadIdx.x >= 16 || idx >= mn) return;
  total = (scratch[0][o] + scratch[1][o]) + (scratch[2][o] + scratch[3][o]);
  const int i = (int)(idx / p.N), j = (int)(idx - (int64_t)i * p.N);
  const Side sm = decode(p.cm, i), sn = decode(p.cn, j);
  if (!sm.valid || !sn.valid) return;
  float* dst = p.C + (uint32_t)(sm.off + sn.off);
  *dst = p.mode == GG_ACCUMULATE ? *dst + total : total;
}

// The same second stage for outputs of any size (weight gradients of the strided / transposed convolutions: up to a few
// hundred thousand outputs x up to a few hundred slices): one thread per output, consecutive threads on consecutive
// outputs (coalesced over every slice), the slices added in slice order -- four loads in flight, one fixed order.
__global__ __launch_bounds__(256) void gg_reduce_partials_wide_kernel(const GatherGemm p, const float* __restrict__ ws) {
  const int64_t mn = (int64_t)p.M * p.N;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= mn) return;
  float total = ws[idx];
  int z = 1;
  for (; z + 3 < p.split_k; z += 4) {
    const float a = ws[(int64_t)z * mn + idx], b = ws[(int64_t)(z + 1) * mn + idx];
    const float c = ws[(int64_t)(z + 2) * mn + idx], d = ws[(int64_t)(z + 3) * mn + idx];
    total = (((total + a) + b) + c) + d;
  }
  for (; z < p.split_k; ++z) total += ws[(int64_t)z * mn + idx];
  const int i = (int)(idx / p.N), j = (int)(idx - (int64_t)i * p.N);
  const Side sm = decode(p.cm, i), sn = decode(p.cn, j);
  if (!sm.valid || !sn.valid) return;
  float* dst = p.C + (uint32_t)(sm.off + sn.off);
  *dst = p.mode == GG_ACCUMULATE ? *dst + total : total;
}

__device__ unsigned int g_gg_split_tickets[SPLIT_TICKET_SETS * SPLIT_TICKET_TILES];

// Partial sums that a second small kernel combines go through a workspace the CALLER owns (srgan_set_workspace: one
// block per (device, stream), at least srgan_workspace_bytes() long; launches on one stream are ordered, so reuse is
// safe): the K-slices of split-K launches with a tiny output (M*N < 512 outputs x at most 1024 slices x 4 B = 2 MiB) and
// the per-workgroup batch-norm parameter sums of the fused data-gradient epilogues (2 x column blocks x channels x 4 B =
// at most 1/32 of the gradient tensor's bytes: 64 MiB covers tensors of up to 2^29 elements; larger ones report "unsupported"
// from srgan_conv2d_bnrelu_supported and take the two-kernel form).  The library never allocates device memory.
constexpr size_t WORKSPACE_BYTES = (size_t)256 << 20;     // (round 5: 256 MiB -- the partial tiles of every K split / grouped weight gradient live here)
struct WorkspaceSlot { float* ptr = nullptr; size_t bytes = 0; int index = -1; };
static std::mutex g_workspace_mutex;
static std::map<std::pair<int, hipStream_t>, WorkspaceSlot> g_workspaces;
static int g_workspace_count[16] = {};        // ticket sets handed out, per device (the ticket arrays are per-device symbols)

// Unregistering (ptr == NULL) keeps the slot and its ticket-set index: a stream that registers again -- a caller that tears its
// workspace down between phases -- gets the SAME set back, so the 64 sets per device are not used up by re-registration (ADVICE
// r5: the index used to be re-drawn, and past 64 the ordered finish silently fell back to fp32 atomics).  A stream beyond the
// 64th of a device has index >= SPLIT_TICKET_SETS and srgan_split_is_ordered() reports 0 for it.
int workspace_register(float* ptr, size_t bytes, hipStream_t stream) {
  int device = 0;
  SRGAN_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_workspace_mutex);
  WorkspaceSlot& slot = g_workspaces[{device, stream}];
  slot.ptr = ptr; slot.bytes = ptr ? bytes : 0;
  if (slot.index < 0 && ptr != nullptr) slot.index = g_workspace_count[device & 15]++;
  return SRGAN_OK;
}

// A small integer per registered (device, stream): kernels that keep a few words of device-global state between the
// workgroups of ONE launch (reduce.hip's row tickets) index it by this, so that streams never share a set.
int workspace_index(hipStream_t stream) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(g_workspace_mutex);
  auto found = g_workspaces.find({device, stream});
  return (found == g_workspaces.end() || found->second.ptr == nullptr) ? -1 : found->second.index;
}

size_t workspace_capacity() { return WORKSPACE_BYTES; }

float* partial_workspace(size_t bytes, hipStream_t stream) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_workspace_mutex);
  auto found = g_workspaces.find({device, stream});
  if (found == g_workspaces.end() || found->second.ptr == nullptr || found->second.bytes < bytes) return nullptr;
  return found->second.ptr;
}

// ------------------------------------------------------------------------------------------- launcher
struct GGConfig { int kind; int bm, bn; int tiles; };   // kind 0 direct, 1 mfma

static GGConfig choose_config(const GatherGemm& p, int force) {
  GGConfig c;
  // M <= 2 (single-channel map heads, count layers): the direct form; from 3 rows up (RGB image gradients, generator
  // output) the MFMA tile wins even at 3/32 row utilisation because the direct form is VALU/latency bound.
  if ((p.M <= 2 && force != 2) || force == 1) {
    c.kind = 0; c.bm = 1; c.bn = 256;
    c.tiles = p.M * ((p.N + 255) / 256);
    return c;
  }
  static const bool no_dot = getenv("SRGAN_NO_DOT") != nullptr;
  if (force == 0 && !no_dot && p.M <= 8 && p.N <= 8 && p.K >= 65536) {     // kind 9: lanes along K
    c.kind = 9; c.bm = 8; c.bn = 8;
    c.tiles = 1;
    return c;
  }
  static const bool no_rows = getenv("SRGAN_NO_ROWS") != nullptr;
  if (force == 0 && !no_rows && p.M <= 8 && p.N >= 4096) {     // kind 5: the few-rows kernel
    c.kind = 5; c.bm = p.M <= 4 ? 4 : 8; c.bn = 256;
    c.tiles = (p.N + 255) / 256;
    return c;
  }
  c.kind = 1;
  // Largest tile that still yields >= target workgroups (4 per CU: staging of one hides under the MFMAs of the
  // others); otherwise the smallest tile of the class, topped up by split-K in choose_split.
  static const int target = getenv("SRGAN_TILE_TARGET") ? atoi(getenv("SRGAN_TILE_TARGET")) : 1024;
  static const int candidates[3][3][2] = {{{128, 128}, {128, 64}, {64, 64}},     // M > 64
                                          {{64, 128}, {64, 64}, {64, 64}},        // 32 < M <= 64
                                          {{32, 256}, {32, 128}, {32, 128}}};     // M <= 32
  const int cls = p.M > 64 ? 0 : (p.M > 32 ? 1 : 2);
  // Long-K problems (weight gradients: K = pixels) get their parallelism from split-K anyway, so they take the
  // largest tile: at 64 x 64 the operand stream (16 FLOP/B) is HBM-bound at ~80 TF/s.
  static const bool big_tiles = getenv("SRGAN_NO_BIG_TILES") == nullptr;
  if (big_tiles && p.K >= 131072 && p.M >= 96 && p.N >= 96) {
    c.bm = 128; c.bn = 128;
    c.tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    return c;
  }
  static const int first = getenv("SRGAN_GG_FIRST_TILE") ? atoi(getenv("SRGAN_GG_FIRST_TILE")) : 0;
  for (int i = first; i < 3; ++i) {
    c.bm = candidates[cls][i][0];
    c.bn = candidates[cls][i][1];
    c.tiles = ((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn);
    if (c.tiles >= target) break;
  }
  return c;
}

static void choose_split(GatherGemm& p, const GGConfig& c, bool allow_split) {
  p.split_k = 1;
  p.k_per_split = p.K > 0 ? ((p.K + GG_BK - 1) / GG_BK) * GG_BK : GG_BK;
  static const int target = getenv("SRGAN_TILE_TARGET") ? atoi(getenv("SRGAN_TILE_TARGET")) : 1024;
  if (!allow_split || p.K < 128 || c.tiles * 4 >= target * 3) return;
  const int want = (target + c.tiles - 1) / c.tiles;
  const int max_split = p.K / 64;
  int split = want < max_split ? want : max_split;
  if (split <= 1) return;
  int per = (p.K + split - 1) / split;
  per = ((per + GG_BK - 1) / GG_BK) * GG_BK;
  p.k_per_split = per;
  p.split_k = (p.K + per - 1) / per;
}

template <int BM, int BN, int WGM, bool VEC>
static void launch_mfma_v(const GatherGemm& p, dim3 grid, hipStream_t stream) {
  if (p.a_kfast && p.b_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, true, true, VEC>), grid, dim3(256), 0, stream, p);
  else if (p.a_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, true, false, VEC>), grid, dim3(256), 0, stream, p);
  else if (p.b_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, false, true, VEC>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, false, false, VEC>), grid, dim3(256), 0, stream, p);
}

// Can a Dec3 be walked in aligned groups of 4 consecutive indices with unit element stride?
static bool dec_vec_fast(const Dec3& d) {
  const int32_t B = (int32_t)d.div_b.d, A = (int32_t)(d.div_ab.d / d.div_b.d);
  if (d.limit % 4 || d.off0 % 4) return false;
  if (B > 1) return d.off_b == 1 && B % 4 == 0 && d.off_a % 4 == 0 && d.off_c % 4 == 0;
  if (A > 1) return d.off_a == 1 && A % 4 == 0 && d.off_c % 4 == 0;
  return d.off_c == 1;
}

static bool dec_vec_slow(const Dec3& d) {
  const int32_t B = (int32_t)d.div_b.d, A = (int32_t)(d.div_ab.d / d.div_b.d);
  // strides of extent-1 components are never applied
  return d.off_c % 4 == 0 && (A == 1 || d.off_a % 4 == 0) && (B == 1 || d.off_b % 4 == 0) && d.off0 % 4 == 0;
}

// Both operands 16-byte stageable (1x1 convolutions / linear layers on contiguous data, no halo).
static bool vec_eligible(const GatherGemm& p) {
  static const bool disabled = getenv("SRGAN_NO_VEC") != nullptr;
  if (disabled || p.hlim != 1 || p.wlim != 1) return false;
  if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
  const bool a = p.a_kfast ? (dec_vec_fast(p.ak) && dec_vec_slow(p.am)) : (dec_vec_fast(p.am) && dec_vec_slow(p.ak));
  const bool b = p.b_kfast ? (dec_vec_fast(p.bk) && dec_vec_slow(p.bn)) : (dec_vec_fast(p.bn) && dec_vec_slow(p.bk));
  return a && b && p.K % 4 == 0;
}

template <int BM, int BN, int WGM, int PREC>
static void launch_mfma_mixed(const GatherGemm& p, dim3 grid, hipStream_t stream) {
  if (p.a_kfast && p.b_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, true, true, false, PREC>), grid, dim3(256), 0, stream, p);
  else if (p.a_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, true, false, false, PREC>), grid, dim3(256), 0, stream, p);
  else if (p.b_kfast) hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, false, true, false, PREC>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((gg_mfma_kernel<BM, BN, WGM, false, false, false, PREC>), grid, dim3(256), 0, stream, p);
}

template <int BM, int BN, int WGM>
static void launch_mfma(const GatherGemm& p, dim3 grid, hipStream_t stream) {
  if (p.precision == 1) launch_mfma_mixed<BM, BN, WGM, 1>(p, grid, stream);        // (scalar staging only)
  else if (p.precision == 2) launch_mfma_mixed<BM, BN, WGM, 2>(p, grid, stream);
  else if (vec_eligible(p)) launch_mfma_v<BM, BN, WGM, true>(p, grid, stream);
  else launch_mfma_v<BM, BN, WGM, false>(p, grid, stream);
}

// Plans one launch: fills split_k / k_per_split; returns whether the launch needs a zeroed (or live) C because
// it combines K-slices with atomics.
// The launch's accumulators per workgroup for the ordered finish (see gg_mfma_kernel: MI x NI fragments of 16 per lane).
static int64_t mfma_tile_floats(const GGConfig& c) { return (int64_t)c.bm * c.bn; }

// Plans one launch: fills split_k / k_per_split and how its K slices are combined (p.use_partial, a SplitCombine); returns
// whether the launch needs a zeroed (or live) C because it combines them with fp32 atomics -- only when the stream has
// no workspace (or SRGAN_ATOMIC_SPLIT=1): otherwise the slices meet in a fixed order, through the workspace.
bool gg_prepare(GatherGemm& p, int force, GGConfig* out, hipStream_t stream) {
  static const int debug = getenv("SRGAN_GG_DEBUG") ? atoi(getenv("SRGAN_GG_DEBUG")) : 0;
  p.debug = debug;
  p.tickets = nullptr;
  GGConfig c = choose_config(p, force);
  if (c.kind == 9) {                       // workgroups of 256 k-lanes, ~16 k per thread
    int groups = (p.K + 4095) / 4096;
    if (groups > 1024) groups = 1024;
    p.split_k = groups; p.k_per_split = p.K;
  } else {
    choose_split(p, c, true);
  }
  if (out) *out = c;
  p.use_partial = GG_COMBINE_ATOMIC;
  if (p.split_k <= 1) return false;
  static const bool no_partial = getenv("SRGAN_NO_PARTIAL") != nullptr;
  const int64_t mn = (int64_t)p.M * p.N;
  const bool fits = (size_t)mn * p.split_k * sizeof(float) <= WORKSPACE_BYTES;
  // (up to 4096 outputs: at 960 outputs x 1024 slices the atomics still serialised -- 310 us for 36 MB of operands)
  if (!no_partial && c.kind == 1 && p.split_k >= 16 && mn <= 4096 && fits) {
    p.use_partial = GG_COMBINE_PARTIAL_TINY;
    return false;
  }
  const bool have_workspace = !split_atomics_forced() && partial_workspace(1, stream) != nullptr;
  if (have_workspace) {
    // few slices of an MFMA launch: the tile's last workgroup adds them (one launch); many slices, or the VALU kernels:
    // partial outputs + a second kernel that adds the slices of every output in slice order
    int set = -1;
    if (c.kind == 1 && p.split_k <= 16 &&
        split_workspace(c.tiles, p.split_k, mfma_tile_floats(c), 0, stream, &set) != nullptr) {
      p.use_partial = GG_COMBINE_ORDERED;
      return false;
    }
    if (fits) {
      p.use_partial = (c.kind == 1 && mn <= 4096) ? GG_COMBINE_PARTIAL_TINY : GG_COMBINE_PARTIAL_WIDE;
      return false;
    }
  }
  return true;
}

// ---- optional live timing of every contraction launch with HIP events on the launch stream -------------------
// (bench.py: roofline.achieved = sum of logical 2*M*N*K over launches / sum of their event-timed durations)
struct ProfileRecord { int32_t M, N, K, kind, bm, bn, split, akf, bkf; double bytes; int32_t precision; };
// Algorithmic HBM bytes of one contraction launch: every operand element once, 4 B each.  `taps` = how many times the
// gather visits one element of the big operand (9 for a 3x3 kernel, R*S of a strided one divided over its classes).
static double algorithmic_bytes(double M, double N, double K, int kind) {
  if (kind == 2) return 4.0 * (M * N + M * K + (K / 9.0) * N);     // conv3x3: K = CI * 9 taps over one input plane
  if (kind == 4) return 4.0 * (M * N + M * K + (N / 9.0) * K);     // conv3x3 weight gradient: N = CI * 9
  return 4.0 * (M * N + M * K + K * N);
}
struct ProfileState {
  std::atomic<bool> enabled{false};
  std::mutex mutex;                   // guards every other member (launches may come from several host threads)
  std::vector<hipEvent_t> events;     // two per slot: start, stop
  std::vector<ProfileRecord> records; // one per slot (M < 0: begun, not ended)
  size_t slots = 0;
  double flops = 0.0, mfma_flops = 0.0, bytes = 0.0;
  int64_t launches = 0;
};
static ProfileState g_profile;

static int gg_launch_unprofiled(const GatherGemm& p, const GGConfig& c, hipStream_t stream);

// Event bracket around one contraction launch (also called from the other translation units).  begin() reserves a slot
// and records its start event on the launch stream; returns -1 when profiling is off.
int profile_bracket_begin(hipStream_t stream) {
  if (!g_profile.enabled.load(std::memory_order_acquire)) return -1;
  hipEvent_t start;
  int slot;
  {
    std::lock_guard<std::mutex> lock(g_profile.mutex);
    slot = (int)g_profile.slots++;
    while (g_profile.events.size() < 2 * g_profile.slots) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { --g_profile.slots; return -1; }
      g_profile.events.push_back(e);
    }
    g_profile.records.resize(g_profile.slots, ProfileRecord{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0});
    start = g_profile.events[2 * slot];
  }
  if (hipEventRecord(start, stream) != hipSuccess) return -1;
  return slot;
}

int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn, int split,
                        int akf, int bkf, int64_t b_unique, int precision) {
  if (slot < 0) return SRGAN_OK;
  hipEvent_t stop;
  {
    std::lock_guard<std::mutex> lock(g_profile.mutex);
    if ((size_t)slot >= g_profile.slots) return SRGAN_OK;          // the region was closed in between
    stop = g_profile.events[2 * slot + 1];
    const double f = 2.0 * (double)M * (double)N * (double)K;
    const double bytes = b_unique > 0 ? 4.0 * ((double)M * N + (double)M * K + (double)b_unique)
                                      : algorithmic_bytes((double)M, (double)N, (double)K, kind);
    g_profile.records[slot] = ProfileRecord{(int32_t)M, (int32_t)N, (int32_t)K, kind, bm, bn, split, akf, bkf, bytes, precision};
    g_profile.flops += f;
    if (kind != 0 && kind != 5 && kind != 9 && kind != 12) g_profile.mfma_flops += f;      // direct / few-rows / lanes-along-K / stem data gradient: VALU
    g_profile.bytes += bytes;
    g_profile.launches += 1;
  }
  SRGAN_HIP(hipEventRecord(stop, stream));
  return SRGAN_OK;
}

// The same bracket end for a launch that states its own algorithmic bytes (the 16-bit kernels of blocked16*.hip: kinds 14
// hconv3x3, 15 hwgrad3x3, 16 hgemm, 17 hlinear_wgrad, 18 hconv4x4s2, 19 hwgrad4x4s2).
int profile_bracket_end_bytes(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn, int split,
                              double bytes, int precision) {
  if (slot < 0) return SRGAN_OK;
  hipEvent_t stop;
  {
    std::lock_guard<std::mutex> lock(g_profile.mutex);
    if ((size_t)slot >= g_profile.slots) return SRGAN_OK;
    stop = g_profile.events[2 * slot + 1];
    const double f = 2.0 * (double)M * (double)N * (double)K;
    g_profile.records[slot] = ProfileRecord{(int32_t)M, (int32_t)N, (int32_t)K, kind, bm, bn, split, 0, 0, bytes, precision};
    g_profile.flops += f;
    g_profile.mfma_flops += f;
    g_profile.bytes += bytes;
    g_profile.launches += 1;
  }
  SRGAN_HIP(hipEventRecord(stop, stream));
  return SRGAN_OK;
}

bool conv3x3_enabled();
int conv3x3_run(const float* in, int64_t in_bs, const float* w, int32_t w_base, int32_t w_so, int32_t w_si, int32_t w_skh,
                int32_t w_skw, const float* bias, float* out, int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t H,
                int32_t W, int accumulate, hipStream_t stream, const float* const* bn = nullptr,
                const BnBackwardEpilogue* epilogue = nullptr, int precision = 0, const struct Conv3Placement* placement = nullptr);
struct Conv3Placement { int32_t taps, out_plane, out_sy, out_sx, out_off; };
bool conv3x3_epilogue_supported(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W);
int64_t conv3x3_epilogue_tiles(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W);
int64_t pointwise_epilogue_tiles(int32_t N, int32_t HW);
int bn_partial_reduce_batched_run(const void* jobs, int count, int max_channels, int max_tiles, const float* scratch,
                                  hipStream_t stream);

bool pointwise_enabled();
int pointwise_run(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, const float* bias, float* out,
                  int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t HW, int accumulate, hipStream_t stream,
                  const float* const* bn = nullptr, const BnBackwardEpilogue* epilogue = nullptr,
                  int* plan_only_split = nullptr);
int conv3x3_splits(int32_t N, int32_t CI, int32_t CO, int32_t H, int32_t W, int precision = 0);

// 1x1 / stride 1 / unpadded: the register-streamed pointwise kernel (any plane size: an image's last 32-pixel group may
// be ragged; planes of fewer than 32 pixels stay on the generic kernel).
static bool use_pointwise(const ConvGeom& g, int out_channels, int force) {
  const int in_channels = out_channels == g.K ? g.C : g.K;     // forward: C -> K; data gradient: K -> C
  static const bool ragged = getenv("SRGAN_PW_NO_RAGGED") == nullptr;
  const bool plane_ok = (g.H * g.W) % 32 == 0 || (ragged && g.H * g.W >= 32);
  return force == 0 && pointwise_enabled() && pointwise(g) && plane_ok && out_channels >= 8 && in_channels % 2 == 0;
}

// 3x3 / stride 1 / pad 1 with enough width to fill most of a 16-pixel tile row (two image rows share a 32-pixel MFMA
// column block): the LDS-halo kernel.
static bool use_conv3x3(const ConvGeom& g, int out_channels, int force) {
  static const int min_width = getenv("SRGAN_CONV3_MIN_W") ? atoi(getenv("SRGAN_CONV3_MIN_W")) : 7;   // the 14- and 7-wide planes at 224
  return force == 0 && conv3x3_enabled() && g.R == 3 && g.S == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 &&
         g.W >= min_width && out_channels >= 8;
}

// Mixed precision also takes the 4 x 4 planes (whole images side by side in a tile: conv3x3_mixed_small_kernel) and
// fewer than 8 output channels -- the gradient with respect to a 3-channel image, which the penalty chain of the VGG / DCGAN
// discriminators asks for (reference srgan.py:366-370): 29 of a 32-row tile's rows are padding there, and the generic
// kernel's gather still made it six times slower (5.9 TF/s on [3 x 524 288] x 576).
static bool use_conv3x3_mixed(const ConvGeom& g, int out_channels) {
  static const bool no_small = getenv("SRGAN_NO_CONV3_SMALL") != nullptr;
  if (use_conv3x3(g, out_channels, 0)) return true;
  const bool geometry = conv3x3_enabled() && g.R == 3 && g.S == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1;
  if (no_small || !geometry) return false;
  static const int min_width = getenv("SRGAN_CONV3_MIN_W") ? atoi(getenv("SRGAN_CONV3_MIN_W")) : 7;
  return (g.H == 4 && g.W == 4 && out_channels >= 8) || (g.W >= min_width && out_channels >= 1);
}

bool stem7x7_enabled();
bool stem7x7_geometry(int32_t C, int32_t K, int32_t R, int32_t S, int32_t sh, int32_t sw, int32_t ph, int32_t pw);
int stem7x7_fwd_run(const float* x, int64_t x_bs, const float* w, float* y, int64_t y_bs, int32_t N, int32_t H, int32_t W,
                    int32_t K, int32_t OH, int32_t OW, hipStream_t stream);
int stem7x7_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t H, int32_t W,
                      int32_t K, int32_t OH, int32_t OW, int accumulate, hipStream_t stream);

int stem7x7_bwd_data_run(const float* gy, int64_t gy_bs, const float* w, float* gx, int64_t gx_bs, int32_t N, int32_t H,
                         int32_t W, int32_t K, int32_t OH, int32_t OW, hipStream_t stream);
bool conv3x3_wgrad_enabled();
int conv3x3_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t CI,
                      int32_t CO, int32_t H, int32_t W, int accumulate, hipStream_t stream, const float* const* bn = nullptr,
                      int precision = 0);

// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with at least one wave's worth of input channels: the
// LDS-patch kernel of conv3x3_wgrad.hip.
// `aligned_only`: widths that are a multiple of 4 with 16-byte aligned rows (the float4 staging; the mixed-precision
// variants have no other); otherwise the kernel's ragged variant also takes the 14- and 7-wide planes of 224 x 224.
static bool wgrad3x3_geometry(const ConvGeom& g, int min_width = 7, bool aligned_only = false) {
  static const bool no_ragged = getenv("SRGAN_WGRAD3_NO_RAGGED") != nullptr;
  const bool aligned = g.W % 4 == 0 && g.x_bs % 4 == 0 && g.y_bs % 4 == 0;
  if (!aligned && (aligned_only || no_ragged)) return false;
  if (no_ragged && !aligned_only) min_width = 16;              // the selection before the ragged variant existed
  return g.R == 3 && g.S == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.W >= min_width && g.C >= 32;
}

static bool use_wgrad3x3(const ConvGeom& g, const float* x, const float* gy, int force) {
  return force == 0 && conv3x3_wgrad_enabled() && wgrad3x3_geometry(g);
}

bool pointwise_ksplit_wanted(int32_t N, int32_t K, int32_t M, int32_t HW, bool fused_bn);
int pointwise_ksplit_run(const float* in, int64_t in_bs, const float* w, const float* bias, float* out, int64_t out_bs,
                         int32_t N, int32_t K, int32_t M, int32_t HW, int accumulate, hipStream_t stream,
                         const float* const* bn);

bool pointwise_wgrad_enabled();
int pointwise_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t CI,
                        int32_t CO, int32_t HW, int accumulate, hipStream_t stream, const float* const* bn = nullptr);

int pointwise_wgrad_group_plan(int64_t x_off, int64_t x_bs, int64_t gy_off, int64_t gy_bs, float* gw, int64_t gw_off, int32_t N, int32_t CI,
                               int32_t CO, int32_t HW, const float* const* bn, int32_t group, int64_t group_weights, int64_t partial_offset,
                               void* job_out, int32_t* grid_x, int32_t* grid_y, int32_t* ragged, int64_t* partial_floats);
int pointwise_wgrad_group_run(const void* jobs, int32_t count, int32_t grid_x, int32_t grid_y, int32_t ragged, int32_t fused_bn,
                              const float* x_base, const float* gy_base, float* gw_base, int64_t flops_mn, int64_t pixels,
                              int64_t elements, int64_t partial_floats, hipStream_t stream);
int conv3x3_wgrad_group_plan(int64_t x_off, int64_t x_bs, int64_t gy_off, int64_t gy_bs, float* gw, int64_t gw_off, int32_t N, int32_t CI,
                             int32_t CO, int32_t H, int32_t W, const float* const* bn, int32_t group, int64_t partial_offset, void* job_out,
                             int32_t* grid_x, int32_t* grid_y, int32_t* ragged, int64_t* partial_floats);
int conv3x3_wgrad_group_run(const void* jobs, int32_t count, int32_t grid_x, int32_t grid_y, int32_t ragged,
                            const float* x_base, const float* gy_base, float* gw_base, int64_t flops_mn, int64_t pixels,
                            int64_t elements, int64_t partial_floats, hipStream_t stream);

static bool pointwise_wgrad_geometry(const ConvGeom& g) {
  static const bool no_ragged = getenv("SRGAN_PWG_NO_RAGGED") != nullptr;
  if (no_ragged && ((g.H * g.W) % 32 != 0 || g.x_bs % 4 != 0 || g.y_bs % 4 != 0)) return false;
  return pointwise(g) && g.H * g.W >= 32 && g.C >= 16 && g.K >= 16;
}

// Weight gradient of a 1x1 / stride 1 convolution: the register-streamed kernel of pointwise_wgrad.hip (whole 32-pixel
// chunks with 16-byte aligned rows, or its ragged variant).
static bool use_pointwise_wgrad(const ConvGeom& g, const float* x, const float* gy, int force) {
  return force == 0 && pointwise_wgrad_enabled() && pointwise_wgrad_geometry(g);
}

int gg_launch(const GatherGemm& p, const GGConfig& c, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0) return SRGAN_OK;
  const int slot = profile_bracket_begin(stream);
  const int status = gg_launch_unprofiled(p, c, stream);
  profile_bracket_end(slot, stream, p.M, p.N, p.K, c.kind, c.bm, c.bn, p.split_k, p.a_kfast, p.b_kfast, p.b_unique,
                      c.kind == 1 ? p.precision : 0);
  return status;
}

static int gg_launch_unprofiled(const GatherGemm& p, const GGConfig& c, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0) return SRGAN_OK;
  if (p.split_k > 1 && (p.use_partial == GG_COMBINE_PARTIAL_TINY || p.use_partial == GG_COMBINE_PARTIAL_WIDE)) {
    // every slice stores its partial output to the workspace, a second kernel adds the slices in slice order
    SRGAN_REQUIRE(p.mode == GG_STORE || p.mode == GG_ACCUMULATE, SRGAN_EINVAL, "partial-sum launch mode");
    const int64_t mn = (int64_t)p.M * p.N;
    float* ws = partial_workspace((size_t)mn * p.split_k * sizeof(float), stream);
    SRGAN_REQUIRE(ws != nullptr, SRGAN_EINVAL,
                  "split-K workspace: call srgan_set_workspace(ptr, >= srgan_workspace_bytes(), stream) for this stream first");
    GatherGemm q = p;
    q.mode = GG_PARTIAL; q.partial = ws; q.use_partial = GG_COMBINE_ATOMIC;
    const int status = gg_launch_unprofiled(q, c, stream);
    if (status != SRGAN_OK) return status;
    if (p.use_partial == GG_COMBINE_PARTIAL_TINY)
      hipLaunchKernelGGL(gg_reduce_partials_kernel, dim3((unsigned)((mn + 15) / 16)), dim3(256), 0, stream, p, ws);
    else
      hipLaunchKernelGGL(gg_reduce_partials_wide_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, stream, p, ws);
    return launch_status();
  }
  if (p.split_k > 1 && p.use_partial == GG_COMBINE_ORDERED) {
    SRGAN_REQUIRE(c.kind == 1 && (p.mode == GG_STORE || p.mode == GG_ACCUMULATE), SRGAN_EINVAL, "ordered split launch mode");
    int set = -1;
    float* ws = split_workspace(c.tiles, p.split_k, mfma_tile_floats(c), 0, stream, &set);
    unsigned int* tickets = ws ? device_tickets(g_gg_split_tickets) : nullptr;
    SRGAN_REQUIRE(ws != nullptr && tickets != nullptr, SRGAN_EINVAL, "ordered split-K: the stream's workspace went away");
    GatherGemm q = p;
    q.mode = p.mode == GG_ACCUMULATE ? GG_ORDERED_ACCUMULATE : GG_ORDERED_STORE;
    q.partial = ws; q.tickets = tickets + (size_t)set * SPLIT_TICKET_TILES; q.use_partial = GG_COMBINE_ATOMIC;
    return gg_launch_unprofiled(q, c, stream);
  }
  if (c.kind == 0) {
    dim3 grid((p.N + 255) / 256, p.M, p.split_k);
    SRGAN_REQUIRE(p.M <= 65535 && p.split_k <= 65535, SRGAN_ERANGE, "direct gather-gemm grid");
    hipLaunchKernelGGL(gg_direct_kernel, grid, dim3(256), 0, stream, p);
    return launch_status();
  }
  dim3 grid(c.tiles, p.split_k, 1);
  SRGAN_REQUIRE(p.split_k <= 65535, SRGAN_ERANGE, "split-k grid");
  if (c.kind == 9) {
    hipLaunchKernelGGL(gg_dot_kernel, dim3(p.split_k), dim3(256), 0, stream, p);
    return launch_status();
  }
  if (c.kind == 5) {
    if (c.bm == 4) hipLaunchKernelGGL(gg_rows_kernel<4>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gg_rows_kernel<8>, grid, dim3(256), 0, stream, p);
    return launch_status();
  }
  if (c.bm == 128 && c.bn == 128) launch_mfma<128, 128, 2>(p, grid, stream);
  else if (c.bm == 128 && c.bn == 64) launch_mfma<128, 64, 2>(p, grid, stream);
  else if (c.bm == 64 && c.bn == 128) launch_mfma<64, 128, 2>(p, grid, stream);
  else if (c.bm == 64 && c.bn == 64) launch_mfma<64, 64, 2>(p, grid, stream);
  else if (c.bm == 32 && c.bn == 256) launch_mfma<32, 256, 1>(p, grid, stream);
  else launch_mfma<32, 128, 1>(p, grid, stream);
  return launch_status();
}

// Runs a group of plans that together define one dense output tensor of c_elems floats.  `accumulate` adds
// into the existing contents; otherwise the output is (over)written.  force: 0 auto, 1 direct, 2 mfma.
int gg_run_group(std::vector<GatherGemm>& plans, float* c_base, int64_t c_elems, int accumulate, int force,
                 hipStream_t stream) {
  std::vector<GGConfig> configs(plans.size());
  bool any_atomic = false;
  for (size_t i = 0; i < plans.size(); ++i) any_atomic |= gg_prepare(plans[i], force, &configs[i], stream);
  if (any_atomic && !accumulate) if (const int status = zero_floats(c_base, c_elems, stream)) return status;
  for (size_t i = 0; i < plans.size(); ++i) {
    GatherGemm& p = plans[i];
    if (p.split_k > 1 && p.use_partial == GG_COMBINE_ATOMIC) p.mode = GG_ATOMIC;
    else p.mode = accumulate ? GG_ACCUMULATE : GG_STORE;
    const int status = gg_launch(p, configs[i], stream);
    if (status != SRGAN_OK) return status;
  }
  return SRGAN_OK;
}

}  // namespace srgan

// ------------------------------------------------------------------------------------------- C ABI
using namespace srgan;

extern "C" {

int srgan_split_is_ordered(void* stream);

struct srgan_conv_desc {
  int32_t N, C, H, W, K, R, S, stride_h, stride_w, pad_h, pad_w, OH, OW;
  int64_t x_batch_stride, y_batch_stride;
  int32_t compute_dtype;      // 0 fp32, 1 bf16, 2 fp16 MFMA operands (fp32 data and accumulation)
};

static bool dtype_ok(int dtype) { return dtype >= 0 && dtype <= 2; }

static bool to_geom(const srgan_conv_desc* d, ConvGeom& g) {
  if (d == nullptr) return false;
  g.N = d->N; g.C = d->C; g.H = d->H; g.W = d->W; g.K = d->K; g.R = d->R; g.S = d->S;
  g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w; g.OH = d->OH; g.OW = d->OW;
  g.x_bs = d->x_batch_stride; g.y_bs = d->y_batch_stride;
  return geom_ok(g);
}

// Descriptor -> geometry with the two failure classes kept apart: a malformed descriptor (-1) and one whose tensors pass
// the 2^31-element limit of the 32-bit offsets (-3, with the extent in the message so that the caller can split the batch).
#define SRGAN_GEOM(desc, g, what)                                                                        \
  do {                                                                                                   \
    SRGAN_REQUIRE(to_geom(desc, g), SRGAN_EINVAL, what " geometry");                                     \
    if (geom_largest_extent(g) >= ((int64_t)1 << 31)) {                                                  \
      set_error(what ": a tensor of %lld elements (batch %d) exceeds the 2^31 - 1 element limit",       \
                (long long)geom_largest_extent(g), (int)g.N);                                            \
      return SRGAN_ERANGE;                                                                               \
    }                                                                                                    \
  } while (0)

const char* srgan_last_error(void) { return last_error(); }

int srgan_conv2d_fwd(const srgan_conv_desc* desc, const float* x, const float* w, const float* bias, float* y,
                     int force_kernel, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_fwd");
  SRGAN_REQUIRE(x && w && y, SRGAN_EINVAL, "srgan_conv2d_fwd pointers");
  const int dtype = desc->compute_dtype;
  SRGAN_REQUIRE(dtype_ok(dtype), SRGAN_EINVAL, "srgan_conv2d_fwd compute_dtype");
  if (dtype && force_kernel == 0 && use_conv3x3_mixed(g, g.K))     // mixed precision: the LDS-halo 3x3 kernel, or
    return conv3x3_run(x, g.x_bs, w, 0, g.C * 9, 9, 3, 1, bias, y, g.y_bs, g.N, g.C, g.K, g.H, g.W, 0, (hipStream_t)stream,
                       nullptr, nullptr, dtype);
  if (dtype) force_kernel = 2;                                      // the generic MFMA kernel for every other geometry
  if (use_pointwise(g, g.K, force_kernel) && ((uintptr_t)w & 15) == 0 &&
      pointwise_ksplit_wanted(g.N, g.C, g.K, g.H * g.W, false))
    return pointwise_ksplit_run(x, g.x_bs, w, bias, y, g.y_bs, g.N, g.C, g.K, g.H * g.W, 0, (hipStream_t)stream, nullptr);
  if (use_pointwise(g, g.K, force_kernel))
    return pointwise_run(x, g.x_bs, w, g.C, 1, bias, y, g.y_bs, g.N, g.C, g.K, g.H * g.W, 0, (hipStream_t)stream);
  if (use_conv3x3(g, g.K, force_kernel))
    return conv3x3_run(x, g.x_bs, w, 0, g.C * 9, 9, 3, 1, bias, y, g.y_bs, g.N, g.C, g.K, g.H, g.W, 0,
                       (hipStream_t)stream);
  if (force_kernel == 0 && dtype == 0 && bias == nullptr && stem7x7_enabled() &&
      stem7x7_geometry(g.C, g.K, g.R, g.S, g.sh, g.sw, g.ph, g.pw))
    return stem7x7_fwd_run(x, g.x_bs, w, y, g.y_bs, g.N, g.H, g.W, g.K, g.OH, g.OW, (hipStream_t)stream);
  std::vector<GatherGemm> plans{plan_conv_fwd(g, x, w, bias, y)};
  plans[0].precision = dtype;
  // A strided-batch output view (a channel slice of a wider buffer) is zeroed with a 2-D memset when the launch
  // combines K-slices with atomics.
  const bool dense_out = g.y_bs == (int64_t)g.K * g.OH * g.OW;
  if (!dense_out) {
    GGConfig c;
    const bool atomic = gg_prepare(plans[0], force_kernel, &c, (hipStream_t)stream);
    if (atomic) {
      if (const int status = zero_rows(y, g.y_bs, (int64_t)g.K * g.OH * g.OW, g.N, (hipStream_t)stream)) return status;
      plans[0].mode = GG_ATOMIC;
    } else {
      plans[0].mode = GG_STORE;
    }
    return gg_launch(plans[0], c, (hipStream_t)stream);
  }
  return gg_run_group(plans, y, (int64_t)g.N * g.y_bs, 0, force_kernel, (hipStream_t)stream);
}

int srgan_conv2d_bwd_data(const srgan_conv_desc* desc, const float* gy, const float* w, const float* bias,
                          float* gx, int accumulate, int force_kernel, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_bwd_data");
  SRGAN_REQUIRE(gy && w && gx, SRGAN_EINVAL, "srgan_conv2d_bwd_data pointers");
  SRGAN_REQUIRE(g.x_bs == (int64_t)g.C * g.H * g.W, SRGAN_EUNSUPPORTED, "srgan_conv2d_bwd_data dense gx");
  const int dtype = desc->compute_dtype;
  SRGAN_REQUIRE(dtype_ok(dtype), SRGAN_EINVAL, "srgan_conv2d_bwd_data compute_dtype");
  if (dtype && force_kernel == 0 && use_conv3x3_mixed(g, g.C))
    return conv3x3_run(gy, g.y_bs, w, 8, 9, g.C * 9, -3, -1, bias, gx, g.x_bs, g.N, g.K, g.C, g.H, g.W, accumulate,
                       (hipStream_t)stream, nullptr, nullptr, dtype);
  if (force_kernel == 0 && dtype == 0 && bias == nullptr && !accumulate && stem7x7_enabled() &&
      stem7x7_geometry(g.C, g.K, g.R, g.S, g.sh, g.sw, g.ph, g.pw) && (g.W & 1) == 0 && (((uintptr_t)gx) & 7) == 0)
    return stem7x7_bwd_data_run(gy, g.y_bs, w, gx, g.x_bs, g.N, g.H, g.W, g.K, g.OH, g.OW, (hipStream_t)stream);
  // k4 / s2 / p1 (the DCGAN generators' transposed convolutions, reference crowd/models.py:132-136, age/models.py:37-41,
  // as forward passes; the discriminators' strided convolutions as data gradients): output pixel (2q + a, 2r + b) only
  // meets the 2x2 taps kh = 3 + a - 2i, kw = 3 + b - 2j of the input pixels (q - 1 + i, r - 1 + j), i in {a, a + 1},
  // j in {b, b + 1} -- four 2x2 sub-windows of the LDS-halo 3x3 kernel with strided stores, instead of four gathered
  // GEMMs on the generic kernel (50 TF/s).
  static const bool no_k4s2 = getenv("SRGAN_NO_K4S2") != nullptr;
  if (force_kernel == 0 && !no_k4s2 && conv3x3_enabled() && g.R == 4 && g.S == 4 && g.sh == 2 && g.sw == 2 && g.ph == 1 &&
      g.pw == 1 && g.H == 2 * g.OH && g.W == 2 * g.OW && g.OW >= 8 && (g.C >= 8 || dtype) && !accumulate) {   // (mixed: also the 3-channel image gradient)
    // small problems split the input channels over the grid and add with atomics: the output is zeroed once for all classes
    // (only when the split adds with atomics: with a workspace the slices meet in a fixed order and the sums are stored)
    const bool split = conv3x3_splits(g.N, g.K, g.C, g.OH, g.OW, dtype) > 1 && !srgan_split_is_ordered(stream);
    if (split) if (const int status = zero_floats(gx, (int64_t)g.N * g.x_bs, (hipStream_t)stream)) return status;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        Conv3Placement placement;
        placement.taps = (0x01B << (3 * a)) << b;
        placement.out_plane = g.H * g.W; placement.out_sy = 2 * g.W; placement.out_sx = 2; placement.out_off = a * g.W + b;
        const int status = conv3x3_run(gy, g.y_bs, w, (3 + a) * 4 + (3 + b), 16, g.C * 16, -8, -2, bias, gx, g.x_bs, g.N, g.K,
                                       g.C, g.OH, g.OW, split ? 2 : 0, (hipStream_t)stream, nullptr, nullptr, dtype, &placement);
        if (status != SRGAN_OK) return status;
      }
    return SRGAN_OK;
  }
  if (dtype) force_kernel = 2;
  if (use_pointwise(g, g.C, force_kernel))   // the data gradient of a 1x1 convolution is the 1x1 convolution with W^T
    return pointwise_run(gy, g.y_bs, w, 1, g.C, bias, gx, g.x_bs, g.N, g.K, g.C, g.H * g.W, accumulate,
                         (hipStream_t)stream);
  if (use_conv3x3(g, g.C, force_kernel))     // the data gradient is the same convolution with flipped taps
    return conv3x3_run(gy, g.y_bs, w, 8, 9, g.C * 9, -3, -1, bias, gx, g.x_bs, g.N, g.K, g.C, g.H, g.W, accumulate,
                       (hipStream_t)stream);
  std::vector<GatherGemm> plans = plan_conv_bwd_data(g, gy, w, bias, gx);
  for (GatherGemm& plan : plans) plan.precision = dtype;
  return gg_run_group(plans, gx, (int64_t)g.N * g.x_bs, accumulate, force_kernel, (hipStream_t)stream);
}

int srgan_conv2d_bwd_weight(const srgan_conv_desc* desc, const float* x, const float* gy, float* gw,
                            int accumulate, int force_kernel, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_bwd_weight");
  SRGAN_REQUIRE(x && gy && gw, SRGAN_EINVAL, "srgan_conv2d_bwd_weight pointers");
  const int dtype = desc->compute_dtype;
  SRGAN_REQUIRE(dtype_ok(dtype), SRGAN_EINVAL, "srgan_conv2d_bwd_weight compute_dtype");
  // (mixed precision also takes 8-wide planes on the 16-wide tile: half of the columns are dead, but the alternative is
  // the gather-bound generic kernel)
  if (dtype && force_kernel == 0 && conv3x3_wgrad_enabled() && wgrad3x3_geometry(g, 8, true) &&
      (((uintptr_t)x | (uintptr_t)gy) & 15) == 0)
    return conv3x3_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.C, g.K, g.H, g.W, accumulate, (hipStream_t)stream, nullptr,
                             dtype);
  if (dtype) force_kernel = 2;
  if (use_pointwise_wgrad(g, x, gy, force_kernel))
    return pointwise_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.C, g.K, g.H * g.W, accumulate, (hipStream_t)stream);
  if (force_kernel == 0 && dtype == 0 && stem7x7_enabled() && stem7x7_geometry(g.C, g.K, g.R, g.S, g.sh, g.sw, g.ph, g.pw))
    return stem7x7_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.H, g.W, g.K, g.OH, g.OW, accumulate, (hipStream_t)stream);
  if (use_wgrad3x3(g, x, gy, force_kernel))
    return conv3x3_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.C, g.K, g.H, g.W, accumulate, (hipStream_t)stream);
  std::vector<GatherGemm> plans{plan_conv_bwd_weight(g, x, gy, gw)};
  plans[0].precision = dtype;
  return gg_run_group(plans, gw, (int64_t)g.K * g.C * g.R * g.S, accumulate, force_kernel, (hipStream_t)stream);
}

// ---- convolutions whose input is relu(batch_norm_eval(x)) evaluated on the fly (DenseNet norm -> relu -> conv)
struct srgan_bn_relu { const float* mean; const float* inv_std; const float* gamma; const float* beta; };

static bool bn_ok(const srgan_bn_relu* bn) { return bn && bn->mean && bn->inv_std && bn->gamma && bn->beta; }

int srgan_conv2d_bnrelu_supported(const srgan_conv_desc* desc, int pass) {
  ConvGeom g;
  if (desc && desc->compute_dtype != 0) return 0;       // the fused forms are fp32 kernels
  if (!to_geom(desc, g) || geom_largest_extent(g) >= ((int64_t)1 << 31)) return 0;
  if (pass == 0) {
    if (pointwise(g)) return use_pointwise(g, g.K, 0) ? 1 : 0;
    return (use_conv3x3(g, g.K, 0) && g.C <= 512) ? 1 : 0;
  }
  if (pass == 1) {
    // the epilogue's per-workgroup parameter sums must fit the caller's workspace (<= 1/32 of the gradient tensor's bytes)
    if ((size_t)g.N * g.H * g.W * g.C / 8 > WORKSPACE_BYTES) return 0;
    if (pointwise(g)) return use_pointwise(g, g.C, 0) ? 1 : 0;
    return (use_conv3x3(g, g.C, 0) && conv3x3_epilogue_supported(g.N, g.K, g.C, g.H, g.W)) ? 1 : 0;
  }
  if (pass == 2) {
    if (pointwise(g)) return pointwise_wgrad_geometry(g) ? 1 : 0;
    return (conv3x3_wgrad_enabled() && wgrad3x3_geometry(g)) ? 1 : 0;
  }
  return 0;
}

// y_state: 0 = y holds anything (stored over), 2 = y is already zero where the convolution writes (a kernel that splits
// K over the grid then skips its own zero-fill launch; see srgan_conv2d_fwd_bnrelu_splits).
static int fwd_bnrelu(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* w, const float* bias,
                      float* y, int y_state, int* plan_only_split, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_fwd_bnrelu");
  SRGAN_REQUIRE(plan_only_split || (x && w && y && bn_ok(bn)), SRGAN_EINVAL, "srgan_conv2d_fwd_bnrelu pointers");
  SRGAN_REQUIRE(srgan_conv2d_bnrelu_supported(desc, 0), SRGAN_EUNSUPPORTED, "srgan_conv2d_fwd_bnrelu geometry support");
  const float* const coefficients[4] = {bn ? bn->mean : nullptr, bn ? bn->inv_std : nullptr, bn ? bn->gamma : nullptr,
                                        bn ? bn->beta : nullptr};
  if (pointwise(g) && (plan_only_split || ((uintptr_t)w & 15) == 0) &&
      pointwise_ksplit_wanted(g.N, g.C, g.K, g.H * g.W, true)) {
    if (plan_only_split) { *plan_only_split = 1; return SRGAN_OK; }
    return pointwise_ksplit_run(x, g.x_bs, w, bias, y, g.y_bs, g.N, g.C, g.K, g.H * g.W, 0, (hipStream_t)stream,
                                coefficients);
  }
  if (pointwise(g))
    return pointwise_run(x, g.x_bs, w, g.C, 1, bias, y, g.y_bs, g.N, g.C, g.K, g.H * g.W, y_state, (hipStream_t)stream,
                         coefficients, nullptr, plan_only_split);
  if (plan_only_split) { *plan_only_split = conv3x3_splits(g.N, g.C, g.K, g.H, g.W); return SRGAN_OK; }
  return conv3x3_run(x, g.x_bs, w, 0, g.C * 9, 9, 3, 1, bias, y, g.y_bs, g.N, g.C, g.K, g.H, g.W, y_state,
                     (hipStream_t)stream, coefficients);
}

int srgan_conv2d_fwd_bnrelu(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* w,
                            const float* bias, float* y, void* stream) {
  return fwd_bnrelu(desc, x, bn, w, bias, y, 0, nullptr, stream);
}

int srgan_conv2d_fwd_bnrelu_into_zeros(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* w,
                                       const float* bias, float* y, void* stream) {
  return fwd_bnrelu(desc, x, bn, w, bias, y, 2, nullptr, stream);
}

int srgan_conv2d_fwd_bnrelu_splits(const srgan_conv_desc* desc) {
  int split = 0;
  return fwd_bnrelu(desc, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &split, nullptr) == SRGAN_OK ? split : -1;
}

static int bwd_data_bnrelu(const srgan_conv_desc* desc, const float* gy, const float* w, const srgan_bn_relu* bn, const float* x,
                          float* gx, float* g_gamma, float* g_beta, float* partials, int accumulate, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_bwd_data_bnrelu");
  SRGAN_REQUIRE(gy && w && x && gx && bn_ok(bn), SRGAN_EINVAL, "srgan_conv2d_bwd_data_bnrelu pointers");
  SRGAN_REQUIRE((g_gamma == nullptr) == (g_beta == nullptr), SRGAN_EINVAL,
                "srgan_conv2d_bwd_data_bnrelu parameter gradients (both or neither)");
  SRGAN_REQUIRE(srgan_conv2d_bnrelu_supported(desc, 1), SRGAN_EUNSUPPORTED,
                "srgan_conv2d_bwd_data_bnrelu geometry support");
  BnBackwardEpilogue epilogue;
  epilogue.x = x; epilogue.x_bs = g.x_bs;
  epilogue.bn[0] = bn->mean; epilogue.bn[1] = bn->inv_std; epilogue.bn[2] = bn->gamma; epilogue.bn[3] = bn->beta;
  epilogue.g_gamma = g_gamma; epilogue.g_beta = g_beta; epilogue.partial_out = partials;
  if (pointwise(g))
    return pointwise_run(gy, g.y_bs, w, 1, g.C, nullptr, gx, g.x_bs, g.N, g.K, g.C, g.H * g.W, accumulate,
                         (hipStream_t)stream, nullptr, &epilogue);
  SRGAN_REQUIRE(!accumulate, SRGAN_EUNSUPPORTED, "srgan_conv2d_bwd_data_bnrelu 3x3: gx is stored, not accumulated");
  return conv3x3_run(gy, g.y_bs, w, 8, 9, g.C * 9, -3, -1, nullptr, gx, g.x_bs, g.N, g.K, g.C, g.H, g.W, 0,
                     (hipStream_t)stream, nullptr, &epilogue);
}

int srgan_conv2d_bwd_data_bnrelu(const srgan_conv_desc* desc, const float* gy, const float* w, const srgan_bn_relu* bn,
                                 const float* x, float* gx, float* g_gamma, float* g_beta, int accumulate, void* stream) {
  return bwd_data_bnrelu(desc, gy, w, bn, x, gx, g_gamma, g_beta, nullptr, accumulate, stream);
}

int64_t srgan_conv2d_bwd_data_bnrelu_tiles(const srgan_conv_desc* desc) {
  ConvGeom g;
  if (!to_geom(desc, g) || !srgan_conv2d_bnrelu_supported(desc, 1)) return -1;
  return pointwise(g) ? pointwise_epilogue_tiles(g.N, g.H * g.W) : conv3x3_epilogue_tiles(g.N, g.K, g.C, g.H, g.W);
}

int srgan_conv2d_bwd_data_bnrelu_partials(const srgan_conv_desc* desc, const float* gy, const float* w, const srgan_bn_relu* bn,
                                          const float* x, float* gx, float* partials, int accumulate, void* stream) {
  SRGAN_REQUIRE(partials != nullptr, SRGAN_EINVAL, "srgan_conv2d_bwd_data_bnrelu_partials: the partial-sum region");
  return bwd_data_bnrelu(desc, gy, w, bn, x, gx, nullptr, nullptr, partials, accumulate, stream);
}

struct srgan_bn_reduce_job { int64_t partial_offset; int32_t tiles, channels; const float* inv_std; float* g_gamma; float* g_beta; };

int srgan_bn_partial_reduce_batched(const srgan_bn_reduce_job* jobs_device, int32_t count, int32_t max_channels, int32_t max_tiles,
                                    const float* scratch, void* stream) {
  SRGAN_REQUIRE(jobs_device && scratch && count > 0 && max_channels > 0 && max_tiles > 0, SRGAN_EINVAL,
                "srgan_bn_partial_reduce_batched arguments");
  SRGAN_REQUIRE(count <= 65535, SRGAN_ERANGE, "srgan_bn_partial_reduce_batched job count");
  return bn_partial_reduce_batched_run(jobs_device, count, max_channels, max_tiles, scratch, (hipStream_t)stream);
}

int srgan_conv2d_bwd_weight_bnrelu(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* gy,
                                   float* gw, int accumulate, void* stream) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_conv2d_bwd_weight_bnrelu");
  SRGAN_REQUIRE(x && gy && gw && bn_ok(bn), SRGAN_EINVAL, "srgan_conv2d_bwd_weight_bnrelu pointers");
  SRGAN_REQUIRE(srgan_conv2d_bnrelu_supported(desc, 2), SRGAN_EUNSUPPORTED,
                "srgan_conv2d_bwd_weight_bnrelu geometry support");
  const float* const coefficients[4] = {bn->mean, bn->inv_std, bn->gamma, bn->beta};
  if (pointwise(g))
    return pointwise_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.C, g.K, g.H * g.W, accumulate, (hipStream_t)stream,
                               coefficients);
  return conv3x3_wgrad_run(x, g.x_bs, gy, g.y_bs, gw, g.N, g.C, g.K, g.H, g.W, accumulate, (hipStream_t)stream,
                           coefficients);
}

// ---- grouped weight gradients: all the norm -> relu -> conv weight gradients of a dense block's backward in two launches
int srgan_wgrad_group_plan(const srgan_conv_desc* desc, const srgan_bn_relu* bn, int64_t x_offset, int64_t gy_offset, float* gw,
                           int64_t gw_offset, int32_t group_size, int64_t group_weights, int64_t partial_offset, void* job, int32_t* grid_x,
                           int32_t* grid_y, int32_t* ragged, int64_t* partial_floats) {
  ConvGeom g;
  SRGAN_GEOM(desc, g, "srgan_wgrad_group_plan");
  SRGAN_REQUIRE(job && grid_x && grid_y && ragged && partial_floats && (bn == nullptr || bn_ok(bn)) && x_offset >= 0 &&
                gy_offset >= 0 && gw_offset >= 0 && partial_offset >= 0 && group_size >= 1 && group_weights >= 0, SRGAN_EINVAL,
                "srgan_wgrad_group_plan arguments");
  SRGAN_REQUIRE(srgan_conv2d_bnrelu_supported(desc, 2), SRGAN_EUNSUPPORTED, "srgan_wgrad_group_plan geometry support");
  const float* const coefficients[4] = {bn ? bn->mean : nullptr, bn ? bn->inv_std : nullptr, bn ? bn->gamma : nullptr,
                                        bn ? bn->beta : nullptr};
  const float* const* fused = bn ? coefficients : nullptr;
  if (pointwise(g))
    return pointwise_wgrad_group_plan(x_offset, g.x_bs, gy_offset, g.y_bs, gw, gw_offset, g.N, g.C, g.K, g.H * g.W, fused,
                                      group_size, group_weights, partial_offset, job, grid_x, grid_y, ragged, partial_floats);
  return conv3x3_wgrad_group_plan(x_offset, g.x_bs, gy_offset, g.y_bs, gw, gw_offset, g.N, g.C, g.K, g.H, g.W, fused, group_size,
                                  partial_offset, job, grid_x, grid_y, ragged, partial_floats);
}

int srgan_wgrad_group_run(const void* jobs, int32_t count, int32_t kernel_size, int32_t grid_x, int32_t grid_y, int32_t ragged,
                          int32_t fused_bn, const float* x_base, const float* gy_base, float* gw_base, int64_t sum_co_ci_taps,
                          int64_t pixels, int64_t operand_elements, int64_t partial_floats, void* stream) {
  SRGAN_REQUIRE(jobs && x_base && gy_base && count >= 1 && grid_x >= 1 && grid_y >= 1 && (kernel_size == 1 || kernel_size == 3) &&
                partial_floats >= 0, SRGAN_EINVAL, "srgan_wgrad_group_run arguments");
  if (kernel_size == 1)
    return pointwise_wgrad_group_run(jobs, count, grid_x, grid_y, ragged, fused_bn, x_base, gy_base, gw_base, sum_co_ci_taps,
                                     pixels, operand_elements, partial_floats, (hipStream_t)stream);
  return conv3x3_wgrad_group_run(jobs, count, grid_x, grid_y, ragged, x_base, gy_base, gw_base, sum_co_ci_taps, pixels,
                                 operand_elements, partial_floats, (hipStream_t)stream);
}

int srgan_profile_begin(void) {
  std::lock_guard<std::mutex> lock(g_profile.mutex);
  g_profile.slots = 0;
  g_profile.records.clear();
  g_profile.flops = g_profile.mfma_flops = g_profile.bytes = 0.0;
  g_profile.launches = 0;
  g_profile.enabled.store(true, std::memory_order_release);
  return SRGAN_OK;
}

int srgan_profile_end(double* kernel_ms, double* flops, double* mfma_flops, int64_t* launches) {
  g_profile.enabled.store(false, std::memory_order_release);
  std::lock_guard<std::mutex> lock(g_profile.mutex);
  double total = 0.0;
  for (size_t i = 0; i < g_profile.slots; ++i) {
    if (g_profile.records[i].M < 0) continue;
    SRGAN_HIP(hipEventSynchronize(g_profile.events[2 * i + 1]));
    float ms = 0.f;
    SRGAN_HIP(hipEventElapsedTime(&ms, g_profile.events[2 * i], g_profile.events[2 * i + 1]));
    total += ms;
  }
  if (kernel_ms) *kernel_ms = total;
  if (flops) *flops = g_profile.flops;
  if (mfma_flops) *mfma_flops = g_profile.mfma_flops;
  if (launches) *launches = g_profile.launches;
  return SRGAN_OK;
}

int srgan_profile_bytes(double* algorithmic_bytes_total) {
  std::lock_guard<std::mutex> lock(g_profile.mutex);
  if (algorithmic_bytes_total) *algorithmic_bytes_total = g_profile.bytes;
  return SRGAN_OK;
}

// The part of the profiled region that ran with bf16 / fp16 MFMA operands: its logical FLOPs and its summed kernel time.
int srgan_profile_mixed(double* flops, double* kernel_ms) {
  std::lock_guard<std::mutex> lock(g_profile.mutex);
  double f = 0.0, total = 0.0;
  for (size_t i = 0; i < g_profile.slots; ++i) {
    const ProfileRecord& r = g_profile.records[i];
    if (r.M < 0 || r.precision == 0) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_profile.events[2 * i], g_profile.events[2 * i + 1]) != hipSuccess) continue;
    total += ms;
    f += 2.0 * (double)r.M * (double)r.N * (double)r.K;
  }
  if (flops) *flops = f;
  if (kernel_ms) *kernel_ms = total;
  return SRGAN_OK;
}

// Per-shape breakdown of the last profiled region as text lines "M N K kind bm bn split akf bkf count ms bytes"
// (kind: 0 direct, 1 gg_mfma, 2 conv3x3_lds, 3 pointwise, 13 pointwise_ring, 4 conv3x3_wgrad, 5 gg_rows, 6 pointwise_wgrad,
// 8 pointwise_ksplit, 9 gg_dot, 10 stem7x7_fwd, 11 stem7x7_wgrad; bytes = algorithmic HBM bytes of all `count` launches; call after srgan_profile_end).
// Returns the number of bytes needed.
int64_t srgan_profile_report(char* buffer, int64_t capacity) {
  struct Key { int32_t v[9]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  struct Acc { int64_t count = 0; double ms = 0.0, bytes = 0.0; };
  std::map<Key, Acc> table;
  std::lock_guard<std::mutex> lock(g_profile.mutex);
  for (size_t i = 0; i < g_profile.slots; ++i) {
    const ProfileRecord& r = g_profile.records[i];
    if (r.M < 0) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_profile.events[2 * i], g_profile.events[2 * i + 1]) != hipSuccess) continue;
    Key k{{r.M, r.N, r.K, r.kind, r.bm, r.bn, r.split, r.akf, r.bkf}};
    Acc& a = table[k];
    a.count += 1;
    a.ms += ms;
    a.bytes += r.bytes;
  }
  std::string out;
  char line[200];
  for (const auto& kv : table) {
    const int32_t* v = kv.first.v;
    snprintf(line, sizeof(line), "%d %d %d %d %d %d %d %d %d %lld %.4f %.0f\n", v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7],
             v[8], (long long)kv.second.count, kv.second.ms, kv.second.bytes);
    out += line;
  }
  if (buffer && capacity > 0) {
    const size_t n = out.size() < (size_t)capacity - 1 ? out.size() : (size_t)capacity - 1;
    memcpy(buffer, out.data(), n);
    buffer[n] = 0;
  }
  return (int64_t)out.size() + 1;
}

// ---- caller-owned workspace, capabilities -----------------------------------------------------------------------
int64_t srgan_workspace_bytes(void) { return (int64_t)WORKSPACE_BYTES; }

int srgan_set_workspace(void* workspace, int64_t bytes, void* stream) {
  SRGAN_REQUIRE(workspace == nullptr || bytes >= (int64_t)WORKSPACE_BYTES, SRGAN_EINVAL,
                "srgan_set_workspace: at least srgan_workspace_bytes() bytes");
  SRGAN_REQUIRE(((uintptr_t)workspace & 15) == 0, SRGAN_EINVAL, "srgan_set_workspace: 16-byte alignment");
  return workspace_register((float*)workspace, (size_t)(bytes > 0 ? bytes : 0), (hipStream_t)stream);
}

// 1 when a K split on this stream is finished in a fixed order through the registered workspace (split_finish.h): such a
// launch stores whole sums -- no zero-filled output needed, the same bits every run; 0: fp32 atomics into a zeroed output.
int srgan_split_is_ordered(void* stream) {
  if (split_atomics_forced()) return 0;
  const int set = workspace_index((hipStream_t)stream);
  return set >= 0 && set < SPLIT_TICKET_SETS ? 1 : 0;
}

struct srgan_capabilities_t {
  int32_t abi_version;         // = srgan_version()
  int32_t struct_bytes;        // sizeof(this struct) as the library knows it
  char arch[16];               // "gfx950"
  uint32_t dtypes;             // bit 0: fp32 (the parity path)
  uint32_t features;           // SRGAN_FEATURE_* bits
  int64_t workspace_bytes;     // = srgan_workspace_bytes()
  int64_t max_tensor_elements; // 2^31 - 1
};

int srgan_capabilities(srgan_capabilities_t* out, int32_t out_bytes) {
  SRGAN_REQUIRE(out != nullptr && out_bytes >= (int32_t)sizeof(srgan_capabilities_t), SRGAN_EINVAL,
                "srgan_capabilities: output struct");
  memset(out, 0, sizeof(*out));
  out->abi_version = 110;
  out->struct_bytes = (int32_t)sizeof(*out);
  snprintf(out->arch, sizeof(out->arch), "gfx950");
  out->dtypes = 0x1u /* fp32 */ | 0x2u /* bf16 MFMA operands */ | 0x4u /* fp16 MFMA operands */;
  out->features = 0x1u /* fused batch-norm + relu prologues / epilogues */ | 0x2u /* split-K through a workspace */ |
                  0x4u /* live event profile of the contraction launches */ |
                  0x8u /* 16-bit blocked data path (srgan_h_*) */;
  out->workspace_bytes = (int64_t)WORKSPACE_BYTES;
  out->max_tensor_elements = ((int64_t)1 << 31) - 1;
  return SRGAN_OK;
}

int srgan_gemm(int32_t M, int32_t N, int32_t K, const float* A, int64_t sai, int64_t sak, const float* B,
               int64_t sbk, int64_t sbj, float* C, int64_t sci, int64_t scj, const float* bias,
               int32_t bias_on_columns, int accumulate, int force_kernel, int compute_dtype, void* stream);

int srgan_gemm_f32(int32_t M, int32_t N, int32_t K, const float* A, int64_t sai, int64_t sak, const float* B,
                   int64_t sbk, int64_t sbj, float* C, int64_t sci, int64_t scj, const float* bias,
                   int32_t bias_on_columns, int accumulate, int force_kernel, void* stream) {
  return srgan_gemm(M, N, K, A, sai, sak, B, sbk, sbj, C, sci, scj, bias, bias_on_columns, accumulate, force_kernel, 0, stream);
}

int srgan_gemm(int32_t M, int32_t N, int32_t K, const float* A, int64_t sai, int64_t sak, const float* B,
               int64_t sbk, int64_t sbj, float* C, int64_t sci, int64_t scj, const float* bias,
               int32_t bias_on_columns, int accumulate, int force_kernel, int compute_dtype, void* stream) {
  SRGAN_REQUIRE(M > 0 && N > 0 && K >= 0 && A && B && C, SRGAN_EINVAL, "srgan_gemm_f32 arguments");
  SRGAN_REQUIRE(dtype_ok(compute_dtype), SRGAN_EINVAL, "srgan_gemm compute_dtype");
  if (compute_dtype) force_kernel = 2;
  const int64_t lim = (int64_t)1 << 31;
  SRGAN_REQUIRE(M * sai + K * sak < lim && K * sbk + N * sbj < lim && M * sci + N * scj < lim, SRGAN_ERANGE,
                "srgan_gemm_f32 extents");
  // C must be a dense M x N matrix in one of the two orientations so that it can be zeroed for split-K.
  SRGAN_REQUIRE((scj == 1 && sci == N) || (sci == 1 && scj == M), SRGAN_EUNSUPPORTED, "srgan_gemm_f32 dense C");
  GatherGemm p = plan_gemm(M, N, K, A, (int32_t)sai, (int32_t)sak, B, (int32_t)sbk, (int32_t)sbj, C, (int32_t)sci,
                           (int32_t)scj, bias, bias_on_columns);
  if (sci == 1 && scj != 1) {   // make the lane dimension the contiguous one
    p = gg_transposed(p);
    choose_staging(p);
  }
  p.precision = compute_dtype;
  std::vector<GatherGemm> plans{p};
  return gg_run_group(plans, C, (int64_t)M * N, accumulate, force_kernel, (hipStream_t)stream);
}

}  // extern "C"
