// pointwise.hip -- 1x1 / stride 1 convolution: out[n, o, p] = sum_i w(o, i) * in[n, i, p]  (+ bias[o]).
//
// 63 % of the DenseNet-201 FLOPs are these (bottleneck and transition convolutions, reference
// crowd/models.py:340-341,369-370) and their data gradients.  On NCHW data it is C[CO x P] = W[CO x CI] * X[CI x P]
// with the pixel dimension P contiguous and at most a few hundred rows of W: the arithmetic intensity sits right at
// the fp32-MFMA / HBM ridge (43 FLOP/B at CI = 256), so the activation stream has to overlap the matrix pipe
// completely.  Design:
//   * B operand (activations) NEVER touches LDS: the MFMA B fragment of v_mfma_f32_32x32x2_f32 is "32 consecutive
//     pixels of channel k (lanes 0-31) and of channel k+1 (lanes 32-63)", i.e. two 128-byte rows -- exactly one
//     coalesced global_load_dword per lane.  Every wave owns its own pixel columns for ALL output rows of the tile, so
//     nothing is shared between waves.  A lane's address is base + k * HW: no index decode at all.
//   * the whole next K-slice of B (16 k-pairs) is loaded into registers before the current slice's MFMAs issue: one
//     full slice of matrix work (4096 cycles at 4 row blocks) hides the HBM latency.
//   * A operand (weights, tiny and shared by every workgroup) is staged through a double-buffered LDS tile
//     [32][BM + 1]: one barrier per K-slice.
//   * 64-row tiles by default: <= 128 VGPRs, so four workgroups share a CU and their load / MFMA / store phases
//     overlap (+3 % on the training step over 128-row tiles at two workgroups per CU).
//   * output tiles of up to 64 rows leave through LDS as float4 rows (512 contiguous bytes of one channel per
//     half-wave); in a data gradient the batch-norm + ReLU backward of the layer in front is applied on the way (EPI).
// Roofline: fp32 MFMA 157.3 TF/s, or HBM when CI is small; algorithmic bytes 4 * (CI + CO) per pixel.
#include "common.h"
#include "split_finish.h"
#include <stdlib.h>
#include <type_traits>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PointwiseParams {
  const float* in;      // [N, CI, HW], batch stride in_bs
  const float* w;       // element (o, i) at w[o * w_so + i * w_si]
  float* out;           // [N, CO, HW], batch stride out_bs
  const float* bias;
  int32_t N, CI, CO, HW;
  int64_t in_bs, out_bs;
  int32_t w_so, w_si;
  int32_t gpi;          // pixel groups (of 32 * NI) per image, the last one possibly ragged
  int32_t tiles_m;
  int32_t xcd_remap;    // grid.x is a multiple of 8 and tiles_m > 1: XCD-aware workgroup order
  int32_t m_base;       // first output row of this launch (a second launch covers a shorter remainder tile)
  int32_t k_per_split;
  int32_t mode;         // 0 store, 1 accumulate, 2 atomic; K split with the ordered finish (split_finish.h): 3 store, 4 accumulate
  float* split_ws;      // modes 3 / 4: [tiles][splits][accumulators of a workgroup] partials in the caller's workspace
  unsigned int* split_tickets;
  int32_t wide_out;     // out rows / batch stride 16-byte aligned: modes 0 and 1 store float4 rows through LDS
  // PRO: the input is relu(batch_norm_eval(in)) computed on the fly (per input channel; NULL otherwise)
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
  // EPI: the output rows go through the backward of relu(batch_norm_eval(epi_x)) on their way out (bn_* then describe
  // the OUTPUT channels): out (=, +=) acc * [fma(x, a, b) > 0] * a, and the two parameter-gradient row sums of this
  // workgroup's 128 pixels are written to epi_partial[q][column block][CO] (q = 0 beta, 1 gamma before inv_std).
  const float* epi_x; int64_t epi_x_bs;
  float* epi_partial; int32_t epi_cols;
  int32_t epi_ragged;   // epi_x / out rows are not whole aligned float4s: the RAGROW variant
};

__device__ unsigned int g_pointwise_split_tickets[SPLIT_TICKET_SETS * SPLIT_TICKET_TILES];

// PRO = frozen batch-norm + ReLU fused into the B-operand stream (reference crowd/models.py:338-341: norm1, relu1,
// conv1): the per-channel (a, b) of the slice are staged in LDS next to the weight tile and every activation goes
// through max(fma(x, a, b), 0) in registers on its way into the MFMA -- the normalised tensor never exists in HBM.
// NI = 32-pixel column groups per wave (1 or 2): with two, every weight fragment read from LDS feeds two MFMAs and the
// non-matrix instructions of a k-pair are amortised over 2*MI MFMAs.
// EPI = the backward of a frozen batch-norm + ReLU fused into the epilogue of a data gradient (reference
// crowd/models.py:338-341 backwards: conv1 -> relu1 -> norm1): see PointwiseParams::epi_x.
// RAGROW (with EPI): planes whose size is not a multiple of 4 pixels or rows that are only 4-byte aligned (7 x 7 at the
// reference's 224 x 224): the epilogue's float4 row accesses become predicated single floats.
template <int MI, int BK, bool PRO, int NI, bool EPI = false, bool RAGROW = false>
__global__ __launch_bounds__(256, MI * NI >= 4 ? 2 : 4) void pointwise_kernel(const PointwiseParams p) {
  static_assert(!EPI || (!PRO && NI == 1), "the batch-norm backward epilogue pairs with the plain single-group kernel");
  constexpr int BM = MI * 32, KP = BK / 2, LDA = BM + 1;
  constexpr int EA = BM * BK / 256;
  constexpr int LDT = 128 + 4;                                     // EPI: row stride of the staged output tile
  constexpr bool STAGED = MI <= 2 && NI == 1;                      // the output tile can go through LDS (<= 64 rows)
  constexpr int LDS_MAIN = 2 * BK * LDA, LDS_EPI = STAGED ? BM * LDT + BM * 4 : 0;
  __shared__ float lds[LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI];
  __shared__ float2 coef[2][PRO ? BK : 1];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: keeps the address bases scalar
  // XCD-aware order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2),
  // while the row tiles that share one activation column block have consecutive LOGICAL ids -- so logical id =
  // (hardware id % 8) * (grid / 8) + hardware id / 8 keeps them on one XCD and the activations are fetched into one L2.
  int bid = (int)blockIdx.x;
  if (p.xcd_remap) bid = (bid & 7) * ((int)gridDim.x >> 3) + (bid >> 3);
  const int tm = bid % p.tiles_m;
  // Pixel groups never straddle images: an image has gpi = ceil(HW / (32 * NI)) groups, the last one ragged when HW is
  // not a multiple of the group (14 x 14 and 7 x 7 planes of the 224-pixel configuration): its surplus lanes read a
  // clamped address and are never stored.
  const int64_t group = (int64_t)(bid / p.tiles_m) * 4 + wave;           // this wave's NI adjacent 32-pixel groups
  const int64_t total = (int64_t)p.N * p.gpi;                           // groups in the whole batch
  const bool live = group < total;
  const int n = live ? (int)(group / p.gpi) : 0;
  const int pix0 = live ? (int)(group - (int64_t)n * p.gpi) * (32 * NI) : 0;
  const int pix = pix0 + l31;
  const int m0 = p.m_base + tm * BM;
  const int kbeg = (int)blockIdx.y * p.k_per_split;
  const int kend = min(p.CI, kbeg + p.k_per_split);

  // Activation addresses: a wave-uniform 64-bit base per channel pair (scalar unit) + ONE 32-bit per-lane offset
  // (pixel, and one image plane for the odd channel of the pair).  CI is even (checked by the caller), so a pair is
  // clamped as a whole.
  const float* b_wave = p.in + (int64_t)n * p.in_bs + pix0;
  const uint32_t lane_off = (uint32_t)(min(pix0 + l31, p.HW - 1) - pix0) + (uint32_t)lhi * (uint32_t)p.HW;

  // A staging coordinates: lanes walk the weight's contiguous direction.
  const bool k_contiguous = p.w_si == 1;
  int a_k[EA], a_m[EA];
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    const int flat = e * 256 + tid;
    a_k[e] = k_contiguous ? flat % BK : flat / BM;
    a_m[e] = k_contiguous ? flat / BK : flat % BM;
  }

  float ra[EA];
  float b0[NI][KP], b1[NI][KP];
  float rc[4];                                 // PRO: raw batch-norm parameters of channel k0 + (tid % BK)
  auto fetch_c = [&](int k0) {
    if (PRO) {
      int k = k0 + (tid & (BK - 1));
      k = k < kend ? k : kend - 1;
      rc[0] = p.bn_mean[k]; rc[1] = p.bn_inv[k]; rc[2] = p.bn_gamma[k]; rc[3] = p.bn_beta[k];
    }
  };
  auto fetch_a = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const int k = min(k0 + a_k[e], kend - 1), m = min(m0 + a_m[e], p.CO - 1);   // clamped, masked at stage time
      ra[e] = p.w[m * p.w_so + k * p.w_si];
    }
    fetch_c(k0);
  };
  auto stage_a = [&](int k0, float* As, int buffer) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const bool ok = (k0 + a_k[e]) < kend && (m0 + a_m[e]) < p.CO;
      As[a_k[e] * LDA + a_m[e]] = ok ? ra[e] : 0.f;
    }
    if (PRO && tid < BK) {
      float a, b;
      bn_coefficients(rc[0], rc[1], rc[2], rc[3], a, b);
      const bool ok = k0 + tid < kend;                       // beyond the split: (0, 0) -> the activation is 0
      coef[buffer][tid] = make_float2(ok ? a : 0.f, ok ? b : 0.f);
    }
  };
  auto fetch_b = [&](int k0, float (&dst)[NI][KP]) {
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int k = min(k0 + 2 * q, kend - 2);          // clamped pair: the matching A rows are zero
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) dst[ni][q] = (b_wave + (int64_t)k * p.HW)[lane_off + 32 * ni];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // One K-slice: prefetch the following slice (B into `bnxt`, A into registers), run this slice's MFMAs from `bcur`
  // and LDS buffer `buffer`, then publish the prefetched A tile into the other LDS buffer.  The two B register sets
  // are used ping-pong by the caller (explicitly unrolled by two) so that a slice's loads are in flight during the
  // whole previous slice of matrix work; a register copy at the end of the loop gets folded away by the compiler
  // and with it the prefetch distance.
  auto slice = [&](int k0, const float (&bcur)[NI][KP], float (&bnxt)[NI][KP], int buffer) {
    const bool more = k0 + BK < kend;
    const float* As = lds + buffer * (BK * LDA) + lhi * LDA + l31;
    // The next slice's WEIGHT (and batch-norm) loads all go out first: stage_a() below waits for them (vmcnt is
    // in-order), so a load issued late in the slice would expose a memory round trip in front of every barrier.
    // (A laboratory copy of this loop measured 97 vs 91 TF/s for "all first" vs "one per k-pair", and the
    // sched_group_barrier pinning of the interleave measured slower than the compiler's own schedule.)
    fetch_a(k0 + BK);
    float a[2][MI];
    float2 cf[2];
    const float2* cs = &coef[PRO ? buffer : 0][PRO ? lhi : 0];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[0][mi] = As[mi * 32];
    if (PRO) cf[0] = cs[0];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      if (q + 1 < KP) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[(q + 1) & 1][mi] = As[(2 * (q + 1)) * LDA + mi * 32];
        if (PRO) cf[(q + 1) & 1] = cs[2 * (q + 1)];
      }
      float bq[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bq[ni] = PRO ? fmaxf(fmaf(bcur[ni][q], cf[q & 1].x, cf[q & 1].y), 0.f) : bcur[ni][q];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mi], bq[ni], acc[mi][ni], 0, 0, 0);
      {   // unconditional (clamped) so that the slice stays one basic block; the last slice's loads are unused
        const int k = min(k0 + BK + 2 * q, kend - 2);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bnxt[ni][q] = (b_wave + (int64_t)k * p.HW)[lane_off + 32 * ni];
      }
    }
    if (more) {
      stage_a(k0 + BK, lds + (buffer ^ 1) * (BK * LDA), buffer ^ 1);   // the other buffer: nobody reads it during this slice
      __syncthreads();
    }
  };

  if (kbeg < kend) {
    fetch_a(kbeg);
    fetch_b(kbeg, b0);
    stage_a(kbeg, lds, 0);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
      slice(k0, b0, b1, 0);
      if (k0 + BK < kend) slice(k0 + BK, b1, b0, 1);
    }
  }

  if (EPI) {
    // The accumulator tile goes through LDS so that the epilogue works on float4s along the pixels: a half-wave owns
    // one output row (512 contiguous bytes per global access instead of 128, a quarter of the instructions), every
    // lane pre-sums its four pixels before the cross-lane reduction of the parameter gradients, and a row's sums are
    // complete inside one half-wave -- no second combine.  Loads first, stores last (the compiler cannot prove that
    // the stores do not alias x or the accumulated gradient, so a load issued after a store would wait for it).
    static_assert(!EPI || MI <= 2, "the staged output tile of the epilogue fits the LDS up to 64 rows");
    __syncthreads();                               // every wave is done with the weight tiles: the LDS is reused
    float* tile = lds;                             // [BM][LDT]
    float* table = lds + BM * LDT;                 // [BM][4]: a, b, mean of output row m0 + i
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDT + wave * 32 + l31] = acc[mi][0][r];
    if (tid < BM) {
      const int o = min(m0 + tid, p.CO - 1);
      const float mu = p.bn_mean[o];
      float a, b;
      bn_coefficients(mu, p.bn_inv[o], p.bn_gamma[o], p.bn_beta[o], a, b);   // the forward's own (a, b): same mask
      table[tid * 4 + 0] = a; table[tid * 4 + 1] = b; table[tid * 4 + 2] = mu;
    }
    __syncthreads();
    constexpr int RPT = BM / 8;                    // rows per thread: row = (tid >> 5) + 8 * e
    const int half = tid >> 5, q4 = (tid & 31) * 4;
    const int64_t my_group = (int64_t)(bid / p.tiles_m) * 4 + (q4 >> 5);               // this lane's 32-pixel group
    const int my_n = my_group < total ? (int)(my_group / p.gpi) : 0;
    const int my_start = my_group < total ? (int)(my_group - (int64_t)my_n * p.gpi) * 32 + (q4 & 31) : 0;
    const bool my_live = my_group < total && my_start < p.HW;        // (HW % 4 == 0: a float4 is inside or outside)
    const int my_pix = my_live ? my_start : 0;
    const int my_count = RAGROW ? (my_live ? min(4, p.HW - my_start) : 0) : 4;     // RAGROW: pixels of the float4 that exist
    auto load4 = [&](const float* src) {
      if constexpr (!RAGROW) return *reinterpret_cast<const float4*>(src);
      float4 v;
      v.x = my_count > 0 ? src[0] : 0.f; v.y = my_count > 1 ? src[1] : 0.f;
      v.z = my_count > 2 ? src[2] : 0.f; v.w = my_count > 3 ? src[3] : 0.f;
      return v;
    };
    const float* x_lane = p.epi_x + (int64_t)my_n * p.epi_x_bs + my_pix;
    float* out_lane = p.out + (int64_t)my_n * p.out_bs + my_pix;
    const bool sums_wanted = p.epi_partial != nullptr;
    auto emit = [&](auto accumulate) {
      float4 xs[RPT], olds[RPT];
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        const int o = min(m0 + half + 8 * e, p.CO - 1);
        xs[e] = load4(x_lane + (int64_t)o * p.HW);
        if constexpr (decltype(accumulate)::value) olds[e] = load4(out_lane + (int64_t)o * p.HW);
      }
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        const int row = half + 8 * e;
        const int o = m0 + row;
        const bool ok = my_live && o < p.CO;
        const float4 t = *reinterpret_cast<const float4*>(table + row * 4);
        float4 v = *reinterpret_cast<const float4*>(tile + row * LDT + q4);
        const float4 x = xs[e];
        v.x = (ok && my_count > 0 && fmaf(x.x, t.x, t.y) > 0.f) ? v.x : 0.f;
        v.y = (ok && my_count > 1 && fmaf(x.y, t.x, t.y) > 0.f) ? v.y : 0.f;
        v.z = (ok && my_count > 2 && fmaf(x.z, t.x, t.y) > 0.f) ? v.z : 0.f;
        v.w = (ok && my_count > 3 && fmaf(x.w, t.x, t.y) > 0.f) ? v.w : 0.f;
        if (ok) {
          typedef float v4f __attribute__((ext_vector_type(4)));
          v4f result;
          result.x = v.x * t.x; result.y = v.y * t.x; result.z = v.z * t.x; result.w = v.w * t.x;
          if constexpr (decltype(accumulate)::value) {
            result.x += olds[e].x; result.y += olds[e].y; result.z += olds[e].z; result.w += olds[e].w;
          }
          if constexpr (RAGROW) {
            float* dst = out_lane + (int64_t)o * p.HW;
            if (my_count > 0) dst[0] = result.x;
            if (my_count > 1) dst[1] = result.y;
            if (my_count > 2) dst[2] = result.z;
            if (my_count > 3) dst[3] = result.w;
          } else {
            v4f* dst = reinterpret_cast<v4f*>(out_lane + (int64_t)o * p.HW);
            if constexpr (decltype(accumulate)::value) *dst = result;
            else __builtin_nontemporal_store(result, dst);
          }
        }
        if (sums_wanted) {
          const float plain = half_wave_sum((v.x + v.y) + (v.z + v.w));
          const float centred = half_wave_sum((v.x * (x.x - t.z) + v.y * (x.y - t.z)) + (v.z * (x.z - t.z) + v.w * (x.w - t.z)));
          if (l31 == 31 && o < p.CO) {
            float* partial = p.epi_partial + (int64_t)(bid / p.tiles_m) * p.CO + o;
            partial[0] = plain;
            partial[(int64_t)p.epi_cols * p.CO] = centred;
          }
        }
      }
    };
    if (p.mode == 0) emit(std::false_type{});
    else emit(std::true_type{});
    return;
  }
  int mode = p.mode;
  if (mode >= 3) {       // K split, ordered finish (split_finish.h): the tile's last workgroup goes on with the sum of all slices
    constexpr int COUNT = MI * NI * 16;
    if (!split_finish_ordered<COUNT, 256>(p.split_ws + (int64_t)bid * gridDim.y * (COUNT * 256), (int)blockIdx.y, (int)gridDim.y,
                                          p.split_tickets + bid, [&](int i) { return acc[i / (NI * 16)][(i / 16) % NI][i % 16]; },
                                          [&](int i, float v) { acc[i / (NI * 16)][(i / 16) % NI][i % 16] = v; }))
      return;
    mode -= 3;
  }
  if (STAGED && p.wide_out && mode != 2) {
    // Store / accumulate through LDS as float4 rows (see the EPI epilogue: 512 contiguous bytes per access, a quarter
    // of the instructions); rows and batch strides are 16-byte aligned (checked by the launcher).
    __syncthreads();
    float* tile = lds;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDT + wave * 32 + l31] = acc[mi][0][r];
    __syncthreads();
    constexpr int RPT = BM / 8;
    const int half = tid >> 5, q4 = (tid & 31) * 4;
    const int64_t my_group = (int64_t)(bid / p.tiles_m) * 4 + (q4 >> 5);
    if (my_group >= total) return;
    const int my_n = (int)(my_group / p.gpi);
    const int my_start = (int)(my_group - (int64_t)my_n * p.gpi) * 32 + (q4 & 31);
    if (my_start >= p.HW) return;                                    // (HW % 4 == 0: a float4 is inside or outside)
    float* out_lane = p.out + (int64_t)my_n * p.out_bs + my_start;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f olds[RPT];
    if (mode == 1) {
#pragma unroll
      for (int e = 0; e < RPT; ++e)
        olds[e] = *reinterpret_cast<const v4f*>(out_lane + (int64_t)min(m0 + half + 8 * e, p.CO - 1) * p.HW);
    }
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const int row = half + 8 * e, o = m0 + row;
      if (o >= p.CO) continue;
      v4f v = *reinterpret_cast<const v4f*>(tile + row * LDT + q4);
      if (p.bias) v += p.bias[o];
      v4f* dst = reinterpret_cast<v4f*>(out_lane + (int64_t)o * p.HW);
      if (mode == 1) *dst = olds[e] + v;
      else __builtin_nontemporal_store(v, dst);                    // consumed by a later kernel
    }
    return;
  }
  if (!live) return;
  const bool add_bias = p.bias != nullptr && (blockIdx.y == 0 || p.mode >= 3);
  float* out_lane = p.out + (int64_t)n * p.out_bs + pix;
  // Three straight-line passes selected once (store / accumulate / atomic): with the mode tested per element the
  // 16 * MI stores of a lane are separated by branches and cannot be issued back to back.
  // A wave whose 32 * NI pixels all exist (every wave unless the plane is not a multiple of the group) stores without
  // per-lane predicates: a lane-varying test around every store would put an exec-mask update between them.
  const bool ragged = pix0 + 32 * NI > p.HW;                       // wave-uniform
  auto emit = [&](auto&& write, auto check_lane) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (o >= p.CO) continue;
        const float bias = add_bias ? p.bias[o] : 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (decltype(check_lane)::value) {
            if (pix + 32 * ni >= p.HW) continue;
          }
          write(out_lane + (int64_t)o * p.HW + 32 * ni, acc[mi][ni][r] + bias);
        }
      }
    }
  };
  // accumulate: the old values of a 32-row block in one batch in front of its stores (loads and stores return through one
  // in-order counter, vmcnt: `*dst += v` per element made every load wait for the store in front of it)
  auto emit_accumulate = [&](auto check_lane) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float previous[16][NI];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bool ok = o < p.CO;
          if constexpr (decltype(check_lane)::value) ok = ok && pix + 32 * ni < p.HW;
          previous[r][ni] = ok ? out_lane[(int64_t)o * p.HW + 32 * ni] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (o >= p.CO) continue;
        const float bias = add_bias ? p.bias[o] : 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (decltype(check_lane)::value) {
            if (pix + 32 * ni >= p.HW) continue;
          }
          out_lane[(int64_t)o * p.HW + 32 * ni] = previous[r][ni] + (acc[mi][ni][r] + bias);
        }
      }
    }
  };
  auto emit_mode = [&](auto check_lane) {
    if (mode == 0) emit([](float* dst, float v) { __builtin_nontemporal_store(v, dst); }, check_lane);   // consumed by a later kernel
    else if (mode == 1) emit_accumulate(check_lane);
    else emit([](float* dst, float v) { unsafeAtomicAdd(dst, v); }, check_lane);
  };
  if (ragged) emit_mode(std::true_type{});
  else emit_mode(std::false_type{});
}

// Second level of the fused batch-norm backward's parameter gradients: column sums of partial[q][column block][CO]
// (lanes along the channels: coalesced rows).  Round 5, ORDERED form (the default): ONE workgroup owns 16 channels, its 16
// column-block lanes walk all the blocks, the sixteen sums meet in a fixed order and the workgroup adds the total to the
// gradient -- one adder per channel and launch, the same bits every run.  With SRGAN_ATOMIC_SPLIT=1: round 4's form, a
// few segments of column blocks per channel, 64 channels x 4 lanes per workgroup, one fp32 atomic per segment.
template <bool ORDERED>
__device__ __forceinline__ void bn_partial_reduce_body(const float* __restrict__ partial, int cols, int CO,
                                                       const float* __restrict__ inv_std, float* __restrict__ g_gamma,
                                                       float* __restrict__ g_beta, int first, int last, int block_x,
                                                       float (*scratch)[256]) {
  constexpr int LANES_C = ORDERED ? 16 : 64, LANES_T = 256 / LANES_C;
  const int lane_c = (int)threadIdx.x % LANES_C, rl = (int)threadIdx.x / LANES_C;
  const int c = block_x * LANES_C + lane_c;
  float plain = 0.f, centred = 0.f;
  if (c < CO)
    for (int cb = first + rl; cb < last; cb += LANES_T) {
      plain += partial[(int64_t)cb * CO + c];
      centred += partial[((int64_t)cols + cb) * CO + c];
    }
  scratch[0][rl * LANES_C + lane_c] = plain;
  scratch[1][rl * LANES_C + lane_c] = centred;
  __syncthreads();
  if (rl != 0 || c >= CO) return;
  plain = centred = 0.f;
#pragma unroll
  for (int r = 0; r < LANES_T; ++r) {                         // fixed order
    plain += scratch[0][r * LANES_C + lane_c];
    centred += scratch[1][r * LANES_C + lane_c];
  }
  if (ORDERED) {
    g_beta[c] += plain;
    g_gamma[c] += centred * inv_std[c];
  } else {
    unsafeAtomicAdd(g_beta + c, plain);
    unsafeAtomicAdd(g_gamma + c, centred * inv_std[c]);
  }
}

template <bool ORDERED>
__global__ __launch_bounds__(256) void bn_partial_reduce_kernel(const float* __restrict__ partial, int cols, int CO,
                                                                const float* __restrict__ inv_std,
                                                                float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                                int cols_per_segment) {
  __shared__ float scratch[2][256];
  const int first = (int)blockIdx.y * cols_per_segment, last = min(cols, first + cols_per_segment);
  bn_partial_reduce_body<ORDERED>(partial, cols, CO, inv_std, g_gamma, g_beta, first, last, (int)blockIdx.x, scratch);
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);
float* partial_workspace(size_t bytes, hipStream_t stream);
bool pointwise_ring_eligible(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, const float* bias,
                             const float* out, int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t HW, bool fused_pro,
                             const BnBackwardEpilogue* epilogue, int* tile_pixels);
int pointwise_ring_run(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, float* out, int64_t out_bs,
                       int32_t N, int32_t CI, int32_t CO, int32_t HW, int accumulate, hipStream_t stream,
                       const float* const* bn, const BnBackwardEpilogue* epilogue, float* epi_partial, int32_t epi_cols,
                       int tile_pixels, int32_t* rows_done);

// g_beta[c] += sum_t partial[0][t][c];  g_gamma[c] += inv_std[c] * sum_t partial[1][t][c]   (t = workgroup tiles)
void bn_partial_reduce_run(const float* partial, int tiles, int CO, const float* inv_std, float* g_gamma, float* g_beta,
                           hipStream_t stream) {
  if (!split_atomics_forced()) {           // ordered: one workgroup per 16 channels walks every tile
    hipLaunchKernelGGL(bn_partial_reduce_kernel<true>, dim3((CO + 15) / 16, 1), dim3(256), 0, stream, partial, tiles, CO, inv_std,
                       g_gamma, g_beta, tiles);
    return;
  }
  int segments = tiles / 32;
  segments = segments < 1 ? 1 : (segments > 64 ? 64 : segments);
  const int per = (tiles + segments - 1) / segments;
  hipLaunchKernelGGL(bn_partial_reduce_kernel<false>, dim3((CO + 63) / 64, (unsigned)((tiles + per - 1) / per)), dim3(256), 0, stream,
                     partial, tiles, CO, inv_std, g_gamma, g_beta, per);
}

// One launch for the deferred parameter sums of MANY convolutions (a dense block's backward: two per layer): job z of
// the table reduces partial[q][tile][c] (q = 0 beta, 1 gamma before inv_std; at scratch + partial_offset) like the kernel
// above.  The table lives in device memory and is built once per block by the caller (the offsets into the scratch
// region and the arena pointers do not change between steps).
struct BnReduceJob { int64_t partial_offset; int32_t tiles, channels; const float* inv_std; float* g_gamma; float* g_beta; };

template <bool ORDERED>
__global__ __launch_bounds__(256) void bn_partial_reduce_batched_kernel(const BnReduceJob* __restrict__ jobs,
                                                                        const float* __restrict__ scratch) {
  __shared__ float buffer[2][256];
  const BnReduceJob job = jobs[blockIdx.z];
  int first = 0, last = job.tiles;
  if (!ORDERED) {
    int segments = job.tiles / 32;
    segments = segments < 1 ? 1 : (segments > 64 ? 64 : segments);
    const int per = (job.tiles + segments - 1) / segments;
    first = (int)blockIdx.y * per;
    last = min(job.tiles, first + per);
  }
  if ((int)blockIdx.x * (ORDERED ? 16 : 64) >= job.channels || first >= job.tiles) return;        // (workgroup-uniform)
  bn_partial_reduce_body<ORDERED>(scratch + job.partial_offset, job.tiles, job.channels, job.inv_std, job.g_gamma, job.g_beta,
                                  first, last, (int)blockIdx.x, buffer);
}

int bn_partial_reduce_batched_run(const void* jobs, int count, int max_channels, int max_tiles, const float* scratch,
                                  hipStream_t stream) {
  if (!split_atomics_forced()) {
    hipLaunchKernelGGL(bn_partial_reduce_batched_kernel<true>, dim3((max_channels + 15) / 16, 1, count), dim3(256), 0, stream,
                       reinterpret_cast<const BnReduceJob*>(jobs), scratch);
    return launch_status();
  }
  int segments = max_tiles / 32;
  segments = segments < 1 ? 1 : (segments > 64 ? 64 : segments);
  hipLaunchKernelGGL(bn_partial_reduce_batched_kernel<false>, dim3((max_channels + 63) / 64, segments, count), dim3(256), 0, stream,
                     reinterpret_cast<const BnReduceJob*>(jobs), scratch);
  return launch_status();
}

// Rows of the per-workgroup parameter sums the fused epilogue of this geometry produces (per quantity).
int64_t pointwise_epilogue_tiles(int32_t N, int32_t HW) {
  const int64_t groups = (int64_t)N * ((HW + 31) / 32);
  return (groups + 3) / 4;
}

bool pointwise_enabled() {
  static const bool disabled = getenv("SRGAN_NO_POINTWISE") != nullptr;
  return !disabled;
}

template <int MI, int BK, int NI>
static void launch_pointwise(const PointwiseParams& p, dim3 grid, hipStream_t stream) {
  if (p.epi_x) {
    if constexpr (NI == 1 && MI <= 2) {
      if (p.epi_ragged) hipLaunchKernelGGL((pointwise_kernel<MI, BK, false, 1, true, true>), grid, dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((pointwise_kernel<MI, BK, false, 1, true>), grid, dim3(256), 0, stream, p);
    }
    return;
  }
  if (p.bn_mean) hipLaunchKernelGGL((pointwise_kernel<MI, BK, true, NI>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((pointwise_kernel<MI, BK, false, NI>), grid, dim3(256), 0, stream, p);
}

// bn (4 pointers: mean, inv_std, gamma, beta; NULL = none): the input is relu(batch_norm_eval(in)) on the fly.
int pointwise_run(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, const float* bias, float* out,
                  int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t HW, int accumulate, hipStream_t stream,
                  const float* const* bn, const BnBackwardEpilogue* epilogue, int* plan_only_split) {
  // accumulate: 0 store, 1 add to out, 2 out is already zero (a K split then skips its zero-fill; otherwise a store).
  // plan_only_split: when given, nothing is launched and the number of K splits this call would use is returned there.
  PointwiseParams p;
  p.epi_x = nullptr; p.epi_x_bs = 0; p.epi_partial = nullptr; p.epi_cols = 0; p.epi_ragged = 0;
  p.in = in; p.w = w; p.out = out; p.bias = bias;
  p.bn_mean = bn ? bn[0] : nullptr; p.bn_inv = bn ? bn[1] : nullptr;
  p.bn_gamma = bn ? bn[2] : nullptr; p.bn_beta = bn ? bn[3] : nullptr;
  p.N = N; p.CI = CI; p.CO = CO; p.HW = HW;
  p.in_bs = in_bs; p.out_bs = out_bs; p.w_so = w_so; p.w_si = w_si;

  // Whole 128-row tiles on planes of 64 / 128-pixel blocks: the LDS-DMA kernel of pointwise_ring.hip (both operands staged
  // by global_load_lds into a two-stage ring; 0.64-0.67 of the matrix peak with the fused prologue against 0.53-0.57 here);
  // the rows beyond the last whole tile (a data gradient has 64 + 32 l of them) follow in a launch of the kernel above.
  int ring_pixels = 0;
  if (pointwise_ring_eligible(in, in_bs, w, w_so, w_si, bias, out, out_bs, N, CI, CO, HW, bn != nullptr, epilogue, &ring_pixels)) {
    if (plan_only_split) {
      *plan_only_split = 1;
      return SRGAN_OK;
    }
    const int64_t col_blocks128 = (int64_t)N * ((HW + 127) / 128);
    float* epi_partial = nullptr;
    if (epilogue) {
      SRGAN_REQUIRE(!bn, SRGAN_EINVAL, "pointwise batch-norm backward epilogue: no prologue");
      if (epilogue->partial_out) {
        epi_partial = epilogue->partial_out;
      } else if (epilogue->g_gamma) {
        epi_partial = partial_workspace((size_t)2 * col_blocks128 * CO * sizeof(float), stream);
        SRGAN_REQUIRE(epi_partial, SRGAN_EINVAL, "pointwise batch-norm backward epilogue: register a workspace for this "
                      "stream first (srgan_set_workspace, >= srgan_workspace_bytes())");
      }
    }
    const int profile_slot = profile_bracket_begin(stream);
    int32_t rows_done = 0;
    if (const int status = pointwise_ring_run(in, in_bs, w, w_so, w_si, out, out_bs, N, CI, CO, HW, accumulate, stream, bn,
                                              epilogue, epi_partial, (int32_t)col_blocks128, ring_pixels, &rows_done)) {
      profile_bracket_end(profile_slot, stream, 0, 0, 0, 13, 128, ring_pixels, 1);      // close the bracket this call opened
      return status;
    }
    // (the ring kernel covers every row: its last 128-row tile may be partial)
    SRGAN_REQUIRE(rows_done == CO, SRGAN_EINVAL, "pointwise ring: rows left uncovered");
    if (epi_partial && !epilogue->partial_out)
      bn_partial_reduce_run(epi_partial, (int)col_blocks128, CO, epilogue->bn[1], epilogue->g_gamma, epilogue->g_beta, stream);
    const int status = launch_status();
    const int64_t pixels = (int64_t)N * HW;
    const int64_t b_elements = (int64_t)CI * pixels + (epilogue ? (int64_t)CO * pixels * (accumulate == 1 ? 2 : 1) : 0) +
                               ((!epilogue && accumulate == 1) ? (int64_t)CO * pixels : 0);
    profile_bracket_end(profile_slot, stream, CO, pixels, CI, 13, 128, ring_pixels, 1, 0, 0, b_elements);      // kind 13: pointwise_ring_kernel
    return status;
  }

  // Two 32-pixel groups per wave (64 x 64 wave tile): a tuning variant (SRGAN_PW_NI=2).  In isolation it is up to
  // 10 % faster on the K = 128 data gradients, inside the training step it measured 0.7 % slower: off by default.
  static const int ni_cap = getenv("SRGAN_PW_NI") ? atoi(getenv("SRGAN_PW_NI")) : 1;
  int ni = (HW % 64 == 0 && ni_cap >= 2 && !epilogue) ? 2 : 1;
  int64_t groups = (int64_t)N * ((HW + 32 * ni - 1) / (32 * ni));
  int64_t col_blocks = (groups + 3) / 4;
  // Tallest row tile that still yields ~4 workgroups per CU; otherwise shorter tiles, then split over input channels.
  static const int mi_cap_long = getenv("SRGAN_PW_MI") ? atoi(getenv("SRGAN_PW_MI")) : 2;
  static const int mi_cap_short = getenv("SRGAN_PW_MI_SHORT") ? atoi(getenv("SRGAN_PW_MI_SHORT")) : mi_cap_long;
  const int mi_cap = CI <= 128 ? mi_cap_short : mi_cap_long;
  int mi = CO > 64 ? 4 : (CO > 32 ? 2 : 1);
  if (mi > mi_cap) mi = mi_cap;
  if (epilogue && mi > 2) mi = 2;
  // NI = 2 pairs with the 64-row tile (acc 64 + 2 x 32 operand registers; 128 rows x 64 pixels would spill)
  if (ni == 2) {
    if (mi > 2) mi = 2;
    if (mi < 2 || col_blocks * ((CO + 63) / 64) < 1024) {
      ni = 1;
      groups = (int64_t)N * ((HW + 31) / 32);
      col_blocks = (groups + 3) / 4;
      mi = CO > 64 ? 4 : (CO > 32 ? 2 : 1);
      if (mi > mi_cap) mi = mi_cap;
    }
  }
  static const int min_wgs = getenv("SRGAN_PW_MIN_WGS") ? atoi(getenv("SRGAN_PW_MIN_WGS")) : 768;
  while (mi > 1 && col_blocks * ((CO + mi * 32 - 1) / (mi * 32)) < min_wgs) mi >>= 1;
  // Rows beyond the last full 128-row tile go to a second launch with a tile just tall enough for them (the data
  // gradients of the bottlenecks have 64 + 32*l rows: a padded 128-row tile would waste up to 3/8 of the matrix work).
  int rest = 0, rest_mi = 0;
  if (ni == 1 && mi == 4 && CO > 128 && CO % 128 != 0 && CO % 128 <= 64) {
    rest = CO % 128;
    rest_mi = rest <= 32 ? 1 : 2;
  }
  const int main_rows = CO - rest;
  p.tiles_m = (main_rows + mi * 32 - 1) / (mi * 32);
  p.m_base = 0;
  const int64_t blocks = col_blocks * p.tiles_m;
  const int slices = (CI + 63) / 64;
  int split = 1;
  static const int split_below = getenv("SRGAN_PW_SPLIT_BELOW") ? atoi(getenv("SRGAN_PW_SPLIT_BELOW")) : (min_wgs * 3) / 4;
  if (blocks < split_below && slices >= 2 && !epilogue) {   // (the fused epilogue needs whole sums per workgroup)
    split = (int)((min_wgs + blocks - 1) / blocks);
    if (split > slices) split = slices;
    if (split < 1) split = 1;
  }
  const int per = (slices + split - 1) / split;
  p.k_per_split = per * 64;
  split = (slices + per - 1) / per;
  SRGAN_REQUIRE(blocks < ((int64_t)1 << 31) && split <= 65535, SRGAN_ERANGE, "pointwise grid");
  if (plan_only_split) {
    *plan_only_split = split;
    return SRGAN_OK;
  }
  p.split_ws = nullptr; p.split_tickets = nullptr;
  if (split > 1) {
    // Ordered finish (split_finish.h): partial tiles through the workspace, summed in slice order by the tile's last
    // workgroup -- one launch, no zero-fill, the same bits every run.  Without a workspace: zero-fill + fp32 atomics.
    int ticket_set = -1;
    float* ws = split_workspace(blocks, split, (int64_t)mi * ni * 16 * 256, 0, stream, &ticket_set);
    unsigned int* tickets = ws ? device_tickets(g_pointwise_split_tickets) : nullptr;
    if (ws && tickets) {
      p.mode = accumulate == 1 ? 4 : 3;
      p.split_ws = ws;
      p.split_tickets = tickets + (size_t)ticket_set * SPLIT_TICKET_TILES;
    } else {
      if (accumulate == 0)
        if (const int status = zero_rows(out, out_bs, (int64_t)CO * HW, N, stream)) return status;
      p.mode = 2;
    }
  } else {
    p.mode = accumulate == 1 ? 1 : 0;
  }
  if (epilogue) {
    SRGAN_REQUIRE(!bn && !bias, SRGAN_EINVAL, "pointwise batch-norm backward epilogue: no prologue, no bias");
    SRGAN_REQUIRE(mi <= 2 && rest == 0, SRGAN_EUNSUPPORTED, "pointwise batch-norm backward epilogue: at most 64-row tiles");
    p.epi_ragged = ((((uintptr_t)epilogue->x | (uintptr_t)out) & 15) | ((epilogue->x_bs | out_bs | HW) & 3)) != 0 ? 1 : 0;
    p.epi_x = epilogue->x; p.epi_x_bs = epilogue->x_bs;
    p.bn_mean = epilogue->bn[0]; p.bn_inv = epilogue->bn[1]; p.bn_gamma = epilogue->bn[2]; p.bn_beta = epilogue->bn[3];
    p.epi_cols = (int32_t)col_blocks;
    if (epilogue->partial_out) {
      p.epi_partial = epilogue->partial_out;
    } else if (epilogue->g_gamma) {
      p.epi_partial = partial_workspace((size_t)2 * col_blocks * CO * sizeof(float), stream);
      SRGAN_REQUIRE(p.epi_partial, SRGAN_EINVAL, "pointwise batch-norm backward epilogue: register a workspace for this "
                    "stream first (srgan_set_workspace, >= srgan_workspace_bytes())");
    }
  }
  static const bool narrow = getenv("SRGAN_PW_NARROW_OUT") != nullptr;
  p.wide_out = (!narrow && (((uintptr_t)out & 15) | (out_bs & 3) | (HW & 3)) == 0) ? 1 : 0;
  p.gpi = (HW + 32 * ni - 1) / (32 * ni);
  const int profile_slot = profile_bracket_begin(stream);
  // (K slices of 64 measured no faster and spill at 128 rows: the slice is 32 channels.)
  auto launch = [&](int mi_, dim3 grid) {
    if (ni == 2) launch_pointwise<2, 32, 2>(p, grid, stream);
    else if (mi_ == 4) launch_pointwise<4, 32, 1>(p, grid, stream);
    else if (mi_ == 2) launch_pointwise<2, 32, 1>(p, grid, stream);
    else launch_pointwise<1, 32, 1>(p, grid, stream);
  };
  static const bool no_xcd = getenv("SRGAN_NO_XCD_ORDER") != nullptr;
  p.xcd_remap = (!no_xcd && p.tiles_m > 1 && blocks % 8 == 0) ? 1 : 0;
  launch(mi, dim3((unsigned)blocks, (unsigned)split, 1));
  if (rest > 0) {
    p.m_base = main_rows;
    p.tiles_m = 1;
    p.xcd_remap = 0;
    launch(rest_mi, dim3((unsigned)col_blocks, (unsigned)split, 1));
  }
  if (p.epi_partial && !epilogue->partial_out)
    bn_partial_reduce_run(p.epi_partial, (int)col_blocks, CO, p.bn_inv, epilogue->g_gamma, epilogue->g_beta, stream);
  const int status = launch_status();
  // algorithmic bytes of the fused epilogue: besides the operands, x is read for the mask and, in accumulate mode, the
  // gradient buffer is read as well as written
  const int64_t pixels = (int64_t)N * HW;
  const int64_t b_elements = (int64_t)CI * pixels + (epilogue ? (int64_t)CO * pixels * (p.mode == 1 ? 2 : 1) : 0) +
                             ((!epilogue && p.mode == 1) ? (int64_t)CO * pixels : 0);
  profile_bracket_end(profile_slot, stream, CO, pixels, CI, 3, mi * 32, 128, split, 0, 0, b_elements);
  return status;
}

}  // namespace srgan
