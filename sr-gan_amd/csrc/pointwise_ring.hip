// pointwise_ring.hip -- the 1x1 / stride 1 convolution with BOTH operands staged by LDS-DMA (round 4).
//
// Same contraction as pointwise.hip -- out[n, o, p] = sum_i w(o, i) * in[n, i, p] on NCHW data, reference
// crowd/models.py:340-341,369-370 and their data gradients -- for the shapes that dominate the 512 x 512 step: whole
// 128-row tiles of output channels, input channels in multiples of 32, planes in multiples of 64 / 128 pixels.  The
// calibration that led here is scratch/lab/gemm_lab.hip (profiles/r04_gemm_calibration.txt): on clean shapes the
// register-streamed structure of pointwise.hip tops out at 0.60-0.72 of the fp32 MFMA peak, this one at 0.67-0.78, and
// with the fused batch-norm prologue at 0.53-0.57 against 0.64-0.67.
//
//   * global_load_lds_dwordx4: a wave moves 1 KB per instruction straight into LDS (no VGPR round trip, no ds_write
//     pass); the loads of stage t + 1 are in flight during the whole matrix work of stage t, counted with
//     s_waitcnt vmcnt(N) and ONE raw s_barrier per 32-deep K stage (a __syncthreads() would drain the DMA queue).
//   * a workgroup owns 128 output rows x 128 (or 64) pixels; wave w owns rows 32w..32w+31 for ALL the pixels, so a
//     128-row convolution reads its activations exactly once, and the activation tile is shared through LDS.
//   * MFMA column block ni of a wave holds the pixels NI*j + ni: a lane's NI accumulators of one row are NI consecutive
//     pixels -- the B fragment is ONE ds_read_b128 / b64 per k step, and the tile leaves as float4 / float2 rows with no
//     trip through LDS; the batch-norm backward epilogue of a data gradient works on the same float4s.
//   * v_mfma_f32_32x32x2_f32 pairs two k per instruction (lanes 0-31 / 32-63); which two is free as long as A and B
//     agree.  Step 4g + t of a stage pairs k = 8g + t with k = 8g + 4 + t, so that with k-contiguous weights (the forward
//     convolution) a lane's four steps of a group come from ONE 16-byte read; the weight rows sit in LDS as 256-byte
//     bank rows whose 16-byte slots are XOR-swizzled by the bank-row index (applied to the DMA's SOURCE addresses: the
//     LDS image of an LDS-DMA is lane-linear) -- conflict-free ds_read_b128.  m-contiguous weights (the data gradient's
//     W^T) are staged as [k][128 rows] and read by conflict-free ds_read_b32.
//   * two stages of 32 channels (64 KB for the 128-pixel tile): two workgroups per CU interleave their barrier and
//     epilogue phases, which measured better than three or four stages at one workgroup per CU.
// Roofline: fp32 MFMA 157.3 TF/s; algorithmic bytes 4 * (CI + CO) per pixel (+ the epilogue streams of a fused data
// gradient).
#include <atomic>
#include "common.h"
#include "lds_dma.h"
#include <stdlib.h>
#include <type_traits>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct RingParams {
  const float* in;      // [N, CI, HW], batch stride in_bs
  const float* w;       // A_KCONTIG: element (o, i) at w[o * w_ld + i]; otherwise at w[i * w_ld + o]
  float* out;           // [N, CO, HW], batch stride out_bs
  int32_t N, CI, CO, HW;
  int64_t in_bs, out_bs;
  int32_t w_ld;
  int32_t bpi;          // pixel blocks (of 32 * NI) per image
  int32_t tiles_m;      // 128-row tiles (the last one may hold 32, 64 or 96 rows)
  int32_t xcd_remap;
  int32_t mode;         // 0 store, 1 accumulate
  // FUSE 1: the input is relu(batch_norm_eval(in)), bn_* per INPUT channel.  FUSE 2: the output rows go through the
  // backward of relu(batch_norm_eval(epi_x)) (bn_* per OUTPUT row), see pointwise.hip
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
  const float* epi_x; int64_t epi_x_bs;
  float* epi_partial; int32_t epi_cols;
};

// NI: 32-pixel column blocks per wave (4: 128-pixel tile; 2 / 1: 64- / 32-pixel tiles for launches with few pixel blocks).
// FUSE: 0 plain, 1 batch-norm + ReLU prologue on the activations, 2 batch-norm + ReLU backward epilogue (NI = 4).
template <int STAGES, int NI, bool A_KCONTIG, int FUSE>
__global__ __launch_bounds__(256, 2) void pointwise_ring_kernel(const RingParams p) {
  constexpr int BK = 32;
  constexpr int A_BYTES = 128 * BK * 4, RB = 128 * NI, B_BYTES = BK * RB, C_BYTES = FUSE == 1 ? 4 * 512 : 0;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES + C_BYTES;
  constexpr int QA = 4, QB = NI, RPI = 1024 / RB;
  constexpr int PER_STAGE = QA + QB + (FUSE == 1 ? 2 : 0);
  static_assert(FUSE != 2 || NI == 4, "the epilogue's partial sums are per 128-pixel block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = (int)blockIdx.x;
  if (p.xcd_remap) bid = (bid & 7) * ((int)gridDim.x >> 3) + (bid >> 3);     // the row tiles of a pixel block on one XCD
  const int tm = bid % p.tiles_m, cb = bid / p.tiles_m;
  const int n = cb / p.bpi, pix0 = (cb - n * p.bpi) * (32 * NI);
  const int m0 = tm * 128;
  const uint32_t lds0 = ring_lds_address(smem);
  const int nst = p.CI / BK;

  // ---- DMA sources ----
  // B: instruction q of wave w covers the RPI k rows (w * QB + q) * RPI .. of the stage, RB bytes each
  const char* in_block = (const char*)(p.in + (int64_t)n * p.in_bs + pix0);
  const int b_row = lane / (RB / 16), b_col = lane % (RB / 16);
  const uint32_t b_lane = (uint32_t)(b_row * p.HW + 4 * b_col) * 4u;
  uint32_t a_lane[QA];
  if (A_KCONTIG) {
    // weight rows of 32 floats; a 256-byte bank row holds two of them = 16 slots of 16 bytes, slot' = slot ^ (bank row & 15)
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int br = lane >> 4, sp = lane & 15;
      const int bank_row = 16 * wave + 4 * q + br;
      const int s = sp ^ (bank_row & 15);
      a_lane[q] = (uint32_t)(min(m0 + 2 * bank_row + (s >> 3), p.CO - 1) * p.w_ld + 4 * (s & 7)) * 4u;
    }
  } else {
#pragma unroll
    for (int q = 0; q < QA; ++q)
      a_lane[q] = (uint32_t)(((wave * QA + q) * 2 + lhi) * p.w_ld + min(m0 + 4 * l31, p.CO - 4)) * 4u;
  }
  // the last row tile may be partial (CO is a multiple of 32: whole waves): its surplus waves stage their share of the
  // operands (clamped rows) and keep the barriers, but skip the matrix work and the stores
  const bool active = m0 + 32 * wave < p.CO;
  // FUSE 1: a wave's own table of the stage's batch-norm vectors [mean: 32][inv_std: 32][gamma: 32][beta: 32]
  const float* c_lane0 = (lhi ? p.bn_inv : p.bn_mean) + l31;
  const float* c_lane1 = (lhi ? p.bn_beta : p.bn_gamma) + l31;
  auto issue = [&](int stage) {
    const uint32_t slot = lds0 + (uint32_t)(stage % STAGES) * STAGE_BYTES;
    const char* xb = in_block + (int64_t)(stage * BK + wave * QB * RPI) * p.HW * 4;
#pragma unroll
    for (int q = 0; q < QB; ++q)
      ring_glds16(xb + (int64_t)(q * RPI) * p.HW * 4, b_lane, slot + A_BYTES + (uint32_t)((wave * QB + q) * RPI) * RB);
    const char* wb = (const char*)p.w + (A_KCONTIG ? (int64_t)stage * BK * 4 : (int64_t)stage * BK * p.w_ld * 4);
#pragma unroll
    for (int q = 0; q < QA; ++q)
      ring_glds16(wb, a_lane[q], slot + (A_KCONTIG ? (uint32_t)(4096 * wave + 1024 * q) : (uint32_t)((wave * QA + q) * 1024)));
    if (FUSE == 1) {
      ring_glds4(c_lane0 + stage * BK, slot + A_BYTES + B_BYTES + wave * 512);
      ring_glds4(c_lane1 + stage * BK, slot + A_BYTES + B_BYTES + wave * 512 + 256);
    }
  };

  // ---- fragment read offsets inside a stage slot ----
  uint32_t a_read[4];
  if (A_KCONTIG) {
    const int row = 32 * wave + l31, bank_row = row >> 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) a_read[g] = (uint32_t)(bank_row * 256 + ((((row & 1) * 8 + 2 * g + lhi) ^ (bank_row & 15)) * 16));
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) a_read[g] = (uint32_t)((8 * g + 4 * lhi) * 512 + (32 * wave + l31) * 4);
  }
  const uint32_t b_read = (uint32_t)(A_BYTES + (4 * lhi) * RB + l31 * (4 * NI));
  const uint32_t c_read = (uint32_t)(A_BYTES + B_BYTES + wave * 512 + 16 * lhi);

  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  // One group = four MFMA steps = the k values 8g + 4*lhi + 0..3 of the stage.  The fragments of group g + 1 are read into
  // a second register set while group g's matrix work issues.
  struct Group { float a[4]; float b[4][NI]; f32x4 mean, inv, gamma, beta; };
  auto load_group = [&](Group& f, const char* slot, int g) {
    if (A_KCONTIG) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slot + a_read[g]);
      f.a[0] = v.x; f.a[1] = v.y; f.a[2] = v.z; f.a[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) f.a[q] = *reinterpret_cast<const float*>(slot + a_read[g] + q * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (NI == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slot + b_read + (8 * g + q) * RB);
        f.b[q][0] = v.x; f.b[q][1] = v.y; f.b[q][2] = v.z; f.b[q][3] = v.w;
      } else if constexpr (NI == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(slot + b_read + (8 * g + q) * RB);
        f.b[q][0] = v.x; f.b[q][1] = v.y;
      } else {
        f.b[q][0] = *reinterpret_cast<const float*>(slot + b_read + (8 * g + q) * RB);
      }
    }
    if (FUSE == 1) {
      f.mean = *reinterpret_cast<const f32x4*>(slot + c_read + 32 * g);
      f.inv = *reinterpret_cast<const f32x4*>(slot + c_read + 128 + 32 * g);
      f.gamma = *reinterpret_cast<const f32x4*>(slot + c_read + 256 + 32 * g);
      f.beta = *reinterpret_cast<const f32x4*>(slot + c_read + 384 + 32 * g);
    }
  };
  auto compute_group = [&](const Group& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float b[NI];
      if (FUSE == 1) {
        float ca, cb;
        bn_coefficients(f.mean[q], f.inv[q], f.gamma[q], f.beta[q], ca, cb);     // the forward's own (a, b): same ReLU mask
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = fmaxf(fmaf(f.b[q][ni], ca, cb), 0.f);
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = f.b[q][ni];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q], b[ni], acc[ni], 0, 0, 0);
    }
  };

  // ---- epilogue operands ----
  const int64_t row0 = m0 + 32 * wave + 4 * lhi;                       // + (r & 3) + 8 * (r >> 2)
  float* out_lane = p.out + (int64_t)n * p.out_bs + row0 * p.HW + pix0 + NI * l31;
  // FUSE 2: the epilogue's own streams -- x for the mask, the accumulated gradient -- are requested while the K loop runs
  // (after the stage-1 / stage-2 DMAs: 128 registers that this kernel has to spare at two workgroups per CU).  The pass
  // moves 4 * (128 + 3 * rows) bytes per pixel for 2 * 128 * rows FLOP: HBM-bound, so what matters is that the loads of
  // a tile overlap its matrix phase instead of following it.
  const float* x_lane = FUSE == 2 ? p.epi_x + (int64_t)n * p.epi_x_bs + row0 * p.HW + pix0 + 4 * l31 : nullptr;
  f32x4 xs[FUSE == 2 ? 16 : 1], olds[FUSE == 2 ? 16 : 1];
  const bool prefetch = FUSE == 2 && STAGES == 2 && nst >= 3 && active;
  const bool accumulate = p.mode == 1;

  for (int s = 0; s < STAGES - 1 && s < nst; ++s) issue(s);
  for (int t = 0; t < nst; ++t) {
    // stage t has landed once at most the (STAGES - 2) younger stages of this wave are still in flight (plus, FUSE 2, the 16
    // prefetch loads issued behind stage t's DMAs); the barrier also tells that every wave is done reading the slot that is
    // refilled next
    if (FUSE == 2 && prefetch && (t == 1 || (t == 2 && accumulate))) ring_wait_and_barrier<16>();
    else if (nst - 1 - t >= STAGES - 2) ring_wait_and_barrier<PER_STAGE * (STAGES - 2)>();
    else ring_wait_and_barrier<0>();
    if (t + STAGES - 1 < nst) issue(t + STAGES - 1);
    if constexpr (FUSE == 2) {
      if (prefetch && t == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[r] = *reinterpret_cast<const f32x4*>(x_lane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.HW);
      }
      if (prefetch && t == 1 && accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) olds[r] = *reinterpret_cast<const f32x4*>(out_lane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.HW);
      }
    }
    if (!active) continue;
    const char* slot = smem + (t % STAGES) * STAGE_BYTES;
    Group f0, f1;
    load_group(f0, slot, 0);
    load_group(f1, slot, 1);
    compute_group(f0);
    load_group(f0, slot, 2);
    compute_group(f1);
    load_group(f1, slot, 3);
    compute_group(f0);
    compute_group(f1);
  }

  // ---- epilogue: a lane holds NI consecutive pixels of 16 rows ----
  if constexpr (FUSE == 2) {
    // out (=, +=) acc * [fma(x, a, b) > 0] * a per row; per row the sums of the masked values and of masked * (x - mean)
    // over the tile's 128 pixels (complete inside this wave: five DPP adds per value), one partial per (pixel block, row)
    __syncthreads();                                                    // (no DMA in flight: the last stage waited vmcnt(0))
    float* table = reinterpret_cast<float*>(smem);                      // [128 rows][4]: a, b, mean
    if (tid < 128) {
      const int o = min(m0 + tid, p.CO - 1);
      const float mu = p.bn_mean[o];
      float a, b;
      bn_coefficients(mu, p.bn_inv[o], p.bn_gamma[o], p.bn_beta[o], a, b);
      table[tid * 4 + 0] = a; table[tid * 4 + 1] = b; table[tid * 4 + 2] = mu;
    }
    __syncthreads();
    if (!active) return;
    const bool sums_wanted = p.epi_partial != nullptr;
    float* partial = p.epi_partial + (int64_t)cb * p.CO + row0;
    auto rows = [&](auto accumulating) {
#pragma unroll
      for (int batch = 0; batch < 4; ++batch) {                         // four rows at a time: loads first, stores last
        if (!prefetch) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t offset = (int64_t)(e + 8 * batch) * p.HW;      // r = 4 * batch + e: row (r & 3) + 8 * (r >> 2)
            xs[4 * batch + e] = *reinterpret_cast<const f32x4*>(x_lane + offset);
            if constexpr (decltype(accumulating)::value) olds[4 * batch + e] = *reinterpret_cast<const f32x4*>(out_lane + offset);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * batch + e, local = e + 8 * batch;
          const f32x4 t = *reinterpret_cast<const f32x4*>(table + (32 * wave + 4 * lhi + local) * 4);
          const f32x4 x = xs[r];
          f32x4 v;
          v.x = fmaf(x.x, t.x, t.y) > 0.f ? acc[0][r] : 0.f;
          v.y = fmaf(x.y, t.x, t.y) > 0.f ? acc[1][r] : 0.f;
          v.z = fmaf(x.z, t.x, t.y) > 0.f ? acc[2][r] : 0.f;
          v.w = fmaf(x.w, t.x, t.y) > 0.f ? acc[3][r] : 0.f;
          f32x4 result = v * t.x;
          f32x4* dst = reinterpret_cast<f32x4*>(out_lane + (int64_t)local * p.HW);
          if constexpr (decltype(accumulating)::value) *dst = result + olds[r];
          else __builtin_nontemporal_store(result, dst);
          if (sums_wanted) {
            const float plain = half_wave_sum((v.x + v.y) + (v.z + v.w));
            const float centred = half_wave_sum((v.x * (x.x - t.z) + v.y * (x.y - t.z)) + (v.z * (x.z - t.z) + v.w * (x.w - t.z)));
            if (l31 == 31) {
              partial[local] = plain;
              partial[(int64_t)p.epi_cols * p.CO + local] = centred;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);                               // keeps a batch's loads from joining the previous one's
      }
    };
    if (accumulate) rows(std::true_type{});
    else rows(std::false_type{});
    return;
  }
  if (!active) return;
  if (p.mode == 1) {
    // accumulate: the sixteen rows' old values in one batch, then the stores (loads and stores return through one in-order
    // counter, vmcnt: `+=` row by row made every load wait for the store in front of it)
    float previous[16][NI];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* src = out_lane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.HW;
      if constexpr (NI == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        previous[r][0] = v.x; previous[r][1] = v.y; previous[r][2] = v.z; previous[r][3] = v.w;
      } else if constexpr (NI == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src);
        previous[r][0] = v.x; previous[r][1] = v.y;
      } else {
        previous[r][0] = *src;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* dst = out_lane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.HW;
      if constexpr (NI == 4) {
        f32x4 v;
        v.x = previous[r][0] + acc[0][r]; v.y = previous[r][1] + acc[1][r]; v.z = previous[r][2] + acc[2][r]; v.w = previous[r][3] + acc[3][r];
        *reinterpret_cast<f32x4*>(dst) = v;
      } else if constexpr (NI == 2) {
        f32x2 v;
        v.x = previous[r][0] + acc[0][r]; v.y = previous[r][1] + acc[1][r];
        *reinterpret_cast<f32x2*>(dst) = v;
      } else {
        *dst = previous[r][0] + acc[0][r];
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* dst = out_lane + (int64_t)((r & 3) + 8 * (r >> 2)) * p.HW;
    if constexpr (NI == 4) {
      f32x4 v;
      v.x = acc[0][r]; v.y = acc[1][r]; v.z = acc[2][r]; v.w = acc[3][r];
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));             // consumed by a later kernel
    } else if constexpr (NI == 2) {
      f32x2 v;
      v.x = acc[0][r]; v.y = acc[1][r];
      __builtin_nontemporal_store(v, reinterpret_cast<f32x2*>(dst));
    } else {
      __builtin_nontemporal_store(acc[0][r], dst);
    }
  }
}

bool pointwise_ring_enabled() {
  static const bool disabled = getenv("SRGAN_NO_PW_RING") != nullptr;
  return !disabled;
}

// Whether the LDS-DMA kernel takes the 128-row tiles of this convolution (the caller sends the remaining rows, if any,
// to pointwise_kernel); *tile_pixels = 128, 64 or 32.  Everything the DMA addresses must be 16-byte aligned.
bool pointwise_ring_eligible(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, const float* bias,
                             const float* out, int64_t out_bs, int32_t N, int32_t CI, int32_t CO, int32_t HW, bool fused_pro,
                             const BnBackwardEpilogue* epilogue, int* tile_pixels) {
  if (!pointwise_ring_enabled() || bias || CO < 128 || CO % 32 != 0 || CI % 32 != 0 || CI < 32 || HW % 32 != 0) return false;
  if (!(w_si == 1 && w_so % 4 == 0) && !(w_so == 1 && w_si % 4 == 0)) return false;
  if ((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) || (in_bs & 3) || (out_bs & 3)) return false;
  // A data gradient with the batch-norm backward epilogue moves 4 * (128 + 3 * rows) bytes per pixel for 2 * 128 * rows
  // FLOP: HBM-bound on either kernel.  Here the epilogue's x / old-gradient rows are requested during the K loop and the
  // last row tile may be partial (no second launch): 37.5 ms per step against 39.9 ms on pointwise_kernel (profiles/r04e).
  static const bool no_epilogue = getenv("SRGAN_NO_PW_RING_EPILOGUE") != nullptr;
  if (epilogue && (no_epilogue || HW % 128 != 0 || (((uintptr_t)epilogue->x) & 15) || (epilogue->x_bs & 3))) return false;
  // Few pixel blocks: narrower tiles multiply the workgroups (16 images of 32 x 32 pixels, 128 rows: 43 us on the 128-pixel
  // tile = half the CUs idle, 30 us on the 64-pixel tile); the epilogue's partial sums are per 128-pixel block.
  static const int narrow_below = getenv("SRGAN_PW_RING_NARROW_BELOW") ? atoi(getenv("SRGAN_PW_RING_NARROW_BELOW")) : 512;
  static const int slim_below = getenv("SRGAN_PW_RING_SLIM_BELOW") ? atoi(getenv("SRGAN_PW_RING_SLIM_BELOW")) : 384;
  static const int min_blocks = getenv("SRGAN_PW_RING_MIN_WGS") ? atoi(getenv("SRGAN_PW_RING_MIN_WGS")) : 192;
  const int64_t tiles = (int64_t)N * ((CO + 127) / 128);
  int pixels = 128;
  if (epilogue) pixels = 128;
  else if (HW % 128 != 0 || tiles * (HW / 128) < narrow_below) pixels = 64;
  if (!epilogue && (HW % 64 != 0 || (pixels == 64 && tiles * (HW / 64) < slim_below))) pixels = 32;
  // too few workgroups even so: the K-split kernel / short tiles of pointwise.hip do better
  if (tiles * (HW / pixels) < min_blocks) return false;
  (void)fused_pro;
  *tile_pixels = pixels;
  return true;
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

template <int NI, bool A_KCONTIG, int FUSE>
static int launch_ring(const RingParams& p, unsigned blocks, hipStream_t stream) {
  constexpr int STAGES = 2;
  constexpr int bytes = STAGES * (128 * 32 * 4 + 32 * 128 * NI + (FUSE == 1 ? 2048 : 0));
  auto kernel = pointwise_ring_kernel<STAGES, NI, A_KCONTIG, FUSE>;
  // More than 64 KB of dynamic LDS needs the attribute -- once per kernel AND per device (the attribute belongs to the
  // device's copy of the function); a bit per device ordinal, set with an atomic OR so that two host threads launching
  // for the first time together at worst both set the (idempotent) attribute.
  static std::atomic<uint64_t> configured_devices{0};
  int device = 0;
  SRGAN_HIP(hipGetDevice(&device));
  const uint64_t bit = (uint64_t)1 << (device & 63);
  if (!(configured_devices.load(std::memory_order_acquire) & bit)) {
    SRGAN_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    configured_devices.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), bytes, stream, p);
  return launch_status();
}

// Every row of the convolution (the last 128-row tile may be partial); *rows_done = the rows it covered.
int pointwise_ring_run(const float* in, int64_t in_bs, const float* w, int32_t w_so, int32_t w_si, float* out, int64_t out_bs,
                       int32_t N, int32_t CI, int32_t CO, int32_t HW, int accumulate, hipStream_t stream,
                       const float* const* bn, const BnBackwardEpilogue* epilogue, float* epi_partial, int32_t epi_cols,
                       int tile_pixels, int32_t* rows_done) {
  RingParams p;
  p.in = in; p.w = w; p.out = out;
  p.N = N; p.CI = CI; p.CO = CO; p.HW = HW; p.in_bs = in_bs; p.out_bs = out_bs;
  const bool k_contiguous = w_si == 1;
  p.w_ld = k_contiguous ? w_so : w_si;
  p.tiles_m = (CO + 127) / 128;
  p.bpi = HW / tile_pixels;
  p.mode = accumulate == 1 ? 1 : 0;
  p.bn_mean = p.bn_inv = p.bn_gamma = p.bn_beta = nullptr;
  p.epi_x = nullptr; p.epi_x_bs = 0; p.epi_partial = nullptr; p.epi_cols = 0;
  if (bn) { p.bn_mean = bn[0]; p.bn_inv = bn[1]; p.bn_gamma = bn[2]; p.bn_beta = bn[3]; }
  if (epilogue) {
    p.bn_mean = epilogue->bn[0]; p.bn_inv = epilogue->bn[1]; p.bn_gamma = epilogue->bn[2]; p.bn_beta = epilogue->bn[3];
    p.epi_x = epilogue->x; p.epi_x_bs = epilogue->x_bs; p.epi_partial = epi_partial; p.epi_cols = epi_cols;
  }
  const int64_t blocks = (int64_t)N * p.bpi * p.tiles_m;
  SRGAN_REQUIRE(blocks < ((int64_t)1 << 31), SRGAN_ERANGE, "pointwise ring grid");
  static const bool no_xcd = getenv("SRGAN_NO_XCD_ORDER") != nullptr;
  p.xcd_remap = (!no_xcd && p.tiles_m > 1 && blocks % 8 == 0) ? 1 : 0;
  *rows_done = CO;
  const unsigned grid = (unsigned)blocks;
  const int fuse = epilogue ? 2 : (bn ? 1 : 0);
  if (fuse == 2) return k_contiguous ? launch_ring<4, true, 2>(p, grid, stream) : launch_ring<4, false, 2>(p, grid, stream);
  if (tile_pixels == 32) {
    if (fuse == 1) return k_contiguous ? launch_ring<1, true, 1>(p, grid, stream) : launch_ring<1, false, 1>(p, grid, stream);
    return k_contiguous ? launch_ring<1, true, 0>(p, grid, stream) : launch_ring<1, false, 0>(p, grid, stream);
  }
  if (tile_pixels == 64) {
    if (fuse == 1) return k_contiguous ? launch_ring<2, true, 1>(p, grid, stream) : launch_ring<2, false, 1>(p, grid, stream);
    return k_contiguous ? launch_ring<2, true, 0>(p, grid, stream) : launch_ring<2, false, 0>(p, grid, stream);
  }
  if (fuse == 1) return k_contiguous ? launch_ring<4, true, 1>(p, grid, stream) : launch_ring<4, false, 1>(p, grid, stream);
  return k_contiguous ? launch_ring<4, true, 0>(p, grid, stream) : launch_ring<4, false, 0>(p, grid, stream);
}

}  // namespace srgan
