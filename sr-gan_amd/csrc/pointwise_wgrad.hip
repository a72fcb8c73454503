// pointwise_wgrad.hip -- weight gradient of 1x1 / stride 1 convolutions (the DenseNet bottleneck and transition
// convolutions, reference crowd/models.py:340-341,369-370; autograd's conv weight gradient behind loss.backward(),
// srgan.py:264,295,304):
//
//   gw[co, ci] += sum_{n, p} gy[n, co, p] * x[n, ci, p]
//
// Both operands are rows of pixels and the reduction runs ALONG the rows, so neither needs LDS: with
// v_mfma_f32_32x32x2_f32 (A[i][k]: lane = i + 32*k, B[k][j]: lane = j + 32*k) lane (i, half) simply owns 16 consecutive
// pixels of row i -- half 0 the first 16 of a 32-pixel chunk, half 1 the last 16 -- as four 16-byte loads, and the
// 16 MFMA steps of the chunk walk those registers (the order of a sum does not matter as long as A and B agree on
// it).  Every 128-byte line is fetched by exactly one load instruction and used completely.  Each WAVE is an
// independent worker over its own range of pixel chunks for the workgroup's 64 x 64 output tile; the next chunk is
// in flight while the current one is in the matrix pipe (explicit ping-pong register sets).  The four waves' tiles
// are summed through LDS and leave as one coalesced pass per workgroup: round 5 -- into the caller's workspace (every
// K-slice worker of an output tile keeps its own 64 x 64 partial there, `pointwise_wgrad_finish_kernel` adds a tile's
// partials in slice order and accumulates into gw: the same bits on every run); fp32 atomics into gw only when the stream
// has no workspace (or SRGAN_ATOMIC_SPLIT=1).
// The generic gather-GEMM staged both operands through LDS with a transpose and reached 55 TF/s on these shapes.
//
// Round 5, the staged form (`pointwise_wgrad_lds_*`) for planes of whole 32-pixel chunks and 16-byte aligned rows: the
// register-streamed kernel above gives every WAVE its own 64 x 64 tile, i.e. 128 operand rows per 64 x 64 x 32 block of
// matrix work, and leaves the sharing of rows between tiles to L2 -- which did not happen: 115 GB of fabric traffic per
// step for 45 GB of operands (profiles/r05z_crowd512_pmc_per_kernel.md), the kernel sat at 4.3 TB/s.  Here a workgroup
// owns a 128 x 128 tile: a 32-pixel chunk of its 128 gy rows and 128 x rows goes to LDS ONCE by LDS-DMA (32 KB per stage,
// two stages, two workgroups per CU) and wave w multiplies gy rows 32w..32w+31 by all 128 x rows: 64 operand rows per
// block of matrix work whatever L2 does (counters: 1.22x the operand bytes, 23.7 ms per step at 0.59 of the fp32 MFMA peak).  A lane still owns 16 consecutive pixels of its row (four ds_read_b128 per
// chunk and row block); rows are 128 bytes = half a 256-byte bank row, 16-byte slots XOR-swizzled by the bank-row index
// (applied to the DMA's source addresses, as in pointwise_ring.hip) -- conflict-free.
#include "common.h"
#include "lds_dma.h"
#include "split_finish.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PwWgradParams {
  const float* x; const float* gy; float* gw;
  int64_t x_bs, gy_bs;
  int32_t N, CI, CO, HW;
  int32_t tiles_n;               // ci tiles
  int32_t chunks, chunks_per_worker, chunks_per_image;
  int32_t mode;                  // 1 accumulate (single K-slice per tile), 2 atomic
  float* partial;                // non-NULL: K-slice `y` of tile `t` stores its 64 x 64 block at partial[(t * split + y) * 4096 ...]
  int32_t split;                 //   and pointwise_wgrad_finish adds the slices in order (no atomics)
  // x is relu(batch_norm_eval(x)) computed on the fly (per input channel) when bn_mean != NULL
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
};

constexpr int PWG_MI = 2, PWG_NI = 2;      // 64 x 64 output tile per workgroup

// RAGGED: planes whose size is not a multiple of 32 pixels (28 x 28, 14 x 14, 7 x 7 at the reference's 224 x 224) or
// whose rows are only 4-byte aligned.  An image is ceil(HW / 32) chunks; a float4 that would cross the end of the row
// is loaded from the row's last four pixels instead (the SAME pixels for both operands), and the elements that repeat
// an earlier float4 of the lane are switched off in the gy operand when it is used -- a zero times a finite value.
template <bool PRO, bool RAGGED>
__device__ __forceinline__ void pointwise_wgrad_body(const PwWgradParams& p, const int block_x, const int block_y, float* red) {
  constexpr int MI = PWG_MI, NI = PWG_NI, ROWS = MI * 32, COLS = NI * 32, LDR = COLS + 1;

  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int tm = block_x / p.tiles_n, tn = block_x - tm * p.tiles_n;
  const int co0 = tm * ROWS, ci0 = tn * COLS;
  const int worker = block_y * 4 + wave;
  const int cbeg = worker * p.chunks_per_worker;
  const int cend = min(p.chunks, cbeg + p.chunks_per_worker);

  uint32_t a_row[MI], b_row[NI];            // element offset of this lane's 16-pixel run inside an image
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a_row[mi] = (uint32_t)min(co0 + mi * 32 + l31, p.CO - 1) * (uint32_t)p.HW + 16u * lhi;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_row[ni] = (uint32_t)min(ci0 + ni * 32 + l31, p.CI - 1) * (uint32_t)p.HW + 16u * lhi;

  float pro_a[NI], pro_b[NI];               // PRO: batch-norm + ReLU of this lane's input-channel rows
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    pro_a[ni] = 1.f; pro_b[ni] = 0.f;
    if (PRO) {
      const int c = min(ci0 + ni * 32 + l31, p.CI - 1);
      bn_coefficients(p.bn_mean[c], p.bn_inv[c], p.bn_gamma[c], p.bn_beta[c], pro_a[ni], pro_b[ni]);
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  float4 a0[MI][4], b0[NI][4], a1[MI][4], b1[NI][4];
  uint32_t live0 = 0xFFFFu, live1 = 0xFFFFu;               // RAGGED: bit 4 * q + e = element e of float4 q is a new pixel
  auto fetch = [&](int chunk, float4 (&a)[MI][4], float4 (&b)[NI][4], uint32_t& live) {
    chunk = min(chunk, p.chunks - 1);                      // clamped: a worker's surplus fetch is never used
    const int n = chunk / p.chunks_per_image;
    const uint32_t pix = (uint32_t)(chunk - n * p.chunks_per_image) * 32u;
    const float* ga = p.gy + (int64_t)n * p.gy_bs + (RAGGED ? 0u : pix);
    const float* xb = p.x + (int64_t)n * p.x_bs + (RAGGED ? 0u : pix);
    uint32_t within[4];                                    // RAGGED: pixel offset of float4 q inside the row
    if (RAGGED) {
      live = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int want = (int)pix + 16 * lhi + 4 * q, last = p.HW - 4;
        const int repeated = min(max(want - last, 0), 4);  // leading elements of the clamped float4 already seen
        within[q] = (uint32_t)min(want, last) - 16u * lhi; // (a_row / b_row already hold the lane half's 16)
        live |= ((0xFu << repeated) & 0xFu) << (4 * q);
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) a[mi][q] = *reinterpret_cast<const float4*>(ga + a_row[mi] + (RAGGED ? within[q] : 4u * q));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) b[ni][q] = *reinterpret_cast<const float4*>(xb + b_row[ni] + (RAGGED ? within[q] : 4u * q));
  };
  auto compute = [&](const float4 (&a)[MI][4], const float4 (&b)[NI][4], const uint32_t live) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float av[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          av[mi] = e == 0 ? a[mi][q].x : (e == 1 ? a[mi][q].y : (e == 2 ? a[mi][q].z : a[mi][q].w));
          if (RAGGED) av[mi] = (live >> (4 * q + e)) & 1u ? av[mi] : 0.f;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bv[ni] = e == 0 ? b[ni][q].x : (e == 1 ? b[ni][q].y : (e == 2 ? b[ni][q].z : b[ni][q].w));
          if (PRO) bv[ni] = fmaxf(fmaf(bv[ni], pro_a[ni], pro_b[ni]), 0.f);     // at use time: the loads stay in flight
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  };

  if (cbeg < cend) {
    fetch(cbeg, a0, b0, live0);
    for (int c = cbeg; c < cend; c += 2) {
      fetch(c + 1, a1, b1, live1);
      compute(a0, b0, live0);
      if (c + 1 < cend) {
        fetch(c + 2, a0, b0, live0);
        compute(a1, b1, live1);
      }
    }
  }

  // ---- sum the four waves' tiles through LDS, then one coalesced pass out.  C/D fragment: column = lane & 31, row =
  // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  Two LDS tiles: waves 2, 3 deposit, waves 0, 1 add them to their
  // registers and deposit the pair sums, all 256 threads add the two tiles.
  float* mine = red + (wave & 1) * (ROWS * LDR);
  auto at = [&](int mi, int ni, int r) { return (mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDR + ni * 32 + l31; };
  if (wave >= 2) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[at(mi, ni, r)] = acc[mi][ni][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += mine[at(mi, ni, r)];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[at(mi, ni, r)] = acc[mi][ni][r];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < ROWS * COLS / 256; ++e) {
    const int idx = tid + 256 * e;
    const int row = idx / COLS, col = idx - row * COLS;
    const float v = red[row * LDR + col] + red[ROWS * LDR + row * LDR + col];
    if (p.partial) {               // (the whole tile, coalesced: rows / columns outside the matrix hold zeros and are never read back)
      p.partial[((int64_t)block_x * p.split + block_y) * (ROWS * COLS) + idx] = v;
      continue;
    }
    if (co0 + row < p.CO && ci0 + col < p.CI) {
      float* dst = p.gw + (int64_t)(co0 + row) * p.CI + ci0 + col;
      if (p.mode == 2) unsafeAtomicAdd(dst, v);
      else *dst += v;
    }
  }
}

// The staged form's column tiles: the 32-channel blocks of CI spread EVENLY over the ceil(blocks / 4) tiles of a row (160
// input channels = 3 + 2 blocks, not 4 + 1: a tile's gy rows cost the same however few x rows it multiplies them with).
__device__ __forceinline__ void pointwise_wgrad_lds_columns(const int CI, const int tiles_n, const int tn, int& ci0, int& blocks) {
  const int all = (CI + 31) >> 5, base = all / tiles_n, more = all - base * tiles_n;
  blocks = base + (tn < more ? 1 : 0);
  ci0 = 32 * (tn * base + min(tn, more));
}

// Second stage of the ordered form: gw[tile] += the tile's K-slice partials, added in slice order.  One thread per element
// of the ROWS x COLS tile (`slab` = which 256 of them), the slices read with lanes along the elements.
template <int ROWS, int COLS, bool STAGED>
__device__ __forceinline__ void pointwise_wgrad_finish_tile(const float* __restrict__ partial, float* __restrict__ gw, int tile,
                                                            int slab, int split, int tiles_n, int CO, int CI) {
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int idx = slab * 256 + (int)threadIdx.x;
  const int row = idx / COLS, col = idx - row * COLS;
  int co0 = tm * ROWS, ci0 = tn * COLS, width = COLS;
  if (STAGED) {
    int blocks;
    pointwise_wgrad_lds_columns(CI, tiles_n, tn, ci0, blocks);
    width = 32 * blocks;
  }
  if (co0 + row >= CO || col >= width || ci0 + col >= CI) return;
  const float* mine = partial + (int64_t)tile * split * (ROWS * COLS) + idx;
  float total = mine[0];
  int y = 1;
  for (; y + 3 < split; y += 4) {                // four loads in flight, one fixed order
    const float a = mine[(int64_t)y * (ROWS * COLS)], b = mine[(int64_t)(y + 1) * (ROWS * COLS)];
    const float c = mine[(int64_t)(y + 2) * (ROWS * COLS)], d = mine[(int64_t)(y + 3) * (ROWS * COLS)];
    total = (((total + a) + b) + c) + d;
  }
  for (; y < split; ++y) total += mine[(int64_t)y * (ROWS * COLS)];
  gw[(int64_t)(co0 + row) * CI + ci0 + col] += total;
}

constexpr int PWG_FINISH_SLABS = PWG_MI * 32 * PWG_NI * 32 / 256;       // 16
constexpr int PWL_TILE = 128, PWL_FINISH_SLABS = PWL_TILE * PWL_TILE / 256;   // the staged form: 128 x 128 tiles, 64 slabs

template <int TILE>
__global__ __launch_bounds__(256) void pointwise_wgrad_finish_kernel(const PwWgradParams p) {
  pointwise_wgrad_finish_tile<TILE, TILE, TILE == PWL_TILE>(p.partial, p.gw, (int)blockIdx.y, (int)blockIdx.x, p.split, p.tiles_n, p.CO, p.CI);
}

// ---- the staged form (see the head of the file).  NI: the tile's 32-column blocks that hold input channels (the last
// tile of a row may hold 1..3); wave w stages the x rows of block w and its own 32 gy rows.
// A stage = a 32-pixel chunk of the 128 gy rows (16 KB) + of the tile's 32 * NI x rows; two stages, three for the narrow tiles
// (NI <= 2, where three fit beside a second workgroup; measured together with the even spread of the column blocks: 25.9 ->
// 25.2 ms per step.  More chunks in flight on the full tiles -- gy rows in registers, four stages of x -- measured SLOWER:
// profiles/r05t_staged_1x1_weight_gradient.md).
constexpr int PWL_A_BYTES = PWL_TILE * 128, PWL_LDS_BYTES = 3 * (PWL_A_BYTES + 2 * 4096);     // 72 KB (>= 2 * 32 KB)

template <bool PRO, int NI>
__device__ __forceinline__ void pointwise_wgrad_lds_body(const PwWgradParams& p, const int block_x, const int block_y, const int co0,
                                                         const int ci0, char* smem) {
  constexpr int STAGES = NI <= 2 ? 3 : 2, PWL_STAGE_BYTES = PWL_A_BYTES + (NI <= 2 ? 2 : 4) * 4096;
  static_assert(STAGES * PWL_STAGE_BYTES <= PWL_LDS_BYTES, "the launch's LDS");
  const int tid = (int)threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbeg = block_y * p.chunks_per_worker;
  const int count = min(p.chunks - cbeg, p.chunks_per_worker);
  const uint32_t lds0 = ring_lds_address(smem);

  // DMA sources: instruction q of wave w fills the bank rows 16w + 4q .. + 3 (eight operand rows) of its operand; lane ->
  // (bank row, 16-byte slot'), the slot it FETCHES is slot' ^ (bank row & 15)
  uint32_t a_src[4], b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int bank_row = 16 * wave + 4 * q + (lane >> 4);
    const int s = (lane & 15) ^ (bank_row & 15);
    const int row = 2 * bank_row + (s >> 3);
    a_src[q] = ((uint32_t)min(co0 + row, p.CO - 1) * (uint32_t)p.HW + 4u * (s & 7)) * 4u;
    b_src[q] = ((uint32_t)min(ci0 + row, p.CI - 1) * (uint32_t)p.HW + 4u * (s & 7)) * 4u;
  }
  auto issue = [&](const int chunk, const int stage) {
    const int n = chunk / p.chunks_per_image;
    const int pix = (chunk - n * p.chunks_per_image) * 32;
    const char* ga = reinterpret_cast<const char*>(p.gy + (int64_t)n * p.gy_bs + pix);
    const char* xb = reinterpret_cast<const char*>(p.x + (int64_t)n * p.x_bs + pix);
    const uint32_t slot = lds0 + (uint32_t)stage * PWL_STAGE_BYTES + 4096u * wave;
#pragma unroll
    for (int q = 0; q < 4; ++q) ring_glds16(ga, a_src[q], slot + 1024u * q);
    if (wave < NI) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ring_glds16(xb, b_src[q], slot + PWL_A_BYTES + 1024u * q);
    }
  };

  float pro_a[NI], pro_b[NI];               // PRO: batch-norm + ReLU of this lane's input-channel rows
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    pro_a[ni] = 1.f; pro_b[ni] = 0.f;
    if (PRO) {
      const int c = min(ci0 + ni * 32 + l31, p.CI - 1);
      bn_coefficients(p.bn_mean[c], p.bn_inv[c], p.bn_gamma[c], p.bn_beta[c], pro_a[ni], pro_b[ni]);
    }
  }

  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  // fragment reads: row block b (gy: the wave's, x: ni) starts 4096 * b into its operand; lane (l31, lhi) reads the float4s
  // 4 * lhi + q of row l31 = slots 8 * (l31 & 1) + 4 * lhi + q of bank row l31 >> 1, swizzled by that bank row's low bits
  // (16 * b + (l31 >> 1)) & 15 = l31 >> 1.  MFMA step (q, e) pairs pixel 4q + e (lanes 0-31) with pixel 16 + 4q + e.
  const bool active = co0 + 32 * wave < p.CO;
  const uint32_t lane_read = (uint32_t)((l31 >> 1) * 256 + (((8 * (l31 & 1) + 4 * lhi) ^ (l31 >> 1)) << 4));
  // The fragments of float4 q + 1 are read into a second register set while the 4 * NI matrix instructions of float4 q issue.
  struct Fragments { f32x4 a; f32x4 b[NI]; };
  auto load = [&](Fragments& f, const char* slot, const int q) {
    const uint32_t at = lane_read ^ (uint32_t)(q << 4);
    f.a = *reinterpret_cast<const f32x4*>(slot + 4096 * wave + at);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) f.b[ni] = *reinterpret_cast<const f32x4*>(slot + PWL_A_BYTES + 4096 * ni + at);
  };
  auto multiply = [&](const Fragments& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float bv = f.b[ni][e];
        if (PRO) bv = fmaxf(fmaf(bv, pro_a[ni], pro_b[ni]), 0.f);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[e], bv, acc[ni], 0, 0, 0);
      }
    }
  };
  auto compute = [&](const int stage) {
    const char* slot = smem + stage * PWL_STAGE_BYTES;
    Fragments f0, f1;
    load(f0, slot, 0);
    load(f1, slot, 1);
    __builtin_amdgcn_sched_barrier(0);        // (the scheduler otherwise sinks every read to just in front of its first use)
    multiply(f0);
    __builtin_amdgcn_sched_barrier(0);
    load(f0, slot, 2);
    __builtin_amdgcn_sched_barrier(0);
    multiply(f1);
    __builtin_amdgcn_sched_barrier(0);
    load(f1, slot, 3);
    __builtin_amdgcn_sched_barrier(0);
    multiply(f0);
    multiply(f1);
  };

  // One barrier per chunk: at the top of chunk t its DMAs have landed (every wave waited for its own: at most the STAGES - 2
  // younger stages' instructions of THIS wave -- 4, or 8 when it also stages x rows -- are still in flight) and every wave is
  // done with chunk t - 1, whose slot the DMAs of chunk t + STAGES - 1 then refill while chunk t is in the matrix pipe.
  for (int s = 0; s < STAGES - 1 && s < count; ++s) issue(cbeg + s, s);
  for (int t0 = 0; t0 < count; t0 += STAGES) {
#pragma unroll
    for (int u = 0; u < STAGES; ++u) {
      const int t = t0 + u;
      if (t >= count) break;
      if (STAGES > 2 && count - 1 - t >= STAGES - 2) {
        if (wave < NI) ring_wait_and_barrier<8 * (STAGES - 2)>();
        else ring_wait_and_barrier<4 * (STAGES - 2)>();
      } else {
        ring_wait_and_barrier<0>();
      }
      if (t + STAGES - 1 < count) issue(cbeg + t + STAGES - 1, (u + STAGES - 1) % STAGES);
      if (active) compute(u);
    }
  }
  if (!active) return;

  // C/D fragment: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): half a wave writes 128 contiguous bytes
  const int row0 = 32 * wave + 4 * lhi;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = 32 * ni + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2);
      const float v = acc[ni][r];
      if (p.partial) {
        p.partial[((int64_t)block_x * p.split + block_y) * (PWL_TILE * PWL_TILE) + row * PWL_TILE + col] = v;
      } else if (co0 + row < p.CO && ci0 + col < p.CI) {
        float* dst = p.gw + (int64_t)(co0 + row) * p.CI + ci0 + col;
        if (p.mode == 2) unsafeAtomicAdd(dst, v);
        else *dst += v;
      }
    }
  }
}


template <bool PRO>
__device__ __forceinline__ void pointwise_wgrad_lds_tile(const PwWgradParams& p, const int block_x, const int block_y, char* smem) {
  const int tm = block_x / p.tiles_n, tn = block_x - tm * p.tiles_n;
  int ci0, blocks;                                                         // (workgroup-uniform)
  pointwise_wgrad_lds_columns(p.CI, p.tiles_n, tn, ci0, blocks);
  const int co0 = tm * PWL_TILE;
  if (blocks == 4) pointwise_wgrad_lds_body<PRO, 4>(p, block_x, block_y, co0, ci0, smem);
  else if (blocks == 3) pointwise_wgrad_lds_body<PRO, 3>(p, block_x, block_y, co0, ci0, smem);
  else if (blocks == 2) pointwise_wgrad_lds_body<PRO, 2>(p, block_x, block_y, co0, ci0, smem);
  else pointwise_wgrad_lds_body<PRO, 1>(p, block_x, block_y, co0, ci0, smem);
}

template <bool PRO>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_lds_kernel(const PwWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char pwl_smem[];
  pointwise_wgrad_lds_tile<PRO>(p, (int)blockIdx.x, (int)blockIdx.y, pwl_smem);
}

template <bool PRO, bool RAGGED>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_kernel(const PwWgradParams p) {
  __shared__ float red[2 * PWG_MI * 32 * (PWG_NI * 32 + 1)];
  pointwise_wgrad_body<PRO, RAGGED>(p, (int)blockIdx.x, (int)blockIdx.y, red);
}

// GROUPED: blockIdx.z selects one of many independent problems (all the bottleneck convolutions of a dense block's
// backward) from a device-resident table; x / gy are offsets from two base pointers, so the table does not change
// between steps.  Workgroups beyond a problem's own grid leave at once.
struct PwWgradJob {
  int64_t x_off, gy_off, x_bs, gy_bs;
  float* gw;
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;
  int32_t N, CI, CO, HW, tiles_n, tiles, chunks, chunks_per_worker, chunks_per_image, mode, split;
  int32_t lds;                   // 1: planned for the staged form (128 x 128 tiles, one K slice per WORKGROUP)
  int64_t partial_off;           // ordered form: this job's partial tiles start here (floats) in the launch's workspace region
};
static_assert(sizeof(PwWgradJob) == 128, "one 128-byte table slot per job");

// (gw_base != NULL: the job's gw field holds an element OFFSET into that per-step buffer instead of a pointer.)
//
// XCD-aware order.  The tiles of one SLICE (one problem, one K range) read the same gy rows (every tile) and the same x
// rows (the tiles_m tiles of a column): dispatched round-robin over the eight XCDs -- workgroup b runs on XCD b % 8,
// MI355X_MICROARCH.md -- each XCD's L2 fetched those rows separately: 129.5 GB of fabric traffic per step for 42.1 GB of
// operands (3.1x, profiles/r02z_pmc_per_kernel.md), 5.3 TB/s, i.e. the kernel sat on the HBM roofline of its own
// re-reads.  Now the grid is one-dimensional and a hardware workgroup id b means: XCD b % 8 works through the slices
// s = 8 * j + (b % 8), j = 0, 1, ..., and within a slice through its tiles, so that all the tiles of a slice are resident
// on ONE XCD at the same time and the rows come from HBM once (slices go to the XCDs round-robin: a problem's slices --
// and the problems' different widths -- spread evenly).
// (workgroup id -> (job, tile, K slice) and the job's parameters; false: a padding workgroup)
__device__ __forceinline__ bool pointwise_wgrad_grouped_slot(const PwWgradJob* __restrict__ jobs, const float* x_base, const float* gy_base,
                                                             float* gw_base, const int grid_x, const int grid_y, const int count,
                                                             float* partial_base, const int staged, PwWgradParams& p, int& tile,
                                                             int& y) {
  const int xcd = (int)blockIdx.x & 7, within_xcd = (int)blockIdx.x >> 3;
  const int round = within_xcd / grid_x;
  tile = within_xcd % grid_x;
  const int slice = round * 8 + ((xcd - round) & 7);        // (rotated per round: a problem's first slices visit every XCD)
  const int z = slice / grid_y;
  y = slice - z * grid_y;
  if (z >= count) return false;                                                    // (the padding of the last round)
  const PwWgradJob job = jobs[z];
  if (job.lds != staged) return false;                                             // a slot planned for the other kernel of the pair
  if (tile >= job.tiles || y >= job.split) return false;                           // (workgroup-uniform)
  p.x = x_base + job.x_off; p.gy = gy_base + job.gy_off;
  p.gw = gw_base ? gw_base + (int64_t)(intptr_t)job.gw : job.gw;
  p.x_bs = job.x_bs; p.gy_bs = job.gy_bs;
  p.N = job.N; p.CI = job.CI; p.CO = job.CO; p.HW = job.HW;
  p.tiles_n = job.tiles_n; p.chunks = job.chunks; p.chunks_per_worker = job.chunks_per_worker;
  p.chunks_per_image = job.chunks_per_image; p.mode = job.mode;
  p.bn_mean = job.bn_mean; p.bn_inv = job.bn_inv; p.bn_gamma = job.bn_gamma; p.bn_beta = job.bn_beta;
  p.partial = partial_base ? partial_base + job.partial_off : nullptr; p.split = job.split;
  return true;
}

template <bool PRO, bool RAGGED>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_grouped_kernel(const PwWgradJob* __restrict__ jobs,
                                                                         const float* x_base, const float* gy_base,
                                                                         float* gw_base, const int grid_x, const int grid_y,
                                                                         const int count, float* partial_base) {
  __shared__ float red[2 * PWG_MI * 32 * (PWG_NI * 32 + 1)];
  PwWgradParams p;
  int tile, y;
  if (!pointwise_wgrad_grouped_slot(jobs, x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base, 0, p, tile, y)) return;
  pointwise_wgrad_body<PRO, RAGGED>(p, tile, y, red);
}

template <bool PRO>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_lds_grouped_kernel(const PwWgradJob* __restrict__ jobs,
                                                                             const float* x_base, const float* gy_base,
                                                                             float* gw_base, const int grid_x, const int grid_y,
                                                                             const int count, float* partial_base) {
  extern __shared__ __attribute__((aligned(16))) char pwl_smem[];
  PwWgradParams p;
  int tile, y;
  if (!pointwise_wgrad_grouped_slot(jobs, x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base, 1, p, tile, y)) return;
  pointwise_wgrad_lds_tile<PRO>(p, tile, y, pwl_smem);
}

// blockIdx.x = 256-element slab of a tile, blockIdx.y = tile, blockIdx.z = job: the second stage of a grouped launch.
template <int TILE>
__global__ __launch_bounds__(256) void pointwise_wgrad_grouped_finish_kernel(const PwWgradJob* __restrict__ jobs, float* gw_base,
                                                                             const float* __restrict__ partial_base) {
  const PwWgradJob job = jobs[blockIdx.z];
  if ((int)blockIdx.y >= job.tiles || job.lds != (TILE == PWL_TILE ? 1 : 0)) return;
  float* gw = gw_base ? gw_base + (int64_t)(intptr_t)job.gw : job.gw;
  pointwise_wgrad_finish_tile<TILE, TILE, TILE == PWL_TILE>(partial_base + job.partial_off, gw, (int)blockIdx.y, (int)blockIdx.x, job.split, job.tiles_n,
                                          job.CO, job.CI);
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

// The un-fused weight gradient stays on the generic gather-GEMM by default (measured equal: both are bound by the
// operand stream at 64 x 64 tiles); this kernel serves the fused batch-norm form, where the generic one cannot.
bool pointwise_wgrad_enabled() {
  static const bool enabled = getenv("SRGAN_PW_WGRAD") != nullptr;
  return enabled;
}

// Grid plan of one problem: tiles, K split over wave workers, chunks per worker.
// `group`: the number of problems launched together (their workgroups share the GPU, so each needs fewer of its own).
static int pointwise_wgrad_plan(int32_t N, int32_t CI, int32_t CO, int32_t HW, PwWgradParams& p, int& tiles, int& split,
                                int group = 1, int64_t group_weights = 0) {
  const int tiles_m = (CO + PWG_MI * 32 - 1) / (PWG_MI * 32);
  p.tiles_n = (CI + PWG_NI * 32 - 1) / (PWG_NI * 32);
  tiles = tiles_m * p.tiles_n;
  SRGAN_REQUIRE(HW >= 4, SRGAN_EUNSUPPORTED, "pointwise wgrad plane of fewer than 4 pixels");
  p.chunks_per_image = (HW + 31) / 32;
  const int64_t chunks = (int64_t)N * p.chunks_per_image;
  SRGAN_REQUIRE(chunks < ((int64_t)1 << 30) && tiles < (1 << 30), SRGAN_ERANGE, "pointwise wgrad grid");
  p.chunks = (int)chunks;
  // Three resident workgroups per CU (166 registers per lane): 768 workgroups = 3072 wave workers over the whole grid, but at least `min_chunks` chunks per worker so the LDS reduction + atomic pass is amortised.
  static const int resident = getenv("SRGAN_PWG_WGS") ? atoi(getenv("SRGAN_PWG_WGS")) : 768;
  static const int min_chunks = getenv("SRGAN_PWG_DEPTH") ? atoi(getenv("SRGAN_PWG_DEPTH")) : 2;   // (8 measured equal at 512 x 512, 2.5 % slower at 224 x 224)
  static const int oversubscription = getenv("SRGAN_GROUP_OVERSUB") ? atoi(getenv("SRGAN_GROUP_OVERSUB")) : 4;
  int wanted = group > 1 ? (resident * oversubscription + group - 1) / group : resident;
  // (shared out by work, see pointwise_wgrad_lds_plan: 8.0 -> 6.6 ms per step on the ragged planes of 224 x 224)
  static const bool equal_shares = getenv("SRGAN_PWG_EQUAL_SHARES") != nullptr;
  if (!equal_shares && group > 1 && group_weights > 0)
    wanted = (int)(((int64_t)resident * oversubscription * ((int64_t)CO * CI) + group_weights - 1) / group_weights);
  split = (wanted + tiles - 1) / tiles;
  const int max_split = (int)((chunks + 4 * min_chunks - 1) / (4 * min_chunks));
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  p.chunks_per_worker = (int)((chunks + 4 * split - 1) / (4 * split));
  split = (int)((chunks + 4 * (int64_t)p.chunks_per_worker - 1) / (4 * (int64_t)p.chunks_per_worker));
  SRGAN_REQUIRE(split <= 65535, SRGAN_ERANGE, "pointwise wgrad split");
  p.mode = split > 1 ? 2 : 1;
  return SRGAN_OK;
}

// The staged form's plan: 128 x 128 tiles, the K split over WORKGROUPS (a workgroup walks `chunks_per_worker` chunks).
static int pointwise_wgrad_lds_plan(int32_t N, int32_t CI, int32_t CO, int32_t HW, PwWgradParams& p, int& tiles, int& split,
                                    int group = 1, int64_t group_weights = 0) {
  const int tiles_m = (CO + PWL_TILE - 1) / PWL_TILE;
  p.tiles_n = (CI + PWL_TILE - 1) / PWL_TILE;
  tiles = tiles_m * p.tiles_n;
  p.chunks_per_image = HW / 32;
  const int64_t chunks = (int64_t)N * p.chunks_per_image;
  SRGAN_REQUIRE(chunks < ((int64_t)1 << 30) && tiles < (1 << 30), SRGAN_ERANGE, "pointwise wgrad grid");
  p.chunks = (int)chunks;
  // Two resident workgroups per CU (72 KB of LDS each).  A worker leaves a 64 KB partial tile behind (written once, read once
  // by the finish): at least `min_chunks` chunks (32 KB of operands each) per worker.
  static const int resident = getenv("SRGAN_PWL_WGS") ? atoi(getenv("SRGAN_PWL_WGS")) : 512;
  static const int min_chunks = getenv("SRGAN_PWL_DEPTH") ? atoi(getenv("SRGAN_PWL_DEPTH")) : 8;
  static const int oversubscription = getenv("SRGAN_PWL_OVERSUB") ? atoi(getenv("SRGAN_PWL_OVERSUB")) : 4;
  // a grouped launch's workgroups are shared out by WORK (the problem's weights over the group's: the same K range per
  // worker whatever the problem's width) when the caller says what the group holds, else in equal parts
  int wanted = group > 1 ? (resident * oversubscription + group - 1) / group : resident;
  static const bool equal_shares = getenv("SRGAN_PWL_EQUAL_SHARES") != nullptr;
  if (group > 1 && group_weights > 0 && !equal_shares)
    wanted = (int)(((int64_t)resident * oversubscription * ((int64_t)CO * CI) + group_weights - 1) / group_weights);
  split = (wanted + tiles - 1) / tiles;
  const int max_split = (int)((chunks + min_chunks - 1) / min_chunks);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  p.chunks_per_worker = (int)((chunks + split - 1) / split);
  split = (int)((chunks + p.chunks_per_worker - 1) / p.chunks_per_worker);
  SRGAN_REQUIRE(split <= 65535, SRGAN_ERANGE, "pointwise wgrad split");
  p.mode = split > 1 ? 2 : 1;
  return SRGAN_OK;
}

// Whole 32-pixel chunks, 16-byte aligned rows (the caller checks the pointers / offsets), byte offsets of a row inside an
// image in 32 bits: the staged form; SRGAN_NO_PW_WGRAD_LDS=1 keeps the register-streamed kernel.
static bool pointwise_wgrad_lds_shape(int32_t CI, int32_t CO, int32_t HW) {
  static const bool disabled = getenv("SRGAN_NO_PW_WGRAD_LDS") != nullptr;
  static const int min_ci = getenv("SRGAN_PWL_MIN_CI") ? atoi(getenv("SRGAN_PWL_MIN_CI")) : 0;
  return !disabled && CI >= min_ci && HW % 32 == 0 && (int64_t)(CI > CO ? CI : CO) * HW * 4 < ((int64_t)1 << 32);
}

// (72 KB of dynamic LDS is beyond the 64 KB a launch gets without the attribute: set once per kernel and device, see pointwise_ring.hip)
template <typename Kernel>
static int pointwise_wgrad_lds_configure(Kernel kernel, std::atomic<uint64_t>& configured_devices) {
  int device = 0;
  SRGAN_HIP(hipGetDevice(&device));
  const uint64_t bit = (uint64_t)1 << (device & 63);
  if (!(configured_devices.load(std::memory_order_acquire) & bit)) {
    SRGAN_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PWL_LDS_BYTES));
    configured_devices.fetch_or(bit, std::memory_order_release);
  }
  return SRGAN_OK;
}

static bool pointwise_wgrad_ragged(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, int32_t HW) {
  return HW % 32 != 0 || x_bs % 4 != 0 || gy_bs % 4 != 0 || (((uintptr_t)x | (uintptr_t)gy) & 15) != 0;
}

// gw (=,+=) the weight gradient of a 1x1 convolution; x / gy may be channel-slice views (batch strides in
// elements).  HW % 32 == 0 with 16-byte aligned rows takes the plain kernel, anything else with HW >= 4 the RAGGED one.
int pointwise_wgrad_run(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw, int32_t N, int32_t CI,
                        int32_t CO, int32_t HW, int accumulate, hipStream_t stream, const float* const* bn) {
  PwWgradParams p;
  p.bn_mean = bn ? bn[0] : nullptr; p.bn_inv = bn ? bn[1] : nullptr;
  p.bn_gamma = bn ? bn[2] : nullptr; p.bn_beta = bn ? bn[3] : nullptr;
  p.x = x; p.gy = gy; p.gw = gw; p.x_bs = x_bs; p.gy_bs = gy_bs;
  p.N = N; p.CI = CI; p.CO = CO; p.HW = HW;
  int tiles = 0, split = 1;
  const bool ragged = pointwise_wgrad_ragged(x, x_bs, gy, gy_bs, HW);
  const bool staged = !ragged && pointwise_wgrad_lds_shape(CI, CO, HW);
  if (const int status = staged ? pointwise_wgrad_lds_plan(N, CI, CO, HW, p, tiles, split) : pointwise_wgrad_plan(N, CI, CO, HW, p, tiles, split))
    return status;
  if (!accumulate) if (const int status = zero_floats(gw, (int64_t)CO * CI, stream)) return status;
  // K split: every slice's partial tile through the workspace, added in slice order by a second kernel (no atomics)
  const int tile_floats = staged ? PWL_TILE * PWL_TILE : PWG_MI * 32 * PWG_NI * 32;
  p.partial = nullptr; p.split = split;
  if (split > 1 && !split_atomics_forced()) p.partial = partial_workspace((size_t)tiles * split * tile_floats * sizeof(float), stream);
  dim3 grid((unsigned)tiles, (unsigned)split, 1);
  if (staged) {
    static std::atomic<uint64_t> configured_plain{0}, configured_bn{0};
    if (const int status = bn ? pointwise_wgrad_lds_configure(pointwise_wgrad_lds_kernel<true>, configured_bn)
                              : pointwise_wgrad_lds_configure(pointwise_wgrad_lds_kernel<false>, configured_plain)) return status;
  }
  const int profile_slot = profile_bracket_begin(stream);
  if (staged && bn) hipLaunchKernelGGL(pointwise_wgrad_lds_kernel<true>, grid, dim3(256), PWL_LDS_BYTES, stream, p);
  else if (staged) hipLaunchKernelGGL(pointwise_wgrad_lds_kernel<false>, grid, dim3(256), PWL_LDS_BYTES, stream, p);
  else if (ragged && bn) hipLaunchKernelGGL((pointwise_wgrad_kernel<true, true>), grid, dim3(256), 0, stream, p);
  else if (ragged) hipLaunchKernelGGL((pointwise_wgrad_kernel<false, true>), grid, dim3(256), 0, stream, p);
  else if (bn) hipLaunchKernelGGL((pointwise_wgrad_kernel<true, false>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((pointwise_wgrad_kernel<false, false>), grid, dim3(256), 0, stream, p);
  if (p.partial && staged)
    hipLaunchKernelGGL(pointwise_wgrad_finish_kernel<PWL_TILE>, dim3(PWL_FINISH_SLABS, (unsigned)tiles), dim3(256), 0, stream, p);
  else if (p.partial)
    hipLaunchKernelGGL(pointwise_wgrad_finish_kernel<PWG_MI * 32>, dim3(PWG_FINISH_SLABS, (unsigned)tiles), dim3(256), 0, stream, p);
  const int status = launch_status();
  profile_bracket_end(profile_slot, stream, CO, CI, (int64_t)N * HW, 6, staged ? PWL_TILE : PWG_MI * 32, staged ? PWL_TILE : PWG_NI * 32, split);
  return status;
}

// One entry of a grouped launch's table (host side; the caller uploads the table once).  x / gy are element offsets from
// the two base pointers given at launch time; the weight gradient is ACCUMULATED into gw.
int pointwise_wgrad_group_plan(int64_t x_off, int64_t x_bs, int64_t gy_off, int64_t gy_bs, float* gw, int64_t gw_off, int32_t N, int32_t CI,
                               int32_t CO, int32_t HW, const float* const* bn, int32_t group, int64_t group_weights, int64_t partial_offset,
                               void* job_out, int32_t* grid_x, int32_t* grid_y, int32_t* ragged, int64_t* partial_floats) {
  PwWgradParams p;
  int tiles = 0, split = 1;
  // offsets that are not multiples of 4 floats, or strides / planes that are not, need the ragged variant
  const bool rag = HW % 32 != 0 || x_bs % 4 != 0 || gy_bs % 4 != 0 || x_off % 4 != 0 || gy_off % 4 != 0;
  const bool staged = !rag && pointwise_wgrad_lds_shape(CI, CO, HW);
  if (const int status = staged ? pointwise_wgrad_lds_plan(N, CI, CO, HW, p, tiles, split, group, group_weights)
                                : pointwise_wgrad_plan(N, CI, CO, HW, p, tiles, split, group, group_weights)) return status;
  PwWgradJob job;
  job.x_off = x_off; job.gy_off = gy_off; job.x_bs = x_bs; job.gy_bs = gy_bs;
  job.gw = gw ? gw : reinterpret_cast<float*>((intptr_t)gw_off);
  job.bn_mean = bn ? bn[0] : nullptr; job.bn_inv = bn ? bn[1] : nullptr;
  job.bn_gamma = bn ? bn[2] : nullptr; job.bn_beta = bn ? bn[3] : nullptr;
  job.N = N; job.CI = CI; job.CO = CO; job.HW = HW; job.tiles_n = p.tiles_n; job.tiles = tiles; job.chunks = p.chunks;
  job.chunks_per_worker = p.chunks_per_worker; job.chunks_per_image = p.chunks_per_image; job.mode = p.mode;
  job.split = split; job.lds = staged ? 1 : 0;
  job.partial_off = partial_offset;
  *partial_floats = (int64_t)tiles * split * (staged ? PWL_TILE * PWL_TILE : PWG_MI * 32 * PWG_NI * 32);
  static_assert(sizeof(PwWgradJob) <= 128, "job slot");
  memset(job_out, 0, 128);
  memcpy(job_out, &job, sizeof(job));
  *grid_x = tiles; *grid_y = split;
  // the kernel this slot was planned for, as a bit: 1 the staged form, 2 the register-streamed kernel, 4 its ragged variant.  A
  // launch gets the OR over its group: slots planned for the staged form (other tiles, other partial layout) and the others
  // are served by one launch each, and every kernel skips the slots of the other; the ragged variant serves 2 and 4.
  *ragged = staged ? 1 : (rag ? 4 : 2);
  return SRGAN_OK;
}

int pointwise_wgrad_group_run(const void* jobs, int32_t count, int32_t grid_x, int32_t grid_y, int32_t ragged, int32_t fused_bn,
                              const float* x_base, const float* gy_base, float* gw_base, int64_t flops_mn, int64_t pixels,
                              int64_t elements, int64_t partial_floats, hipStream_t stream) {
  SRGAN_REQUIRE(count >= 1 && count <= 65535 && grid_y <= 65535, SRGAN_ERANGE, "grouped pointwise wgrad grid");
  // the ordered form: every job's K-slice partials in the stream's workspace (the table holds each job's offset), a second
  // launch adds them in slice order -- when the workspace holds them; otherwise fp32 atomics into gw
  float* partial_base = nullptr;
  if (partial_floats > 0 && grid_y > 1 && !split_atomics_forced())
    partial_base = partial_workspace((size_t)partial_floats * sizeof(float), stream);
  const bool misaligned = (((uintptr_t)x_base | (uintptr_t)gy_base) & 15) != 0;
  const bool staged = (ragged & 1) != 0, streamed = (ragged & 6) != 0 || ragged == 0, rag = (ragged & 4) != 0 || misaligned;
  SRGAN_REQUIRE(!staged || !misaligned, SRGAN_EINVAL, "grouped pointwise wgrad: the staged form needs 16-byte aligned bases");
  // one-dimensional: 8 XCDs x (slices per XCD, rounded up) x tiles (see the kernel)
  const int64_t slices = (int64_t)grid_y * count, rounds = (slices + 7) / 8;
  SRGAN_REQUIRE(rounds * 8 * grid_x < ((int64_t)1 << 31), SRGAN_ERANGE, "grouped pointwise wgrad grid");
  dim3 grid((unsigned)(rounds * 8 * grid_x), 1, 1);
  const PwWgradJob* table = reinterpret_cast<const PwWgradJob*>(jobs);
  if (staged) {
    static std::atomic<uint64_t> configured_plain{0}, configured_bn{0};
    if (const int status = fused_bn ? pointwise_wgrad_lds_configure(pointwise_wgrad_lds_grouped_kernel<true>, configured_bn)
                                    : pointwise_wgrad_lds_configure(pointwise_wgrad_lds_grouped_kernel<false>, configured_plain))
      return status;
  }
  const int profile_slot = profile_bracket_begin(stream);
#define SRGAN_PWG_LAUNCH(PRO, RAG) hipLaunchKernelGGL((pointwise_wgrad_grouped_kernel<PRO, RAG>), grid, dim3(256), 0, stream, \
                                                      table, x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base)
#define SRGAN_PWL_LAUNCH(PRO) hipLaunchKernelGGL(pointwise_wgrad_lds_grouped_kernel<PRO>, grid, dim3(256), PWL_LDS_BYTES, stream, \
                                                 table, x_base, gy_base, gw_base, grid_x, grid_y, count, partial_base)
  if (staged && fused_bn) SRGAN_PWL_LAUNCH(true);
  else if (staged) SRGAN_PWL_LAUNCH(false);
  if (streamed) {
    if (fused_bn && rag) SRGAN_PWG_LAUNCH(true, true);
    else if (fused_bn) SRGAN_PWG_LAUNCH(true, false);
    else if (rag) SRGAN_PWG_LAUNCH(false, true);
    else SRGAN_PWG_LAUNCH(false, false);
  }
#undef SRGAN_PWG_LAUNCH
#undef SRGAN_PWL_LAUNCH
  if (partial_base && staged)
    hipLaunchKernelGGL(pointwise_wgrad_grouped_finish_kernel<PWL_TILE>, dim3(PWL_FINISH_SLABS, (unsigned)grid_x, (unsigned)count), dim3(256), 0,
                       stream, table, gw_base, partial_base);
  if (partial_base && streamed)
    hipLaunchKernelGGL(pointwise_wgrad_grouped_finish_kernel<PWG_MI * 32>, dim3(PWG_FINISH_SLABS, (unsigned)grid_x, (unsigned)count), dim3(256), 0,
                       stream, table, gw_base, partial_base);
  const int status = launch_status();
  // logical shape of the group: M x (sum of the input widths) x pixels, i.e. flops_mn = sum CO * CI
  profile_bracket_end(profile_slot, stream, 1, flops_mn, pixels, 6, staged ? PWL_TILE : PWG_MI * 32, staged ? PWL_TILE : PWG_NI * 32, grid_y, 0, 0,
                      elements > pixels ? elements - pixels : 0);
  return status;
}

}  // namespace srgan
