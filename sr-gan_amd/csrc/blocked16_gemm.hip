// blocked16_gemm.hip -- the linear layers of the 16-bit data path (VGG-16's classifier, reference age/vgg.py:33-41,48-53; the
// full-plane convolutions of the DCGAN discriminator / generator, age/models.py:37,65, which are linear maps over a flattened
// plane).  A [N, F] activation matrix is the H = W = 1 case of the blocked layout (blocked16.h): row-major 16-bit.
//
//   hgemm_kernel          out[n][m] = epi(sum_k A[m][k] * B[n][k] [+ bias[m]])     both operands k-contiguous: fragments are
//                                                                                 ds_read_b128 from an XOR-swizzled LDS image
//   hlinear_wgrad_kernel  gw[m][k] += sum_n S[n][m] * X[n][k]                      both operands reduced over ROWS: fragments by
//                                                                                 ds_read_b64_tr_b16 (blocked16.h)
//   h_pack_matrix_kernel  the 16-bit shadow of a weight matrix, optionally transposed and with the row / column order of a
//                         flattened blocked plane ([c][p] -> [c / 8][p][c % 8])
#include <type_traits>
#include "blocked16.h"
#include "split_finish.h"

namespace srgan {

// index of element i' of a flattened BLOCKED plane tensor ([C / 8][P][8]) in the flattened NCHW order ([C][P])
__host__ __device__ __forceinline__ int64_t h_plane_index(int64_t i, int32_t plane) {
  if (plane <= 1) return i;
  const int64_t slot = i >> 3;
  return ((slot / plane) * 8 + (i & 7)) * plane + slot % plane;
}

// out16[r][c] (pitch ld elements, ld % 8 == 0) = src[map(r, row_plane) * rs + map(c, col_plane) * cs] for map(r) < rows_real,
// map(c) < cols_real, else 0.  One thread per slot of 8 consecutive c.
template <int PREC>
__global__ __launch_bounds__(256) void h_pack_matrix_kernel(const float* __restrict__ src, Slot* __restrict__ out, int64_t slots,
                                                            int32_t row_slots, int64_t rows_real, int64_t cols_real, int64_t rs,
                                                            int64_t cs, int32_t row_plane, int32_t col_plane) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= slots) return;
  const int64_t r = slot / row_slots;
  const int64_t c0 = (slot % row_slots) * 8;
  const int64_t rr = h_plane_index(r, row_plane);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t cc = h_plane_index(c0 + j, col_plane);
    v[j] = (rr < rows_real && cc < cols_real) ? src[rr * rs + cc * cs] : 0.f;
  }
  out[slot] = h_pack8<PREC>(v);
}

// The same for row_stride == 1 (the TRANSPOSED shadow of a row-major matrix: consecutive output rows are consecutive source
// addresses): a 64 x 64 tile goes through LDS, so both the fp32 reads (along r) and the 16-byte slot stores (along c) are
// coalesced -- the one-thread-per-slot kernel read eight floats a row pitch apart and fetched every line seven times
// (4.1 GB per step for VGG-16's two large linear layers, PMC profiles/r06f).
template <int PREC>
__global__ __launch_bounds__(256) void h_pack_matrix_transposed_kernel(const float* __restrict__ src, Slot* __restrict__ out,
                                                                       int64_t rows, int32_t row_slots, int64_t rows_real,
                                                                       int64_t cols_real, int64_t cs, int32_t row_plane,
                                                                       int32_t col_plane, int32_t tiles_r) {
  __shared__ float tile[64][65];
  const int tid = (int)threadIdx.x;
  const int64_t r0 = (int64_t)((int)blockIdx.x % tiles_r) * 64, c0 = (int64_t)((int)blockIdx.x / tiles_r) * 64;
  const int r_load = tid & 63;
  const int64_t rr = h_plane_index(r0 + r_load, row_plane);
  const bool row_ok = r0 + r_load < rows && rr < rows_real;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c_local = (tid >> 6) + 4 * i;
    const int64_t cc = h_plane_index(c0 + c_local, col_plane);
    tile[c_local][r_load] = (row_ok && c0 + c_local < (int64_t)row_slots * 8 && cc < cols_real) ? src[cc * cs + rr] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot_local = tid & 7, r_local = (tid >> 3) + 32 * i;
    if (r0 + r_local >= rows || c0 / 8 + slot_local >= row_slots) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[slot_local * 8 + j][r_local];
    out[(r0 + r_local) * row_slots + c0 / 8 + slot_local] = h_pack8<PREC>(v);
  }
}

struct HGemmParams {
  const Slot* a; const Slot* b; Slot* out; const float* bias; const Slot* ref;
  float slope; int32_t epi;
  int32_t M, N, KS;              // rows of A that exist, rows of B, k slots (K / 8)
  int32_t lda, ldb, ldo;         // row pitches in SLOTS
  int32_t M_real;                // bias entries
  int32_t steps, steps_per_split;
  int32_t tiles_m;
  float* split_ws; unsigned int* split_tickets;
};

__device__ unsigned int g_hgemm_split_tickets[SPLIT_TICKET_SETS * SPLIT_TICKET_TILES];

// 128 x 128 output tile, K in steps of 64 (8 slots per row); 4 waves as 2 x 2, each 64 x 64 (MI = NI = 2).  LDS image per
// operand: [128 rows][8 slots], slot (row, kq) at row * 8 + (kq ^ ((row >> 1) & 7)): the 16 lanes of a ds_read_b128 group (16
// consecutive rows, one kq) hit 16 different slots of the 256-byte bank row, and a thread's staging store lands conflict-free
// too.  One LDS stage; the next step's 16-byte loads are in flight during the current step's 16 MFMAs.
template <int PREC>
__global__ __launch_bounds__(256, 2) void hgemm_kernel(const HGemmParams p) {
  __shared__ Slot lds[2 * 1024];
  Slot* as = lds;
  Slot* bs = lds + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = (int)blockIdx.x % p.tiles_m, tn = (int)blockIdx.x / p.tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  // the tile's 128 bias values in LDS, staged here and read in the epilogue (as in the convolution kernels)
  __shared__ __attribute__((aligned(16))) float bias_rows[128];
  const bool with_bias = p.epi == 1 && p.bias != nullptr;
  if (with_bias) {
    if ((int)threadIdx.x < 128) bias_rows[threadIdx.x] = m0 + (int)threadIdx.x < p.M_real ? p.bias[m0 + threadIdx.x] : 0.f;
    __syncthreads();
  }
  const int sbeg = (int)blockIdx.y * p.steps_per_split;
  const int send = min(p.steps, sbeg + p.steps_per_split);

  Slot ra[4], rb[4];
  auto fetch = [&](int step) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int flat = e * 256 + tid;
      const int row = flat >> 3, kq = step * 8 + (flat & 7);
      const bool oka = m0 + row < p.M && kq < p.KS, okb = n0 + row < p.N && kq < p.KS;
      ra[e] = p.a[oka ? (int64_t)(m0 + row) * p.lda + kq : 0];
      rb[e] = p.b[okb ? (int64_t)(n0 + row) * p.ldb + kq : 0];
    }
  };
  auto stage = [&](int step) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int flat = e * 256 + tid;
      const int row = flat >> 3, kq = flat & 7, kqg = step * 8 + kq;
      const int at = row * 8 + (kq ^ ((row >> 1) & 7));
      Slot va = ra[e], vb = rb[e];
      if (!(m0 + row < p.M && kqg < p.KS)) va = Slot{{0u, 0u, 0u, 0u}};
      if (!(n0 + row < p.N && kqg < p.KS)) vb = Slot{{0u, 0u, 0u, 0u}};
      as[at] = va;
      bs[at] = vb;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  if (sbeg < send) {
    fetch(sbeg);
    for (int step = sbeg; step < send; ++step) {
      __syncthreads();                       // the previous step's fragment reads are done
      stage(step);
      __syncthreads();
      if (step + 1 < send) fetch(step + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Slot a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + l31;
          a[mi] = as[row * 8 + ((2 * ks + lhi) ^ ((row >> 1) & 7))];
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int row = wn * 64 + ni * 32 + l31;
          b[ni] = bs[row * 8 + ((2 * ks + lhi) ^ ((row >> 1) & 7))];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = h_mfma<PREC>(a[mi], b[ni], acc[mi][ni]);
      }
    }
  }

  if (gridDim.y > 1) {
    if (!split_finish_ordered<64, 256>(p.split_ws + (int64_t)blockIdx.x * gridDim.y * (64 * 256), (int)blockIdx.y, (int)gridDim.y,
                                       p.split_tickets + blockIdx.x,
                                       [&](int i) { return acc[i / 32][(i / 16) % 2][i % 16]; },
                                       [&](int i, float v) { acc[i / 32][(i / 16) % 2][i % 16] = v; }))
      return;
  }

  // C/D fragment: column = n (lane & 31), registers 4q .. 4q + 3 = four consecutive m: 8 bytes of out[n][m ..]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    if (n >= p.N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int m = m0 + wm * 64 + mi * 32 + 8 * qd + 4 * lhi;
        if ((m >> 3) >= p.ldo) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * qd + j];
        uint2* dst = reinterpret_cast<uint2*>(p.out + (int64_t)n * p.ldo + (m >> 3)) + lhi;
        if (p.epi == 1) {
          if (with_bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(&bias_rows[wm * 64 + mi * 32 + 8 * qd + 4 * lhi]);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
        } else if (p.epi == 2) {
          const uint2 r = *(reinterpret_cast<const uint2*>(p.ref + (int64_t)n * p.ldo + (m >> 3)) + lhi);
          v[0] *= h_mask(r.x & 0xFFFFu, p.slope); v[1] *= h_mask(r.x >> 16, p.slope);
          v[2] *= h_mask(r.y & 0xFFFFu, p.slope); v[3] *= h_mask(r.y >> 16, p.slope);
        }
        uint2 packed;
        packed.x = h_pack2<PREC>(v[0], v[1]);
        packed.y = h_pack2<PREC>(v[2], v[3]);
        *dst = packed;
      }
    }
  }
}

struct HLinearWgradParams {
  const Slot* s; const Slot* x; float* gw;
  int32_t N, MS, KS;            // rows, slots per row of S (output side) and of X (input side)
  int32_t M_real, K_real;       // the weight matrix gw[M_real][K_real] (fp32, row pitch ldw)
  int64_t ldw_m, ldw_k;         // element strides of gw along m and k (a transposed-convolution weight is [K][M])
  int32_t row_plane, col_plane;
  int32_t tiles_m;
};

constexpr int HLW_NC = 64;      // rows (the reduction index) per staged chunk

// 128 (m) x 128 (k) block of gw per workgroup; 4 waves as 2 x 2, each 64 x 64.  Per chunk of 64 rows the S slots
// [16 m-groups][64 rows] and X slots [16 k-groups][64 rows] are staged (group stride 68 slots) and read with transpose reads:
// rows are what the MFMA reduces over.  One workgroup owns its block: no split, a fixed order.
template <int PREC>
__global__ __launch_bounds__(256, 2) void hlinear_wgrad_kernel(const HLinearWgradParams p) {
  constexpr int STR = 68;
  __shared__ Slot lds[2 * 16 * STR];
  Slot* ss = lds;
  Slot* xs = lds + 16 * STR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = (int)blockIdx.x % p.tiles_m, tk = (int)blockIdx.x / p.tiles_m;
  const int mg0 = tm * 16, kg0 = tk * 16;

  Slot rs_[4], rx[4];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int flat = e * 256 + tid;
      const int n = chunk * HLW_NC + (flat >> 4), grp = flat & 15;
      const bool oks = n < p.N && mg0 + grp < p.MS, okx = n < p.N && kg0 + grp < p.KS;
      rs_[e] = p.s[oks ? (int64_t)n * p.MS + mg0 + grp : 0];
      rx[e] = p.x[okx ? (int64_t)n * p.KS + kg0 + grp : 0];
    }
  };
  auto stage = [&](int chunk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int flat = e * 256 + tid;
      const int nl = flat >> 4, grp = flat & 15, n = chunk * HLW_NC + nl;
      Slot vs = rs_[e], vx = rx[e];
      if (!(n < p.N && mg0 + grp < p.MS)) vs = Slot{{0u, 0u, 0u, 0u}};
      if (!(n < p.N && kg0 + grp < p.KS)) vx = Slot{{0u, 0u, 0u, 0u}};
      ss[grp * STR + nl] = vs;
      xs[grp * STR + nl] = vx;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int G = lane >> 4, rb = G & 1, khalf = G >> 1, s16 = lane & 15, j = s16 >> 2, u = s16 & 3;
  uint32_t a_base[2], b_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_base[i] = h_lds_address(ss) + (uint32_t)(((4 * (2 * wm + i) + 2 * rb + (u >> 1)) * STR + 8 * khalf + j) * 16 + (u & 1) * 8);
    b_base[i] = h_lds_address(xs) + (uint32_t)(((4 * (2 * wn + i) + 2 * rb + (u >> 1)) * STR + 8 * khalf + j) * 16 + (u & 1) * 8);
  }

  const int chunks = (p.N + HLW_NC - 1) / HLW_NC;
  fetch(0);
  for (int chunk = 0; chunk < chunks; ++chunk) {
    __syncthreads();
    stage(chunk);
    __syncthreads();
    if (chunk + 1 < chunks) fetch(chunk + 1);
#pragma unroll
    for (int ks = 0; ks < HLW_NC / 16; ++ks) {
      Slot a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint2 a0 = h_tr_read(a_base[i] + ks * 256), a1 = h_tr_read(a_base[i] + ks * 256 + 64);
        a[i].v[0] = a0.x; a[i].v[1] = a0.y; a[i].v[2] = a1.x; a[i].v[3] = a1.y;
        const uint2 b0 = h_tr_read(b_base[i] + ks * 256), b1 = h_tr_read(b_base[i] + ks * 256 + 64);
        b[i].v[0] = b0.x; b[i].v[1] = b0.y; b[i].v[2] = b1.x; b[i].v[3] = b1.y;
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = h_mfma<PREC>(a[mi], b[ni], acc[mi][ni]);
    }
  }

  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t kk = h_plane_index((int64_t)kg0 * 8 + wn * 64 + ni * 32 + l31, p.col_plane);
    if (kk >= p.K_real) continue;
    // gw += acc as "load all, then add and store all": loads and stores return through one in-order counter (vmcnt), so the
    // element-wise `+=` made every load wait for the store in front of it -- 64 round trips per lane (4096 x 4096 x 128 rows:
    // 77 us for 134 MB)
    float old_values[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t mm = h_plane_index((int64_t)mg0 * 8 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, p.row_plane);
        old_values[mi][r] = mm < p.M_real ? p.gw[mm * p.ldw_m + kk * p.ldw_k] : 0.f;
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t mm = h_plane_index((int64_t)mg0 * 8 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, p.row_plane);
        if (mm < p.M_real) p.gw[mm * p.ldw_m + kk * p.ldw_k] = old_values[mi][r] + acc[mi][ni][r];
      }
  }
}

static int check_dtype_g(int dtype) {
  SRGAN_REQUIRE(dtype == 1 || dtype == 2, SRGAN_EINVAL, "blocked 16-bit tensors are bf16 (1) or fp16 (2)");
  return SRGAN_OK;
}

}  // namespace srgan

using namespace srgan;

extern "C" {

// out16[rows][cols] (row pitch = ceil(cols / 8) slots) = src[map(r, row_plane) * row_stride + map(c, col_plane) * col_stride],
// zero where the mapped index is outside [rows_real) x [cols_real).  map(i, P) = the NCHW-flattened index of element i of a
// flattened blocked plane tensor with P pixels (P <= 1: identity).
int srgan_h_pack_matrix(const float* src, void* out, int64_t rows, int64_t cols, int64_t rows_real, int64_t cols_real,
                        int64_t row_stride, int64_t col_stride, int32_t row_plane, int32_t col_plane, int dtype,
                        hipStream_t stream) {
  if (const int status = check_dtype_g(dtype)) return status;
  SRGAN_REQUIRE(src && out && rows > 0 && cols > 0, SRGAN_EINVAL, "srgan_h_pack_matrix arguments");
  const int64_t row_slots = (cols + 7) / 8, slots = rows * row_slots;
  SRGAN_REQUIRE(row_slots < ((int64_t)1 << 31), SRGAN_ERANGE, "srgan_h_pack_matrix row length");
  if (row_stride == 1 && col_stride != 1) {
    const int tiles_r = (int)((rows + 63) / 64), tiles_c = (int)((row_slots * 8 + 63) / 64);
    const dim3 tgrid((unsigned)(tiles_r * tiles_c));
    if (dtype == 1) hipLaunchKernelGGL(h_pack_matrix_transposed_kernel<1>, tgrid, dim3(256), 0, stream, src, (Slot*)out, rows, (int32_t)row_slots, rows_real, cols_real, col_stride, row_plane, col_plane, tiles_r);
    else hipLaunchKernelGGL(h_pack_matrix_transposed_kernel<2>, tgrid, dim3(256), 0, stream, src, (Slot*)out, rows, (int32_t)row_slots, rows_real, cols_real, col_stride, row_plane, col_plane, tiles_r);
    return launch_status();
  }
  const dim3 grid((unsigned)((slots + 255) / 256));
  if (dtype == 1) hipLaunchKernelGGL(h_pack_matrix_kernel<1>, grid, dim3(256), 0, stream, src, (Slot*)out, slots, (int32_t)row_slots, rows_real, cols_real, row_stride, col_stride, row_plane, col_plane);
  else hipLaunchKernelGGL(h_pack_matrix_kernel<2>, grid, dim3(256), 0, stream, src, (Slot*)out, slots, (int32_t)row_slots, rows_real, cols_real, row_stride, col_stride, row_plane, col_plane);
  return launch_status();
}

// out[N][M'] = epi(B[N][K] * A[M][K]^T [+ bias]) on 16-bit row-major matrices (pitches = ceil(. / 8) slots); M' = out_cols.
// epi as srgan_h_conv3x3.  The output's columns beyond M (inside its last slot) are written as zeros.
int srgan_h_gemm(const void* a, const void* b, const float* bias, const void* ref, float slope, int epi, void* out, int32_t M,
                 int32_t N, int32_t K, int32_t out_cols, int32_t bias_entries, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_g(dtype)) return status;
  SRGAN_REQUIRE(a && b && out && M > 0 && N > 0 && K > 0 && out_cols > 0 && epi >= 0 && epi <= 2 && (epi != 2 || ref), SRGAN_EINVAL,
                "srgan_h_gemm arguments");
  HGemmParams p;
  p.a = (const Slot*)a; p.b = (const Slot*)b; p.out = (Slot*)out; p.bias = bias; p.ref = (const Slot*)ref;
  p.slope = slope; p.epi = epi;
  p.M = M; p.N = N; p.KS = (K + 7) / 8;
  p.lda = p.KS; p.ldb = p.KS; p.ldo = (out_cols + 7) / 8;
  p.M_real = bias_entries;
  p.steps = (p.KS + 7) / 8;
  const int rows_out = p.ldo * 8;                             // rows of the tile grid: every stored column is written
  p.tiles_m = (rows_out + 127) / 128;
  const int tiles_n = (N + 127) / 128;
  const int tiles = p.tiles_m * tiles_n;
  int split = 1;
  if (tiles < 192 && p.steps >= 4 && tiles <= SPLIT_TICKET_TILES) {
    split = (384 + tiles - 1) / tiles;
    if (split > p.steps / 2) split = p.steps / 2;
    if (split > 16) split = 16;
  }
  p.steps_per_split = (p.steps + split - 1) / split;
  split = (p.steps + p.steps_per_split - 1) / p.steps_per_split;
  p.split_ws = nullptr; p.split_tickets = nullptr;
  if (split > 1) {
    int ticket_set = -1;
    float* ws = split_workspace(tiles, split, 64 * 256, 0, stream, &ticket_set);
    unsigned int* tickets = ws ? device_tickets(g_hgemm_split_tickets) : nullptr;
    if (ws && tickets) {
      p.split_ws = ws;
      p.split_tickets = tickets + (size_t)ticket_set * SPLIT_TICKET_TILES;
    } else {
      split = 1;
      p.steps_per_split = p.steps;
    }
  }
  const dim3 grid((unsigned)tiles, (unsigned)split);
  const int slot = profile_bracket_begin(stream);
  if (dtype == 1) hipLaunchKernelGGL(hgemm_kernel<1>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(hgemm_kernel<2>, grid, dim3(256), 0, stream, p);
  const int status = launch_status();
  profile_bracket_end_bytes(slot, stream, M, N, K, 16, 128, 128, split,
                            2.0 * ((double)M * K + (double)N * K + (double)N * out_cols * (epi == 2 ? 2 : 1)), dtype);
  return status;
}

// gw (fp32) element (m, k) at gw[map(m, row_plane) * ldw_m + map(k, col_plane) * ldw_k] += sum_n S[n][m] * X[n][k] for
// mapped indices inside [M_real) x [K_real); S is [N][M cols], X is [N][K cols] (16-bit, pitches ceil(. / 8) slots).
int srgan_h_linear_wgrad(const void* s, const void* x, float* gw, int32_t N, int32_t M, int32_t K, int64_t M_real, int64_t K_real,
                         int64_t ldw_m, int64_t ldw_k, int32_t row_plane, int32_t col_plane, int dtype, hipStream_t stream) {
  if (const int status = check_dtype_g(dtype)) return status;
  SRGAN_REQUIRE(s && x && gw && N > 0 && M > 0 && K > 0, SRGAN_EINVAL, "srgan_h_linear_wgrad arguments");
  HLinearWgradParams p;
  p.s = (const Slot*)s; p.x = (const Slot*)x; p.gw = gw;
  p.N = N; p.MS = (M + 7) / 8; p.KS = (K + 7) / 8;
  p.M_real = (int32_t)M_real; p.K_real = (int32_t)K_real;
  p.ldw_m = ldw_m; p.ldw_k = ldw_k; p.row_plane = row_plane; p.col_plane = col_plane;
  p.tiles_m = (p.MS + 15) / 16;
  const int tiles_k = (p.KS + 15) / 16;
  const dim3 grid((unsigned)(p.tiles_m * tiles_k));
  const int slot = profile_bracket_begin(stream);
  if (dtype == 1) hipLaunchKernelGGL(hlinear_wgrad_kernel<1>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(hlinear_wgrad_kernel<2>, grid, dim3(256), 0, stream, p);
  const int status = launch_status();
  profile_bracket_end_bytes(slot, stream, M, K, N, 17, 128, 128, 1, 2.0 * (double)N * (M + K) + 8.0 * (double)M * K, dtype);
  return status;
}

}  // extern "C"
