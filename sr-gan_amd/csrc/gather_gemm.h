// gather_gemm.h -- the one contraction primitive of the library.
//
//   C[cm(i) + cn(j)]  (=, +=, atomic+=)  sum_k  A[am(i) + ak(k)] * B[bk(k) + bn(j)]   (+ bias[c(i)])
//
// Every index role (A rows, A depth, B depth, B columns, C rows, C columns) is decoded by a Dec3:
// idx -> (c, a, b) by two constant divisions, then affine maps to an element offset and a 2-D
// coordinate (h, w).  B elements are zero when the summed (h, w) falls outside [0,hlim)x[0,wlim):
// that is how convolution padding, stride-parity classes and ragged tiles are expressed.
// Convolution forward / backward-data / backward-weight, transposed convolution and plain strided
// GEMM are all instances (conv_plan.h builds them).  Shared by the device kernels (.hip) and by the
// CPU index-math emulator used in tests (tests/csrc), hence plain C++ with a host/device macro.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GG_HD __host__ __device__ __forceinline__
#else
#define GG_HD inline
#endif

namespace srgan {

// Division of a non-negative 31-bit integer by a constant (round-up magic number form).
struct FastDiv {
  uint32_t d, mul, shr, add_mask;   // q = (umulhi(n, mul) + (n & add_mask)) >> shr  (branch-free, d == 1 included)
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d ? d : 1u;
  f.add_mask = 0u;
  if (f.d == 1u) { f.mul = 0u; f.shr = 0u; f.add_mask = 0xFFFFFFFFu; return f; }
  uint32_t log2d = 0;
  while ((1ull << log2d) < f.d) ++log2d;           // ceil(log2 d)
  const uint32_t p = 31u + log2d;
  f.mul = (uint32_t)(((1ull << p) + f.d - 1ull) / f.d);
  f.shr = p - 32u;
  return f;
}

GG_HD uint32_t fd_umulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

GG_HD uint32_t fd_div(uint32_t n, const FastDiv& f) {
  return (fd_umulhi(n, f.mul) + (n & f.add_mask)) >> f.shr;
}

// idx -> (c, a, b) with idx = (c * A + a) * B + b; then affine offset / coordinate maps.
struct Dec3 {
  FastDiv div_ab;       // A * B
  FastDiv div_b;        // B
  int32_t limit;        // idx valid iff 0 <= idx < limit
  int32_t off_c, off_a, off_b, off0;   // element offset = c*off_c + a*off_a + b*off_b + off0 (wrapping)
  int32_t h_a, h0;      // h = a*h_a + h0
  int32_t w_b, w0;      // w = b*w_b + w0
};

struct Side {
  uint32_t off;
  int32_t h, w;
  int32_t c;            // leading component (bias index for C rows)
  bool valid;
};

GG_HD Side decode(const Dec3& d, int32_t idx) {
  Side s;
  s.valid = (uint32_t)idx < (uint32_t)d.limit;
  const uint32_t u = s.valid ? (uint32_t)idx : 0u;
  const uint32_t c = fd_div(u, d.div_ab);
  const uint32_t r = u - c * d.div_ab.d;
  const uint32_t a = fd_div(r, d.div_b);
  const uint32_t b = r - a * d.div_b.d;
  s.off = c * (uint32_t)d.off_c + a * (uint32_t)d.off_a + b * (uint32_t)d.off_b + (uint32_t)d.off0;
  s.h = (int32_t)a * d.h_a + d.h0;
  s.w = (int32_t)b * d.w_b + d.w0;
  s.c = (int32_t)c;
  return s;
}

inline Dec3 dec_linear(int32_t limit, int32_t stride) {
  Dec3 d;
  d.div_ab = make_fastdiv(1); d.div_b = make_fastdiv(1);
  d.limit = limit; d.off_c = stride; d.off_a = 0; d.off_b = 0; d.off0 = 0;
  d.h_a = 0; d.h0 = 0; d.w_b = 0; d.w0 = 0;
  return d;
}

inline Dec3 dec_3d(int32_t count_c, int32_t A, int32_t B, int32_t off_c, int32_t off_a, int32_t off_b, int32_t off0,
                   int32_t h_a, int32_t h0, int32_t w_b, int32_t w0) {
  Dec3 d;
  d.div_ab = make_fastdiv((uint32_t)(A * B)); d.div_b = make_fastdiv((uint32_t)B);
  d.limit = count_c * A * B; d.off_c = off_c; d.off_a = off_a; d.off_b = off_b; d.off0 = off0;
  d.h_a = h_a; d.h0 = h0; d.w_b = w_b; d.w0 = w0;
  return d;
}

// GG_ORDERED_*: a K split whose slices meet in a fixed order inside the launch (split_finish.h): partial tiles through
// `partial`, one ticket per output tile in `tickets`, the tile's last workgroup stores / accumulates the total.
enum StoreMode { GG_STORE = 0, GG_ACCUMULATE = 1, GG_ATOMIC = 2, GG_PARTIAL = 3, GG_ORDERED_STORE = 4, GG_ORDERED_ACCUMULATE = 5 };
// How the K slices of a split launch are combined (GatherGemm::use_partial, chosen by gg_prepare)
enum SplitCombine { GG_COMBINE_ATOMIC = 0, GG_COMBINE_PARTIAL_TINY = 1, GG_COMBINE_PARTIAL_WIDE = 2, GG_COMBINE_ORDERED = 3 };

struct GatherGemm {
  const float* A; Dec3 am, ak;
  const float* B; Dec3 bk, bn; int32_t hlim, wlim;     // B bounds on (bk.h + bn.h, bk.w + bn.w)
  float* C; Dec3 cm, cn;
  const float* bias;                                     // optional, indexed by the leading component of
  int32_t bias_cols;                                     //   cm (0) or cn (1)
  int32_t M, N, K;
  int32_t a_kfast, b_kfast;                              // staging order: lanes along k (1) or along m/n (0)
  int32_t mode;                                          // StoreMode
  int32_t split_k, k_per_split;                          // filled by the launcher
  int32_t debug;                                         // tuning experiments (SRGAN_GG_DEBUG): 1 no re-staging, 2 no MFMA
  float* partial;                                        // GG_PARTIAL: K-slice z stores to partial[(z*M + i)*N + j]; GG_ORDERED_*: [tile][slice][accumulators]
  unsigned int* tickets;                                 // GG_ORDERED_*: one per output tile
  int32_t use_partial;                                   // launcher: SplitCombine -- how the K slices are combined
  int64_t b_unique;                                      // distinct B elements the gather touches (0: K * N); bookkeeping
  int32_t precision;                                     // MFMA operand type: 0 fp32 (exact), 1 bf16, 2 fp16 (fp32 accumulate,
};                                                       //   for the live profile only.  precision: fp32 data in HBM and LDS, rounded when the MFMA operands are formed

// Reference semantics of one output element (used by the CPU emulator and by the direct kernel).
GG_HD float gg_a(const GatherGemm& p, const Side& m, const Side& k) {
  return (m.valid && k.valid) ? p.A[m.off + k.off] : 0.0f;
}

GG_HD float gg_b(const GatherGemm& p, const Side& k, const Side& n) {
  const bool ok = k.valid && n.valid && (uint32_t)(k.h + n.h) < (uint32_t)p.hlim &&
                  (uint32_t)(k.w + n.w) < (uint32_t)p.wlim;
  return ok ? p.B[k.off + n.off] : 0.0f;
}

}  // namespace srgan
