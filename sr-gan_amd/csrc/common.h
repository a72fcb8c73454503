// common.h -- error plumbing and small device helpers shared by every translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace srgan {

// Status convention of the C ABI (include/srgan_hip.h): 0 ok, <0 argument/shape error, >0 hipError_t.
enum { SRGAN_OK = 0, SRGAN_EINVAL = -1, SRGAN_EUNSUPPORTED = -2, SRGAN_ERANGE = -3 };

void set_error(const char* fmt, ...);

#define SRGAN_HIP(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::srgan::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (int)e_;                                                                    \
    }                                                                                    \
  } while (0)

#define SRGAN_REQUIRE(cond, code, msg)                                                   \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      ::srgan::set_error("%s: requirement (%s) failed", msg, #cond);                     \
      return code;                                                                       \
    }                                                                                    \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return SRGAN_OK;
}

// Zero-fills are kernels of this library, not hipMemsetAsync: a memset NODE of a captured HIP graph replayed wrongly for
// sizes that are not a multiple of 16 bytes (ROCm 7.2: 400 bytes left garbage behind), and the runtime's own fill kernel
// takes ~5 us per launch.  `count` / `width` in floats; rows `pitch` floats apart.
int zero_floats(float* p, int64_t count, hipStream_t stream);
int zero_rows(float* p, int64_t pitch, int64_t width, int64_t rows, hipStream_t stream);

// y = fma(x, a, b) is frozen batch-norm: a = inv_std * gamma, b = beta - mean * a.  Every kernel that evaluates the
// normalisation (forward, fused prologues, the mask recomputed in the backward) goes through this one function so
// that the sign of y -- the ReLU mask -- is bit-identical everywhere.
__device__ __forceinline__ void bn_coefficients(float mean, float inv_std, float gamma, float beta, float& a, float& b) {
  a = __fmul_rn(inv_std, gamma);
  b = __fsub_rn(beta, __fmul_rn(mean, a));
}

// Sum over the 32 lanes of each half of the wave, valid in lanes 16-31 / 48-63: four rotations inside the rows of 16
// (DPP row_ror) and one row broadcast (row_bcast:15 into rows 1 and 3) -- five VALU instructions, no LDS.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_move<0x128, 0xF>(v);
  v += dpp_move<0x124, 0xF>(v);
  v += dpp_move<0x122, 0xF>(v);
  v += dpp_move<0x121, 0xF>(v);
  v += dpp_move<0x142, 0xA>(v);
  return v;
}

// Backward of relu(batch_norm_eval(x)) applied in the epilogue of a data-gradient kernel (pointwise.hip, conv3x3.hip):
// x is the tensor the normalisation read (batch stride x_bs, same channels as the kernel's output), bn = {mean,
// inv_std, gamma, beta}; g_gamma / g_beta (both or neither) are accumulated into.
struct BnBackwardEpilogue {
  const float* x; int64_t x_bs;
  const float* bn[4];
  float* g_gamma; float* g_beta;
  // Deferred form: the per-workgroup parameter sums go to this caller-owned region ([2][tiles][C] floats) and are NOT
  // reduced by the call -- a whole dense block's convolutions are reduced by one srgan_bn_partial_reduce_batched launch.
  float* partial_out;
};

// Grid for a grid-stride streaming kernel: enough blocks to fill 256 CUs x 8, never more than the work.
inline unsigned stream_grid(int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (unsigned)blocks;
}

// Sum across the 64 lanes of a wavefront (DPP/shuffle butterfly); every lane returns the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int offset = 32; offset > 0; offset >>= 1) v += __shfl_xor(v, offset, 64);
  return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0.
__device__ __forceinline__ float block_sum_256(float v, float* scratch4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch4[wave] = v;
  __syncthreads();
  float total = 0.f;
  if (threadIdx.x == 0) total = scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
  return total;
}

}  // namespace srgan
