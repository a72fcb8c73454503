// pointwise_ksplit.hip -- 1x1 / stride 1 convolution for FEW pixels and MANY input channels:
//   out[n, o, p] (=, +=) sum_i w[o, i] * f(in[n, i, p])      f = identity, or relu(batch_norm_eval(.)) (PRO)
//
// The bottleneck convolutions of DenseNet block 4 (16x16 planes, 512 ... 1000 input channels, reference
// crowd/models.py:338-341) have only 4096 pixels per batch of 16.  The streaming kernel of pointwise.hip gives each
// workgroup 128 pixels, so it has to shrink its row tile to 32 rows and split K over the grid (memset + fp32 atomics,
// the activation stream read four times) to fill the chip: 26 us for a 5 us problem.
// Here a workgroup owns 64 output rows x 32 pixels and its FOUR WAVES SPLIT K: wave w takes the 32-channel chunks
// w, w + 4, ...  Both MFMA operands are streamed from global memory straight into registers -- lane (i, half) owns 16
// consecutive input channels of weight row i (four 16-byte loads) and the matching 16 activation rows of its pixel
// (the order of a sum is free as long as A and B agree on it) -- so the main loop has no LDS traffic and no barrier;
// the next chunk is in flight while the current one is in the matrix pipe.  The four partial tiles are summed through
// LDS at the end.  No memset, no atomics, the activations are read twice (once per 64-row tile of 128 rows).
#include "common.h"
#include <stdlib.h>

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwKsplitParams {
  const float* in;      // [N, K, HW], batch stride in_bs
  const float* w;       // [M, K] row-major (k contiguous)
  float* out;           // [N, M, HW], batch stride out_bs
  const float* bias;
  int32_t N, K, M, HW;
  int64_t in_bs, out_bs;
  int32_t groups_per_image;
  int32_t accumulate;
  const float* bn_mean; const float* bn_inv; const float* bn_gamma; const float* bn_beta;   // PRO
};

constexpr int PKS_MI = 1;                 // 32 output rows per workgroup
constexpr int PKS_MAX_K = 2048;           // PRO: (a, b) of every input channel live in LDS

template <bool PRO>
__global__ __launch_bounds__(256, 2) void pointwise_ksplit_kernel(const PwKsplitParams p) {
  constexpr int MI = PKS_MI, ROWS = MI * 32, LDR = 33;
  // One LDS block: the (a, b) table of the prologue during the K loop, the four partial tiles afterwards (a barrier
  // separates the two uses) -- 34 KB instead of 50, i.e. four workgroups per CU instead of three.
  constexpr int RED_FLOATS = 4 * ROWS * LDR, COEF_FLOATS = PRO ? 2 * PKS_MAX_K : 0;
  __shared__ float smem[RED_FLOATS > COEF_FLOATS ? RED_FLOATS : COEF_FLOATS];
  float* red = smem;
  float2* coef = reinterpret_cast<float2*>(smem);

  const int tid = (int)threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = (int)blockIdx.x;
  const int n = group / p.groups_per_image;
  const int pix0 = (group - n * p.groups_per_image) * 32;
  const int m0 = (int)blockIdx.y * ROWS;

  if (PRO) {
    for (int k = tid; k < p.K; k += 256) {
      float a, b;
      bn_coefficients(p.bn_mean[k], p.bn_inv[k], p.bn_gamma[k], p.bn_beta[k], a, b);
      coef[k] = make_float2(a, b);
    }
    __syncthreads();
  }

  // weights: lane (i, half) reads 16 consecutive k of row i;  activations: lane (pixel, half) reads the same 16 k rows
  const float* w_lane[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) w_lane[mi] = p.w + (int64_t)min(m0 + mi * 32 + l31, p.M - 1) * p.K + 16 * lhi;
  const float* x_wave = p.in + (int64_t)n * p.in_bs + pix0;
  const uint32_t x_lane = (uint32_t)l31 + 16u * (uint32_t)lhi * (uint32_t)p.HW;

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  const int chunks = p.K / 32;
  float4 a0[MI][4], a1[MI][4];
  float b0[16], b1[16];
  auto fetch = [&](int chunk, float4 (&a)[MI][4], float (&b)[16]) {
    chunk = min(chunk, chunks - 1);                               // a wave's surplus fetch is never used
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) a[mi][q] = *reinterpret_cast<const float4*>(w_lane[mi] + chunk * 32 + 4 * q);
    const float* xc = x_wave + (int64_t)(chunk * 32) * p.HW;
#pragma unroll
    for (int s = 0; s < 16; ++s) b[s] = (xc + (int64_t)s * p.HW)[x_lane];
  };
  auto compute = [&](int chunk, const float4 (&a)[MI][4], const float (&b)[16]) {
    const float2* cf = &coef[PRO ? chunk * 32 + 16 * lhi : 0];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float bv = b[s];
      if (PRO) {
        const float2 c = cf[s];
        bv = fmaxf(fmaf(bv, c.x, c.y), 0.f);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float4 v = a[mi][s >> 2];
        const float av = (s & 3) == 0 ? v.x : ((s & 3) == 1 ? v.y : ((s & 3) == 2 ? v.z : v.w));
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi], 0, 0, 0);
      }
    }
  };

  int chunk = wave;
  if (chunk < chunks) {
    fetch(chunk, a0, b0);
    // (The scheduler sinks part of each prefetch towards its first use; pinning the order with sched_barrier(0)
    // measured 25 % slower, so the schedule is left to the compiler.)
    for (; chunk < chunks; chunk += 8) {
      fetch(chunk + 4, a1, b1);
      compute(chunk, a0, b0);
      if (chunk + 4 < chunks) {
        fetch(chunk + 8, a0, b0);
        compute(chunk + 4, a1, b1);
      }
    }
  }

  // ---- sum the four waves' partial tiles (C/D fragment: column = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5))
  if (PRO) __syncthreads();                  // every wave is done with the coefficient table that `red` overlays
  float* mine = red + wave * (ROWS * LDR);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDR + l31] = acc[mi][r];
  __syncthreads();
  float* out_n = p.out + (int64_t)n * p.out_bs + pix0;
#pragma unroll
  for (int e = 0; e < ROWS * 32 / 256; ++e) {
    const int idx = tid + 256 * e;
    const int row = idx >> 5, col = idx & 31;
    const int at = row * LDR + col;
    float v = (red[at] + red[ROWS * LDR + at]) + (red[2 * ROWS * LDR + at] + red[3 * ROWS * LDR + at]);
    const int o = m0 + row;
    if (o >= p.M) continue;
    if (p.bias) v += p.bias[o];
    float* dst = out_n + (int64_t)o * p.HW + col;
    if (p.accumulate == 1) *dst += v;
    else __builtin_nontemporal_store(v, dst);
  }
}

int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn,
                        int split, int akf = 0, int bkf = 0, int64_t b_unique = 0, int precision = 0);

bool pointwise_ksplit_enabled() {
  static const bool disabled = getenv("SRGAN_NO_PW_KSPLIT") != nullptr;
  return !disabled;
}

// Few pixels, long K: where the streaming kernel would have to shrink its tile and split K over the grid.
bool pointwise_ksplit_wanted(int32_t N, int32_t K, int32_t M, int32_t HW, bool fused_bn) {
  // measured (608 ... 992 -> 128 channels, batch 16): 16x16 planes 15.5 vs 26 us for the streaming kernel, 32x32 planes
  // 50 vs 46 us -- the strided weight rows of 1024 workgroups saturate the L2 -> CU path -- so only the smallest take it
  static const int max_groups = getenv("SRGAN_PKS_GROUPS") ? atoi(getenv("SRGAN_PKS_GROUPS")) : 256;
  if (!pointwise_ksplit_enabled() || K % 32 != 0 || K < 256 || HW % 32 != 0) return false;
  if (fused_bn && K > PKS_MAX_K) return false;
  const int64_t groups = (int64_t)N * HW / 32;
  return groups * ((M + 127) / 128) <= max_groups;
}

int pointwise_ksplit_run(const float* in, int64_t in_bs, const float* w, const float* bias, float* out, int64_t out_bs,
                         int32_t N, int32_t K, int32_t M, int32_t HW, int accumulate, hipStream_t stream,
                         const float* const* bn) {
  PwKsplitParams p;
  p.in = in; p.w = w; p.out = out; p.bias = bias;
  p.N = N; p.K = K; p.M = M; p.HW = HW; p.in_bs = in_bs; p.out_bs = out_bs; p.accumulate = accumulate;
  p.bn_mean = bn ? bn[0] : nullptr; p.bn_inv = bn ? bn[1] : nullptr;
  p.bn_gamma = bn ? bn[2] : nullptr; p.bn_beta = bn ? bn[3] : nullptr;
  p.groups_per_image = HW / 32;
  const int64_t groups = (int64_t)N * p.groups_per_image;
  const int tiles_m = (M + PKS_MI * 32 - 1) / (PKS_MI * 32);
  SRGAN_REQUIRE(groups < ((int64_t)1 << 31) && tiles_m <= 65535, SRGAN_ERANGE, "pointwise (K split over waves) grid");
  SRGAN_REQUIRE(K % 32 == 0 && (((uintptr_t)w) & 15) == 0, SRGAN_EINVAL, "pointwise (K split over waves) weights");
  dim3 grid((unsigned)groups, (unsigned)tiles_m, 1);
  const int profile_slot = profile_bracket_begin(stream);
  if (bn) hipLaunchKernelGGL(pointwise_ksplit_kernel<true>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(pointwise_ksplit_kernel<false>, grid, dim3(256), 0, stream, p);
  const int status = launch_status();
  profile_bracket_end(profile_slot, stream, M, (int64_t)N * HW, K, 8, PKS_MI * 32, 32, 4);
  return status;
}

}  // namespace srgan
