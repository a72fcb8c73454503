// Which ingredient of the pointwise loop costs MFMA throughput?  4 accumulators, 2 workgroups of 4 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE bits: 1 = A operand from LDS (4 ds_read per 4 MFMAs), 2 = B operand from global (1 load per 4 MFMAs, prefetched
// one slice of 16 ahead), 4 = __syncthreads every 16 k-pairs, 8 = stores of acc at the end of every 64 k-pairs
template <int MODE>
__global__ __launch_bounds__(256, 2) void mix(const float* __restrict__ in, float* __restrict__ out, int iters, int stride) {
  __shared__ float lds[32 * 129];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  for (int i = tid; i < 32 * 129; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* src = in + (size_t)blockIdx.x * 4096 + tid;
  float b0[16], b1[16];
  for (int q = 0; q < 16; ++q) b0[q] = (MODE & 2) ? src[(size_t)q * stride] : (float)q;
  float a_reg[4] = {1.f, 2.f, 3.f, 4.f};
  const float* a_lane = lds + lhi * 129 + l31;
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float (&bc)[16] = half ? b1 : b0;
      float (&bn)[16] = half ? b0 : b1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float a[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[mi] = (MODE & 1) ? a_lane[(2 * q) * 129 + mi * 32] : a_reg[mi];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bc[q], acc[mi], 0, 0, 0);
        if (MODE & 2) bn[q] = src[(size_t)((it + half + 1) * 16 + q) * stride];
        else bn[q] = bc[q];
      }
      if (MODE & 4) __syncthreads();
    }
    if ((MODE & 8) && (it & 2)) {
      for (int mi = 0; mi < 4; ++mi)
        for (int r = 0; r < 16; ++r) out[(size_t)(blockIdx.x * 64 + mi * 16 + r) * 256 + tid] = acc[mi][r];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run() {
  float *in, *out;
  (void)hipMalloc(&in, (size_t)1 << 30); (void)hipMalloc(&out, (size_t)1 << 28);
  (void)hipMemset(in, 0, (size_t)1 << 30);
  const int iters = 256, grid = 512;
  const int stride = 65536;     // floats between consecutive k rows (an image plane)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 8, stride % 4096);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters, 4096);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 4096.0 * 4 * 16 * iters * 4.0 * grid;
  printf("mode %2d (lds %d global %d barrier %d stores %d): %.1f TF/s\n", MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, (MODE >> 3) & 1, flops / ms / 1e9);
  (void)hipFree(in); (void)hipFree(out);
}
int main() {
  run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<7>(); run<8>(); run<15>();
  return 0;
}
