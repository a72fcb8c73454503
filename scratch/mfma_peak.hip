// Pure-MFMA ceiling of v_mfma_f32_32x32x2_f32: ACC independent accumulators per wave, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ACC>
__global__ __launch_bounds__(256) void peak(float* out, int iters, float a0, float b0) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // operands: 'random-looking' per-lane values (data-dependent power: constant operands clock higher)
  const unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float a = a0 * ((float)((h >> 8) & 0xffff) / 32768.f - 1.f), b = b0 * ((float)((h >> 12) & 0xffff) / 32768.f - 1.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a;     // keeps the accumulators bounded and the operand bits toggling
  }
  float s = 0.f;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC>
void run(int wgs_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 4096, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<ACC>, dim3(grid), dim3(256), 0, 0, out, 64, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(peak<ACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 4096.0 * ACC * iters * 4.0 * grid;
  printf("acc %d  waves/SIMD %d : %.1f TF/s\n", ACC, wgs_per_cu, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
  return 0;
}
