// Laboratory copy of the pointwise loop: out[M][P] = W[M][K] * X[K][P], stripped to the essentials.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// PF: prefetch the A fragments of the next k-pair (double-buffered registers).  LB: __launch_bounds__ second argument.
template <int MI, int NI, int BK, bool PF, int LB>
__global__ __launch_bounds__(256, LB) void pw(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ out,
                                              int K, int P, int M) {
  constexpr int BM = MI * 32, KP = BK / 2, LDA = BM + 1, EA = BM * BK / 256;
  __shared__ float lds[2 * BK * LDA];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = M / BM;
  const int tm = blockIdx.x % tiles_m;
  const int pix0 = ((blockIdx.x / tiles_m) * 4 + wave) * 32 * NI;
  const int m0 = tm * BM;
  const float* b_wave = X + pix0;
  const uint32_t lane_off = (uint32_t)l31 + (uint32_t)lhi * (uint32_t)P;
  int a_k[EA], a_m[EA];
#pragma unroll
  for (int e = 0; e < EA; ++e) { const int flat = e * 256 + tid; a_k[e] = flat % BK; a_m[e] = flat / BK; }
  float ra[EA], b0[NI][KP], b1[NI][KP];
  f32x16 acc[MI][NI];
  for (int mi = 0; mi < MI; ++mi) for (int ni = 0; ni < NI; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  auto fetch_a = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) ra[e] = W[(m0 + a_m[e]) * K + min(k0 + a_k[e], K - 1)];
  };
  auto stage_a = [&](float* As) {
#pragma unroll
    for (int e = 0; e < EA; ++e) As[a_k[e] * LDA + a_m[e]] = ra[e];
  };
  auto fetch_b = [&](int k0, float (&dst)[NI][KP]) {
#pragma unroll
    for (int q = 0; q < KP; ++q)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) dst[ni][q] = (b_wave + (size_t)(k0 + 2 * q) * P)[lane_off + 32 * ni];
  };
  auto slice = [&](int k0, const float (&bc)[NI][KP], float (&bn)[NI][KP], int buffer) {
    const bool more = k0 + BK < K;
    const float* As = lds + buffer * (BK * LDA) + lhi * LDA + l31;
    fetch_a(k0 + BK);
    float a[2][MI];
    if (PF) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[0][mi] = As[mi * 32];
    }
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      if (PF) {
        if (q + 1 < KP) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[(q + 1) & 1][mi] = As[(2 * (q + 1)) * LDA + mi * 32];
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[q & 1][mi] = As[(2 * q) * LDA + mi * 32];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mi], bc[ni][q], acc[mi][ni], 0, 0, 0);
      const int kn = min(k0 + BK + 2 * q, K - 2);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bn[ni][q] = (b_wave + (size_t)kn * P)[lane_off + 32 * ni];
    }
    if (more) { stage_a(lds + (buffer ^ 1) * (BK * LDA)); __syncthreads(); }
  };
  fetch_a(0); fetch_b(0, b0); stage_a(lds); __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    slice(k0, b0, b1, 0);
    if (k0 + BK < K) slice(k0 + BK, b1, b0, 1);
  }
  float* out_lane = out + pix0 + l31;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        out_lane[(size_t)(m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P + 32 * ni] = acc[mi][ni][r];
}

template <int MI, int NI, int BK, bool PF, int LB>
void run(int K, int P, int M) {
  float *X, *W, *out;
  (void)hipMalloc(&X, (size_t)K * P * 4); (void)hipMalloc(&W, (size_t)M * K * 4); (void)hipMalloc(&out, (size_t)M * P * 4);
  {
    size_t nx = (size_t)K * P, nw = (size_t)M * K;
    float* h = (float*)malloc((nx > nw ? nx : nw) * 4);
    for (size_t i = 0; i < nx; ++i) h[i] = (float)((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 1e-3f;
    (void)hipMemcpy(X, h, nx * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h, nw * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = P / (128 * NI) * (M / (MI * 32)), reps = 10;
  hipLaunchKernelGGL((pw<MI, NI, BK, PF, LB>), dim3(grid), dim3(256), 0, 0, X, W, out, K, P, M);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pw<MI, NI, BK, PF, LB>), dim3(grid), dim3(256), 0, 0, X, W, out, K, P, M);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("MI %d NI %d BK %2d PF %d LB %d | M %4d K %4d P %7d: %7.1f us  %6.1f TF/s\n", MI, NI, BK, (int)PF, LB, M, K, P, 1e3 * ms / reps,
         2.0 * M * K * (double)P * reps / ms / 1e9);
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(out);
}

// Vector variant: lane j of a 32-lane block owns NI CONSECUTIVE pixels (one NI*4-byte load / store per k row); MFMA
// column block ni then holds pixels {NI*j + ni}.  Half the VMEM instructions of the scalar NI form, rows are touched in
// 32*NI*4-byte runs.
template <int MI, int NI, int BK, int LB>
__global__ __launch_bounds__(256, LB) void pwv(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ out,
                                               int K, int P, int M) {
  typedef float vec __attribute__((ext_vector_type(NI)));
  constexpr int BM = MI * 32, KP = BK / 2, LDA = BM + 1, EA = BM * BK / 256;
  __shared__ float lds[2 * BK * LDA];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = M / BM;
  const int tm = blockIdx.x % tiles_m;
  const int pix0 = ((blockIdx.x / tiles_m) * 4 + wave) * 32 * NI;
  const int m0 = tm * BM;
  const float* b_wave = X + pix0;
  const uint32_t lane_off = (uint32_t)l31 * NI + (uint32_t)lhi * (uint32_t)P;
  int a_k[EA], a_m[EA];
#pragma unroll
  for (int e = 0; e < EA; ++e) { const int flat = e * 256 + tid; a_k[e] = flat % BK; a_m[e] = flat / BK; }
  float ra[EA];
  vec b0[KP], b1[KP];
  f32x16 acc[MI][NI];
  for (int mi = 0; mi < MI; ++mi) for (int ni = 0; ni < NI; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  auto fetch_a = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) ra[e] = W[(m0 + a_m[e]) * K + min(k0 + a_k[e], K - 1)];
  };
  auto stage_a = [&](float* As) {
#pragma unroll
    for (int e = 0; e < EA; ++e) As[a_k[e] * LDA + a_m[e]] = ra[e];
  };
  auto fetch_b = [&](int k0, vec (&dst)[KP]) {
#pragma unroll
    for (int q = 0; q < KP; ++q) dst[q] = *reinterpret_cast<const vec*>(b_wave + (size_t)(k0 + 2 * q) * P + lane_off);
  };
  auto slice = [&](int k0, const vec (&bc)[KP], vec (&bn)[KP], int buffer) {
    const bool more = k0 + BK < K;
    const float* As = lds + buffer * (BK * LDA) + lhi * LDA + l31;
    fetch_a(k0 + BK);
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      float a[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = As[(2 * q) * LDA + mi * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bc[q][ni], acc[mi][ni], 0, 0, 0);
      const int kn = min(k0 + BK + 2 * q, K - 2);
      bn[q] = *reinterpret_cast<const vec*>(b_wave + (size_t)kn * P + lane_off);
    }
    if (more) { stage_a(lds + (buffer ^ 1) * (BK * LDA)); __syncthreads(); }
  };
  fetch_a(0); fetch_b(0, b0); stage_a(lds); __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    slice(k0, b0, b1, 0);
    if (k0 + BK < K) slice(k0 + BK, b1, b0, 1);
  }
  float* out_lane = out + pix0 + l31 * NI;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      vec v;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) v[ni] = acc[mi][ni][r];
      *reinterpret_cast<vec*>(out_lane + (size_t)(m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P) = v;
    }
}

template <int MI, int NI, int BK, int LB>
void runv(int K, int P, int M) {
  float *X, *W, *out;
  (void)hipMalloc(&X, (size_t)K * P * 4); (void)hipMalloc(&W, (size_t)M * K * 4); (void)hipMalloc(&out, (size_t)M * P * 4);
  {
    size_t nx = (size_t)K * P, nw = (size_t)M * K;
    float* h = (float*)malloc((nx > nw ? nx : nw) * 4);
    for (size_t i = 0; i < nx; ++i) h[i] = (float)((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 1e-3f;
    (void)hipMemcpy(X, h, nx * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h, nw * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = P / (128 * NI) * (M / (MI * 32)), reps = 10;
  hipLaunchKernelGGL((pwv<MI, NI, BK, LB>), dim3(grid), dim3(256), 0, 0, X, W, out, K, P, M);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pwv<MI, NI, BK, LB>), dim3(grid), dim3(256), 0, 0, X, W, out, K, P, M);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("VEC MI %d NI %d BK %2d LB %d | M %4d K %4d P %7d: %7.1f us  %6.1f TF/s\n", MI, NI, BK, LB, M, K, P, 1e3 * ms / reps,
         2.0 * M * K * (double)P * reps / ms / 1e9);
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(out);
}
#define BOTHV(...) runv<__VA_ARGS__>(256, 262144, 128); runv<__VA_ARGS__>(1024, 65536, 128); runv<__VA_ARGS__>(128, 262144, 256);

#define BOTH(...) run<__VA_ARGS__>(256, 262144, 128); run<__VA_ARGS__>(1024, 65536, 128); run<__VA_ARGS__>(128, 262144, 256);
int main() {
  BOTH(4, 1, 32, false, 2)
  BOTHV(2, 2, 32, 2)
  BOTHV(4, 2, 16, 2)
  BOTHV(2, 4, 16, 2)
  BOTHV(1, 4, 32, 2)
  return 0;
}
