// crowd_labels.hip -- the offline ikNN label of the crowd application ON the device: for every pixel
// of a scene, 1 / (mean distance to its k nearest annotated heads + epsilon) -- generate_knn_map() and the i{k}nn_maps
// of the reference's database preprocessor (crowd/database_preprocessor.py:93-101,266-290), which runs a scikit-learn
// ball tree over all H*W pixel positions on the CPU.  Here: brute force, one thread per pixel, the head list streamed
// through LDS in chunks, the k (<= 8) smallest squared distances kept in registers.  H*W*M distance evaluations (0.8 G
// for a 768 x 1024 scene with 1000 heads): VALU-bound, milliseconds.
#include "common.h"

namespace srgan {

constexpr int KNN_MAX = 8, KNN_CHUNK = 1024;

__global__ __launch_bounds__(256) void crowd_iknn_kernel(const float* __restrict__ heads_yx, int M, int H, int W, int k,
                                                         float epsilon, float upper_bound, float* __restrict__ out) {
  __shared__ float2 chunk[KNN_CHUNK];
  const int64_t pixel = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = pixel < (int64_t)H * W;
  const float py = live ? (float)(pixel / W) : 0.f, px = live ? (float)(pixel % W) : 0.f;
  float best[KNN_MAX];                               // ascending squared distances
#pragma unroll
  for (int i = 0; i < KNN_MAX; ++i) best[i] = INFINITY;
  for (int base = 0; base < M; base += KNN_CHUNK) {
    const int count = min(KNN_CHUNK, M - base);
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 256) chunk[i] = make_float2(heads_yx[2 * (base + i)], heads_yx[2 * (base + i) + 1]);
    __syncthreads();
    for (int i = 0; i < count; ++i) {
      const float dy = chunk[i].x - py, dx = chunk[i].y - px;
      float d = fmaf(dy, dy, dx * dx);
      if (d < best[KNN_MAX - 1]) {                   // sorted insertion, fully unrolled: registers only
#pragma unroll
        for (int j = 0; j < KNN_MAX; ++j) {
          const float lower = fminf(best[j], d);
          d = fmaxf(best[j], d);
          best[j] = lower;
        }
      }
    }
  }
  if (!live) return;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < KNN_MAX; ++j)
    if (j < k) sum += fminf(sqrtf(best[j]), upper_bound);
  out[pixel] = 1.f / (sum / (float)k + epsilon);
}

}  // namespace srgan

using namespace srgan;

extern "C" int srgan_crowd_iknn_map(const float* heads_yx, int32_t M, int32_t H, int32_t W, int32_t k, float epsilon,
                                    float upper_bound, float* out, void* stream) {
  SRGAN_REQUIRE(heads_yx && out && M > 0 && H > 0 && W > 0 && k > 0 && k <= KNN_MAX, SRGAN_EINVAL,
                "srgan_crowd_iknn_map arguments");
  if (k > M) k = M;                                  // as the reference: min(number_of_neighbors, len(head_positions))
  const int64_t pixels = (int64_t)H * W;
  hipLaunchKernelGGL(crowd_iknn_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, heads_yx, M,
                     H, W, k, epsilon, upper_bound > 0.f ? upper_bound : INFINITY, out);
  return launch_status();
}
