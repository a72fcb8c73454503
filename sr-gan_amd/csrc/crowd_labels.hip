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

// ---- Gaussian density label (generate_density_label() of the reference with perspective = None, include_body = False:
// crowd/database_preprocessor.py:110-252, the "density{beta}" labels of :82-91).  Pass 1, one thread per head: the mean
// distance to its (at most 11) nearest heads INCLUDING itself -> sigma = mean * beta, window half-size r = int(2 sigma),
// and the normaliser 1 / sum of the unclipped (2r + 1)^2 window.  Pass 2, one thread per pixel: the sum of every head's
// normalised Gaussian whose window covers the pixel (clipping at the image border is implicit).  The caller rescales
// the label to the head count, as the reference's force_full_image_count_normalize.
struct HeadGaussian { float y, x, radius, inv_two_sigma_sq, inv_sum; };   // 20 bytes per head

constexpr int SPACING_NEIGHBOURS = 11;

__global__ __launch_bounds__(256) void head_gaussians_kernel(const float* __restrict__ heads_yx, int M, int H, int W,
                                                             float beta, HeadGaussian* __restrict__ out) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (i >= M) return;
  const float hy = heads_yx[2 * i], hx = heads_yx[2 * i + 1];
  const int k = min(SPACING_NEIGHBOURS, M);
  float best[SPACING_NEIGHBOURS];
#pragma unroll
  for (int j = 0; j < SPACING_NEIGHBOURS; ++j) best[j] = INFINITY;
  for (int m = 0; m < M; ++m) {
    const float dy = heads_yx[2 * m] - hy, dx = heads_yx[2 * m + 1] - hx;
    float d = fmaf(dy, dy, dx * dx);
    if (d < best[SPACING_NEIGHBOURS - 1]) {
#pragma unroll
      for (int j = 0; j < SPACING_NEIGHBOURS; ++j) {
        const float lower = fminf(best[j], d);
        d = fmaxf(best[j], d);
        best[j] = lower;
      }
    }
  }
  float mean = 0.f;
#pragma unroll
  for (int j = 0; j < SPACING_NEIGHBOURS; ++j)
    if (j < k) mean += sqrtf(best[j]);
  mean /= (float)k;
  const float sigma = mean * beta;
  const int r = (int)(sigma * 2.f);
  const float inv = 1.f / (2.f * sigma * sigma);
  float sum = 0.f;                                   // separable: (sum_d exp(-d^2 inv))^2
  for (int d = -r; d <= r; ++d) sum += expf(-(float)(d * d) * inv);
  HeadGaussian g;
  g.y = rintf(hy); g.x = rintf(hx);                  // np.rint: half to even, like rintf
  g.radius = (float)r; g.inv_two_sigma_sq = inv; g.inv_sum = 1.f / (sum * sum);
  // a window that does not reach the image at all is skipped by the reference ("Offset out of head gaussian bounds")
  const bool outside = g.y + r < 0.f || g.x + r < 0.f || g.y - r > (float)(H - 1) || g.x - r > (float)(W - 1);
  if (outside) g.inv_sum = 0.f;
  out[i] = g;
}

__global__ __launch_bounds__(256) void density_label_kernel(const HeadGaussian* __restrict__ heads, int M, int H, int W,
                                                            float* __restrict__ out) {
  __shared__ HeadGaussian chunk[512];
  const int64_t pixel = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = pixel < (int64_t)H * W;
  const float py = live ? (float)(pixel / W) : -1e9f, px = live ? (float)(pixel % W) : -1e9f;
  float acc = 0.f;
  for (int base = 0; base < M; base += 512) {
    const int count = min(512, M - base);
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 256) chunk[i] = heads[base + i];
    __syncthreads();
    for (int i = 0; i < count; ++i) {
      const HeadGaussian g = chunk[i];
      const float dy = py - g.y, dx = px - g.x;
      if (fabsf(dy) <= g.radius && fabsf(dx) <= g.radius) acc += g.inv_sum * expf(-fmaf(dy, dy, dx * dx) * g.inv_two_sigma_sq);
    }
  }
  if (live) out[pixel] = acc;
}

}  // namespace srgan

using namespace srgan;

extern "C" int srgan_crowd_density_label(const float* heads_yx, int32_t M, int32_t H, int32_t W, float beta, void* workspace,
                                         float* out, void* stream) {
  SRGAN_REQUIRE(heads_yx && workspace && out && M > 1 && H > 0 && W > 0 && beta > 0.f, SRGAN_EINVAL,
                "srgan_crowd_density_label arguments");
  HeadGaussian* gaussians = reinterpret_cast<HeadGaussian*>(workspace);
  hipLaunchKernelGGL(head_gaussians_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, heads_yx, M, H, W, beta,
                     gaussians);
  const int64_t pixels = (int64_t)H * W;
  hipLaunchKernelGGL(density_label_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     gaussians, M, H, W, out);
  return launch_status();
}

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
