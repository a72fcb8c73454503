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

// ---- Gaussian density label (generate_density_label() of the reference, crowd/database_preprocessor.py:110-223, with
// make_gaussian :226-249; the "density{beta}" labels of :82-91 are its perspective = None form).  Pass 1, one thread per head
// -> up to two windowed Gaussians (head, body) with their normalisers:
//   * sigma of the head: perspective map given: 0.2 m x perspective[y, x] (perspective[y, 0] when x is outside the map;
//     `ignore_tiny` drops heads whose perspective is < 3.1 -- they are not counted either); no map: beta x the mean distance
//     to its (at most 11) nearest heads INCLUDING itself; `perspective_resizing = False`: 8 pixels.  Window half-size
//     r = int(2 sigma), normaliser 1 / (body_parts x sum of the unclipped window), body_parts = 2 with `include_body`;
//   * `include_body` with a perspective map: a second Gaussian 0.875 m below the head, sigma (0.2 m, 0.5 m) x perspective
//     (x, y), window (int(2 sigma_x), int(2 sigma_y)), normalised the same way.
// Pass 2, one thread per pixel: the sum of every Gaussian whose window covers the pixel (clipping at the image border is
// implicit).  The caller rescales the label to the number of counted heads (force_full_image_count_normalize).
struct LabelGaussian { float y, x, ry, rx, inv_two_sy_sq, inv_two_sx_sq, inv_sum, counted; };   // 32 bytes

constexpr int SPACING_NEIGHBOURS = 11;
constexpr int LABEL_INCLUDE_BODY = 1, LABEL_IGNORE_TINY = 2, LABEL_FIXED_SIGMA = 4, LABEL_XY_ORDER = 8;

__device__ __forceinline__ float window_sum(int r, float inv) {
  float sum = 0.f;
  for (int d = -r; d <= r; ++d) sum += expf(-(float)(d * d) * inv);
  return sum;
}

__global__ __launch_bounds__(256) void head_gaussians_kernel(const float* __restrict__ heads, int M, int H, int W, float beta,
                                                             const float* __restrict__ perspective, int flags,
                                                             LabelGaussian* __restrict__ out) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (i >= M) return;
  const int first = (flags & LABEL_XY_ORDER) ? 1 : 0;             // positions are (y, x) pairs unless the order is (x, y)
  const float hy = heads[2 * i + first], hx = heads[2 * i + 1 - first];
  LabelGaussian head, body;
  head.y = rintf(hy); head.x = rintf(hx);                         // np.rint: half to even, like rintf
  head.counted = 1.f;
  body = head;
  body.inv_sum = 0.f; body.counted = 0.f; body.ry = body.rx = 0.f; body.inv_two_sy_sq = body.inv_two_sx_sq = 0.f;
  const float parts = (flags & LABEL_INCLUDE_BODY) ? 2.f : 1.f;
  float sigma, scale = 0.f;                                       // scale: the perspective value at the head (pixels per metre)
  if (flags & LABEL_FIXED_SIGMA) {
    sigma = 8.f;
  } else if (perspective != nullptr) {
    const int py = min(max((int)head.y, 0), H - 1);
    const int px = (head.x >= 0.f && head.x < (float)W) ? (int)head.x : 0;
    scale = perspective[(int64_t)py * W + px];
    sigma = scale * 0.2f;                                         // head_standard_deviation_meters
    if ((flags & LABEL_IGNORE_TINY) && scale < 3.1f) {
      head.inv_sum = 0.f; head.counted = 0.f; head.ry = head.rx = 0.f; head.inv_two_sy_sq = head.inv_two_sx_sq = 0.f;
      out[2 * i] = head; out[2 * i + 1] = body;
      return;
    }
  } else {
    const int k = min(SPACING_NEIGHBOURS, M);
    float best[SPACING_NEIGHBOURS];
#pragma unroll
    for (int j = 0; j < SPACING_NEIGHBOURS; ++j) best[j] = INFINITY;
    for (int m = 0; m < M; ++m) {
      const float dy = heads[2 * m + first] - hy, dx = heads[2 * m + 1 - first] - hx;
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
    sigma = mean / (float)k * beta;
  }
  const int r = (int)(sigma * 2.f);
  const float inv = 1.f / (2.f * sigma * sigma);
  const float sum = window_sum(r, inv);                           // separable: (sum_d exp(-d^2 inv))^2
  head.ry = head.rx = (float)r; head.inv_two_sy_sq = head.inv_two_sx_sq = inv; head.inv_sum = 1.f / (parts * sum * sum);
  // a window that does not reach the image at all is skipped by the reference ("Offset out of head gaussian bounds"),
  // together with the person's body; the head still counts
  const bool outside = head.y + r < 0.f || head.x + r < 0.f || head.y - r > (float)(H - 1) || head.x - r > (float)(W - 1);
  if (outside) head.inv_sum = 0.f;
  if (!outside && perspective != nullptr && (flags & LABEL_INCLUDE_BODY) && !(flags & LABEL_FIXED_SIGMA)) {
    const float sx = scale * 0.2f, sy = scale * 0.5f;             // body_width / body_height_standard_deviation_meters
    const int rx = (int)(sx * 2.f), ry = (int)(sy * 2.f);
    body.y = head.y + (float)(int)(scale * 0.875f);               // body_height_offset_meters
    body.x = head.x;
    body.ry = (float)ry; body.rx = (float)rx;
    body.inv_two_sy_sq = 1.f / (2.f * sy * sy); body.inv_two_sx_sq = 1.f / (2.f * sx * sx);
    body.inv_sum = 1.f / (parts * window_sum(ry, body.inv_two_sy_sq) * window_sum(rx, body.inv_two_sx_sq));
  }
  out[2 * i] = head; out[2 * i + 1] = body;
}

__global__ __launch_bounds__(256) void density_label_kernel(const LabelGaussian* __restrict__ gaussians, int count, int H, int W,
                                                            float* __restrict__ out) {
  __shared__ LabelGaussian chunk[512];
  const int64_t pixel = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = pixel < (int64_t)H * W;
  const float py = live ? (float)(pixel / W) : -1e9f, px = live ? (float)(pixel % W) : -1e9f;
  float acc = 0.f;
  for (int base = 0; base < count; base += 512) {
    const int here = min(512, count - base);
    __syncthreads();
    for (int i = threadIdx.x; i < here; i += 256) chunk[i] = gaussians[base + i];
    __syncthreads();
    for (int i = 0; i < here; ++i) {
      const LabelGaussian g = chunk[i];
      const float dy = py - g.y, dx = px - g.x;
      if (g.inv_sum != 0.f && fabsf(dy) <= g.ry && fabsf(dx) <= g.rx)
        acc += g.inv_sum * expf(-fmaf(dy * dy, g.inv_two_sy_sq, dx * dx * g.inv_two_sx_sq));
    }
  }
  if (live) out[pixel] = acc;
}

}  // namespace srgan

using namespace srgan;

// workspace: 64 bytes per head (two LabelGaussian records; float 7 of the first = 1 when the head counts).
// perspective: NULL or a device map [H][W]; flags: 1 include_body, 2 ignore_tiny, 4 perspective_resizing = False
// (sigma = 8 pixels), 8 positions are (x, y) pairs instead of (y, x).
extern "C" int srgan_crowd_density_label(const float* heads, int32_t M, int32_t H, int32_t W, float beta,
                                         const float* perspective, int32_t flags, void* workspace, float* out, void* stream) {
  SRGAN_REQUIRE(heads && workspace && out && M > 0 && H > 0 && W > 0 && (flags & ~15) == 0 &&
                (perspective != nullptr || (flags & LABEL_FIXED_SIGMA) || (M > 1 && beta > 0.f)), SRGAN_EINVAL,
                "srgan_crowd_density_label arguments");
  LabelGaussian* gaussians = reinterpret_cast<LabelGaussian*>(workspace);
  hipLaunchKernelGGL(head_gaussians_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, heads, M, H, W, beta,
                     perspective, flags, gaussians);
  const int64_t pixels = (int64_t)H * W;
  hipLaunchKernelGGL(density_label_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     gaussians, 2 * M, H, W, out);
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
