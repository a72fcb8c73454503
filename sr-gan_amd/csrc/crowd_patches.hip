// crowd_patches.hip -- training-batch assembly of the crowd application ON the device: for every
// example of the batch, the P x P patch around a given centre of a full scene that already lives in HBM as uint8 RGB
// (H, W, 3) + float label / map (H, W), optionally mirrored left-right, the image normalised to [-1, 1] and laid out
// planar.  This is ExtractPatchForPosition(allow_padded) -> RandomHorizontalFlip -> NegativeOneToOneNormalizeImage ->
// NumpyArraysToTorchTensors of the reference's 4-worker NumPy pipeline (crowd/data.py:41-128,370-453,
// crowd/shanghai_tech_data.py:76-104) in one HBM-bound kernel: 3 bytes + 8 bytes read, 20 bytes written per pixel.
// Pixels outside the scene are zero BEFORE normalisation (the reference pads the uint8 image), i.e. -1 afterwards.
#include "common.h"

namespace srgan {

__global__ __launch_bounds__(256) void crowd_patches_kernel(const uint8_t* const* __restrict__ images,
                                                            const float* const* __restrict__ labels,
                                                            const float* const* __restrict__ maps,
                                                            const int32_t* __restrict__ heights,
                                                            const int32_t* __restrict__ widths,
                                                            const int32_t* __restrict__ ys, const int32_t* __restrict__ xs,
                                                            const int32_t* __restrict__ flips, int P,
                                                            float* __restrict__ out_images, float* __restrict__ out_labels,
                                                            float* __restrict__ out_maps) {
  const int b = blockIdx.y, row = blockIdx.x;
  const int H = heights[b], W = widths[b], half = P / 2;
  const int src_row = ys[b] - half + row;
  const bool row_ok = (unsigned)src_row < (unsigned)H;
  const bool flip = flips[b] != 0;
  const uint8_t* image = images[b];
  const float* label = labels ? labels[b] : nullptr;
  const float* map = maps ? maps[b] : nullptr;
  const int64_t plane = (int64_t)P * P;
  float* oi = out_images + (int64_t)b * 3 * plane + (int64_t)row * P;
  float* ol = out_labels ? out_labels + (int64_t)b * plane + (int64_t)row * P : nullptr;
  float* om = out_maps ? out_maps + (int64_t)b * plane + (int64_t)row * P : nullptr;
  for (int col = threadIdx.x; col < P; col += 256) {
    const int src_col = xs[b] - half + (flip ? P - 1 - col : col);
    const bool ok = row_ok && (unsigned)src_col < (unsigned)W;
    const int64_t at = ok ? (int64_t)src_row * W + src_col : 0;
    float r = 0.f, g = 0.f, bl = 0.f;
    if (ok) { r = (float)image[at * 3]; g = (float)image[at * 3 + 1]; bl = (float)image[at * 3 + 2]; }
    oi[col] = r / 127.5f - 1.f;
    oi[plane + col] = g / 127.5f - 1.f;
    oi[2 * plane + col] = bl / 127.5f - 1.f;
    if (ol) ol[col] = (ok && label) ? label[at] : 0.f;
    if (om) om[col] = (ok && map) ? map[at] : 0.f;
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" int srgan_crowd_extract_patches(const void* const* images_u8, const float* const* labels,
                                           const float* const* maps, const int32_t* heights, const int32_t* widths,
                                           const int32_t* ys, const int32_t* xs, const int32_t* flips, int32_t B,
                                           int32_t P, float* out_images, float* out_labels, float* out_maps, void* stream) {
  SRGAN_REQUIRE(images_u8 && heights && widths && ys && xs && flips && out_images && B > 0 && P > 0 && P % 2 == 0 &&
                B <= 65535, SRGAN_EINVAL, "srgan_crowd_extract_patches arguments");
  hipLaunchKernelGGL(crowd_patches_kernel, dim3(P, B), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint8_t* const*>(images_u8), labels, maps, heights, widths, ys, xs, flips, P,
                     out_images, out_labels, out_maps);
  return launch_status();
}
