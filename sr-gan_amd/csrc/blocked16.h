// blocked16.h -- the 16-bit data path (BASELINE.json configs[1] "bf16" and configs[4] "fp16"): what the kernels of
// blocked16*.hip share.
//
// Layout in HBM ("blocked"): a tensor of logical shape [N, C, H, W] is stored as [N][ceil(C / 8)][H][W][8] elements of
// bf16 (dtype 1) or fp16 (dtype 2): the 8 channels of a group at one pixel are ONE 16-byte slot.  That slot is exactly the
// k = 8 operand fragment of v_mfma_f32_32x32x16_{bf16,f16} when the reduction runs over channels (forward convolution, data
// gradient, linear layers): a workgroup copies slots from HBM to LDS and a lane's fragment is one ds_read_b128 -- no
// conversion, no gather (the fp32-storage mixed kernels of conv3x3.hip spend eight 4-byte loads and eight conversions per
// slot).  When the reduction runs over PIXELS (weight gradients) the same LDS image is read with ds_read_b64_tr_b16, the
// hardware's 16-bit transpose read: 16 lanes fetch a 4-pixel x 16-channel block and every lane receives four consecutive
// pixels of its own channel.  Channels beyond C inside the last group are stored as zeros (the packed weights are zero
// there too, so they never contribute).  A [N, F] matrix is the H = W = 1 case: plain row-major.
#pragma once
#include <type_traits>
#include "common.h"

namespace srgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 h_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h_f16x2 __attribute__((ext_vector_type(2)));
typedef short h_s4 __attribute__((ext_vector_type(4)));

struct alignas(16) Slot { uint32_t v[4]; };      // 8 x 16-bit: channels 8g .. 8g + 7 of one pixel

// two floats -> one packed pair (a in the low half), round to nearest even
template <int PREC>
__device__ __forceinline__ uint32_t h_pack2(float a, float b) {
  if constexpr (PREC == 1) {
    h_bf16x2 h; h[0] = (__bf16)a; h[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, h);
  } else {
    h_f16x2 h; h[0] = (_Float16)a; h[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, h);
  }
}
template <int PREC>
__device__ __forceinline__ float h_lo(uint32_t u) {
  if constexpr (PREC == 1) return __uint_as_float(u << 16);
  else return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFFu));
}
template <int PREC>
__device__ __forceinline__ float h_hi(uint32_t u) {
  if constexpr (PREC == 1) return __uint_as_float(u & 0xFFFF0000u);
  else return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
}
// [value > 0] of a 16-bit float given its bits (bf16 and fp16 alike: sign clear, not zero; NaN counts as positive)
__device__ __forceinline__ bool h_positive(uint32_t bits16) { return (int16_t)(uint16_t)bits16 > 0; }

template <int PREC>
__device__ __forceinline__ void h_unpack8(const Slot& s, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = h_lo<PREC>(s.v[i]); f[2 * i + 1] = h_hi<PREC>(s.v[i]); }
}
template <int PREC>
__device__ __forceinline__ Slot h_pack8(const float (&f)[8]) {
  Slot s;
#pragma unroll
  for (int i = 0; i < 4; ++i) s.v[i] = h_pack2<PREC>(f[2 * i], f[2 * i + 1]);
  return s;
}
// Channels per 16-byte slot: 8 for the 16-bit types; dtype 0 = fp32 in the same blocked layout with FOUR channels per slot
// ([N][ceil(C / 4)][H][W][4] floats): the exact path of the 4x4 / stride 2 family (blocked16_k4s2.hip) -- the fp32 gradient-
// penalty chain of the fp16 configuration and the crowd generator -- where v_mfma_f32_32x32x2_f32 takes one float per lane
// and a slot feeds four matrix instructions.
template <int PREC> struct HGroup { static constexpr int N = PREC == 0 ? 4 : 8; };
template <int PREC>
__device__ __forceinline__ void h_unpack(const Slot& s, float (&f)[HGroup<PREC>::N]) {
  if constexpr (PREC == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(s.v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = h_lo<PREC>(s.v[i]); f[2 * i + 1] = h_hi<PREC>(s.v[i]); }
  }
}
template <int PREC>
__device__ __forceinline__ Slot h_pack(const float (&f)[HGroup<PREC>::N]) {
  Slot s;
  if constexpr (PREC == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = __float_as_uint(f[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = h_pack2<PREC>(f[2 * i], f[2 * i + 1]);
  }
  return s;
}
// [element j of the slot > 0] for the mask by reference
template <int PREC>
__device__ __forceinline__ bool h_slot_positive(const Slot& s, int j) {
  if constexpr (PREC == 0) return __uint_as_float(s.v[j]) > 0.f;
  else return h_positive((s.v[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
}

// The derivative of relu (slope 0) / leaky_relu / identity (slope 1) from the sign pattern of the ACTIVATED value:
// act(z) > 0 <=> z > 0 for slope >= 0, so the 16-bit output itself is the mask (reference: torch's threshold / leaky_relu
// backward use x > 0).
__device__ __forceinline__ float h_mask(uint32_t bits16, float slope) { return h_positive(bits16) ? 1.f : slope; }

// ds_read_b64_tr_b16: within every group of 16 lanes, lane t (0..15) supplies the address of 4 consecutive 16-bit elements
// and RECEIVES element (t & 3) of the chunks supplied by lanes 4j + (t >> 2), j = 0..3 (guide: "lane l, elem j reads
// lds[(l & 15) + j * 16 + (l >> 4) * 64]" for chunks laid out lane-linearly).  Pointed at blocked slots -- lane s = 4j + u
// supplies pixel p0 + j, channels 4u .. 4u + 3 of a 16-channel block (u >> 1 picks the group, u & 1 the half of the slot) --
// lane t receives pixels p0 .. p0 + 3 of channel t: four consecutive k of a pixel-reduction operand.
__device__ __forceinline__ uint2 h_tr_read(uint32_t lds_byte_address) {
  const h_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h_s4*)(uintptr_t)lds_byte_address);
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ uint32_t h_lds_address(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int PREC>
__device__ __forceinline__ f32x16 h_mfma(const Slot& a, const Slot& b, f32x16 c) {
  if constexpr (PREC == 0) {           // fp32: the slot's four channels are four k-steps of v_mfma_f32_32x32x2_f32 (k pair = the
#pragma unroll                         // same component of the two lane halves' slots)
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.v[i]), __uint_as_float(b.v[i]), c, 0, 0, 0);
    return c;
  } else if constexpr (PREC == 1)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h_bf16x8, a), __builtin_bit_cast(h_bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h_f16x8, a), __builtin_bit_cast(h_f16x8, b), c, 0, 0, 0);
}

// LDS-DMA: 64 lanes x 16 bytes from PER-LANE global addresses to LDS at the wave-uniform byte address `lds_dst` + lane * 16
// (global_load_lds_dwordx4; M0 carries the LDS base and is compiler-reserved: saved and restored inside the statement).
__device__ __forceinline__ void h_glds16(const void* lane_pointer, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_pointer), "s"(lds_dst) : "memory");
}
// This wave's DMAs down to the N youngest have landed and its LDS reads are done; then the workgroup barrier (raw: a
// __syncthreads() would drain the whole DMA queue).
template <int N> __device__ __forceinline__ void h_dma_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}
// ---- weight-shadow packers: the bodies (one thread = one slot) are shared by the single-layer kernels and by the batched
// launch that re-rounds every convolution shadow of a network behind its optimizer update (h_pack_batched_kernel, blocked16.hip)
constexpr int K4_CK = 4;          // k-slots per chunk of the 4x4 / stride 2 family (blocked16_k4s2.hip)
constexpr int K4_TAPS = 4;

// conv weights (R x S taps) as the operand of the LDS-staged kernels: slot (chunk, tap, half g, row o) holds the 8 reduced
// channels chunk * 16 + g * 8 .. + 7 of row o at that tap.
template <int PREC>
__device__ __forceinline__ void h_pack_conv_weights_slot(const float* __restrict__ w, Slot* __restrict__ packed, int64_t slot,
                                                         int32_t CO, int32_t CI, int32_t R, int32_t S, int32_t base, int32_t so,
                                                         int32_t si, int32_t skh, int32_t skw) {
  const int T = R * S;
  const int o = (int)(slot % CO);
  const int64_t rest = slot / CO;
  const int g = (int)(rest & 1);
  const int64_t ct = rest >> 1;
  const int tap = (int)(ct % T), chunk = (int)(ct / T);
  const int kh = tap / S, kw = tap - kh * S;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = chunk * 16 + g * 8 + j;
    v[j] = c < CI ? w[base + o * so + c * si + kh * skh + kw * skw] : 0.f;
  }
  packed[slot] = h_pack8<PREC>(v);
}

// 4x4 / stride 2 weights as 2x2-tap operands (mode 0: "down", 1 .. 4: the "up" parity classes; layout in blocked16_k4s2.hip)
template <int PREC>
__device__ __forceinline__ void h_pack_k4s2_weights_slot(const float* __restrict__ w, Slot* __restrict__ packed, int64_t slot,
                                                         int32_t rows, int32_t reduced, int64_t row_stride, int64_t reduced_stride,
                                                         int32_t mode) {
  const int o = (int)(slot % rows);
  int64_t rest = slot / rows;
  const int j = (int)(rest % K4_CK); rest /= K4_CK;
  const int tap = (int)(rest % K4_TAPS);
  const int chunk = (int)(rest / K4_TAPS);
  const int a = tap >> 1, b = tap & 1;
  constexpr int G = HGroup<PREC>::N;
  int kh, kw, first;
  if (mode == 0) { kh = 2 * a + (j >> 1); kw = 2 * b + (j & 1); first = G * chunk; }
  else {
    const int ry = (mode - 1) >> 1, rx = (mode - 1) & 1;
    kh = ry == 0 ? 3 - 2 * a : 2 - 2 * a;
    kw = rx == 0 ? 3 - 2 * b : 2 - 2 * b;
    first = G * (K4_CK * chunk + j);
  }
  float v[G];
#pragma unroll
  for (int i = 0; i < G; ++i)
    v[i] = first + i < reduced ? w[o * row_stride + (int64_t)(first + i) * reduced_stride + kh * 4 + kw] : 0.f;
  packed[slot] = h_pack<PREC>(v);
}

// one problem of the batched launch (80 bytes, filled on the host by srgan_h_pack_job_*)
struct HPackJob {
  const float* w; Slot* packed;
  int64_t slots, first_block;       // this job's workgroups are [first_block, first_block + ceil(slots / 256))
  int32_t kind, prec;               // kind 0: conv weights, 1: 4x4 / stride 2 weights; prec = dtype
  int32_t p[10];                    // the body's integer arguments, in its order
};

const struct Slot* h_zero_slots();      // 64 bytes of zeros in device memory (blocked16.hip): the source of padding slots

// Declared in gather_gemm_kernels.hip: the bench's live event bracket around a contraction launch, with explicit bytes.
int profile_bracket_begin(hipStream_t stream);
int profile_bracket_end_bytes(int slot, hipStream_t stream, int64_t M, int64_t N, int64_t K, int kind, int bm, int bn, int split,
                              double bytes, int precision);

}  // namespace srgan
