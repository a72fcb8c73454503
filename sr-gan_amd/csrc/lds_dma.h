// lds_dma.h -- global -> LDS loads that bypass the registers (global_load_lds_*), used by the kernels that stage MFMA
// operands asynchronously (pointwise_ring.hip, pointwise_wgrad.hip).
#pragma once
#include "common.h"

namespace srgan {

// 64 lanes x 16 bytes from (scalar base + per-lane 32-bit byte offset) to LDS at the wave-uniform byte address `lds_dst` +
// lane * 16.  M0 carries the LDS base and is compiler-reserved: saved and restored inside the statement; the s_nop is the
// wait state between the M0 write and the DMA (guide 5.7).
__device__ __forceinline__ void ring_glds16(const void* base, uint32_t lane_byte_offset, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_offset), "s"(base), "s"(lds_dst) : "memory");
}
// 64 lanes x 4 bytes from per-lane 64-bit addresses (the four batch-norm vectors of a stage into one table).
__device__ __forceinline__ void ring_glds4(const float* lane_pointer, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_pointer), "s"(lds_dst) : "memory");
}
// This wave's DMAs down to the N youngest have landed and its LDS reads are done; then the workgroup barrier (a
// __syncthreads() would drain the whole DMA queue).
template <int N> __device__ __forceinline__ void ring_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}
__device__ __forceinline__ uint32_t ring_lds_address(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace srgan
