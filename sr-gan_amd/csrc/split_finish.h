// split_finish.h -- the ORDERED finish of a contraction whose K range is split over several workgroups.
//
// Until round 4 the K slices of a split launch met in the output through fp32 atomics (plus a zero-fill launch in front):
// the sum's order was the order in which the workgroups happened to arrive, two runs of one schedule differed at rounding
// level, and a pre-activation within rounding of zero then flipped its ReLU mask -- the gradient penalty of two runs moved
// by up to 6e-4 (profiles/r04h_*, gpurun_out/r5c: the drift disappears with every K split switched off).  The reference's
// CPU path is bitwise repeatable.
//
// Here every workgroup of an output tile leaves its partial accumulators in the caller's workspace (thread-linear layout:
// element i of thread t at [i * THREADS + t], so the stores and the later loads are coalesced and need no index math), waits
// until the stores are acknowledged, and takes a ticket of its tile; the workgroup that draws the LAST ticket re-reads all
// `splits` partials in slice order 0, 1, 2, ... and goes on to the kernel's ordinary epilogue with the total -- one launch,
// no zero-fill launch, no atomics on data, and the same bits on every run, whatever else is in flight.
//
// Memory model (gfx942 / gfx950, eight XCDs with an L2 each): the partials are written with agent-scope relaxed atomic
// stores (global_store_dword ... sc1: write-through) and read with agent-scope relaxed atomic loads (sc1); store and ticket
// live in different L2 channels, so the ticket's fetch_add must not be ISSUED before the stores are acknowledged:
// s_waitcnt vmcnt(0) in every thread, then the workgroup barrier, then thread 0 takes the ticket (ADVICE r4 on reduce.hip:
// a workgroup-scope release fence emits no instruction).  The tickets are device globals, one array per kernel family
// and one row per registered (device, stream) workspace -- launches on one stream are ordered, streams do not share a row --
// and the last workgroup puts its tile's ticket back to zero.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace srgan {

constexpr int SPLIT_TICKET_SETS = 64;       // registered (device, stream) workspaces that can use the ordered finish
constexpr int SPLIT_TICKET_TILES = 2048;    // output tiles per launch (split launches are the FEW-tile launches)

// COUNT accumulator floats per thread, THREADS threads per workgroup.  `get(i)` / `set(i, v)` access accumulator i.
// `ws_tile` = this output tile's region of the workspace: [splits][COUNT * THREADS] floats.  Returns true in the
// workgroup that holds the tile's total afterwards (all its threads), false in the others (which are done).
template <int COUNT, int THREADS, typename Get, typename Set>
__device__ __forceinline__ bool split_finish_ordered(float* ws_tile, int split, int splits, unsigned int* ticket, Get get, Set set) {
  __shared__ int split_finish_last;
  const int tid = (int)threadIdx.x;
  float* mine = ws_tile + (int64_t)split * (COUNT * THREADS) + tid;
#pragma unroll
  for (int i = 0; i < COUNT; ++i) __hip_atomic_store(mine + i * THREADS, get(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every partial of this thread is acknowledged ...
  __syncthreads();                                          // ... and so is every thread's, before the ticket is taken
  if (tid == 0)
    split_finish_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(splits - 1);
  __syncthreads();
  if (!split_finish_last) return false;
  // (slices outermost: the element loop unrolls completely -- the accumulators stay in registers -- and COUNT loads are in
  // flight per slice; every element is still summed in slice order 0, 1, 2, ...)
  const float* all = ws_tile + tid;
  float total[COUNT];
#pragma unroll
  for (int i = 0; i < COUNT; ++i) total[i] = __hip_atomic_load(all + i * THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int s = 1; s < splits; ++s) {
    const float* slice = all + (int64_t)s * (COUNT * THREADS);
#pragma unroll
    for (int i = 0; i < COUNT; ++i) total[i] += __hip_atomic_load(slice + i * THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int i = 0; i < COUNT; ++i) set(i, total[i]);
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// The same idea for a few SCALARS per workgroup that several workgroups add into one output row (the two parameter sums of
// a batch-norm backward per channel, a per-example loss sum).  Called by ALL 256 threads of the workgroup; thread 0 holds the
// workgroup's V values.  `partial_row` = the row's [parts][V] slots in the workspace, `ticket` = the row's ticket.  Returns true
// in thread 0 of the workgroup that drew the last ticket, with v = the sum over all parts: thread t adds the parts t, t + 256,
// ... and the 256 sums meet in the fixed tree of block_sum_256 -- the same order on every run, and no thread walks the parts
// alone (a single thread reading 768 parts one acknowledged load at a time cost 2 ms per launch).  The caller then adds v to
// the output: one adder per row and launch, no atomics on data.
template <int V>
__device__ __forceinline__ bool ordered_row_finish(float (&v)[V], float* partial_row, int part, int parts, unsigned int* ticket,
                                                   float* scratch4) {
  if (parts == 1) return threadIdx.x == 0;
  __shared__ int row_finish_last;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < V; ++i) __hip_atomic_store(partial_row + part * V + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    row_finish_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(parts - 1);
  }
  __syncthreads();
  if (!row_finish_last) return false;
  float mine[V];
#pragma unroll
  for (int i = 0; i < V; ++i) mine[i] = 0.f;
  for (int s = (int)threadIdx.x; s < parts; s += 256)
#pragma unroll
    for (int i = 0; i < V; ++i) mine[i] += __hip_atomic_load(partial_row + s * V + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    __syncthreads();                      // (scratch4 is reused)
    const float total = block_sum_256(mine[i], scratch4);
    if (threadIdx.x == 0) v[i] = total;
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return threadIdx.x == 0;
}

constexpr int ROW_FINISH_ROWS = 4096;       // output rows (channels / examples) per launch that can use ordered_row_finish

// Host side: the workspace region and the ticket row of a split launch, or nullptr when the ordered finish cannot be used
// (no workspace registered for the stream, too many tiles / bytes, or SRGAN_ATOMIC_SPLIT=1: the round-4 atomics).
float* partial_workspace(size_t bytes, hipStream_t stream);
int workspace_index(hipStream_t stream);

// The device address of a ticket array (per device: the symbol has one copy on each).  The cache is keyed by the SYMBOL'S
// ADDRESS, not by its type: several translation units declare arrays of one type (unsigned[64 * 2048]), and a cache per template
// instantiation handed whichever array was resolved first to all of them (benign while launches on a stream are
// ordered, wrong as a statement).  Guarded by a mutex: host threads may launch for the first time together.
template <typename Symbol>
inline unsigned int* device_tickets(const Symbol& symbol) {
  static std::mutex guard;
  static std::map<std::pair<const void*, int>, unsigned int*> cache;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(guard);
  const std::pair<const void*, int> key{static_cast<const void*>(&symbol), device};
  auto found = cache.find(key);
  if (found != cache.end()) return found->second;
  void* address = nullptr;
  if (hipGetSymbolAddress(&address, HIP_SYMBOL(symbol)) != hipSuccess) return nullptr;
  return cache[key] = static_cast<unsigned int*>(address);
}

inline bool split_atomics_forced() {
  static const bool forced = getenv("SRGAN_ATOMIC_SPLIT") != nullptr;
  return forced;
}

// `offset_bytes`: the part of the workspace the launch already uses for something else (packed weights).
inline float* split_workspace(int64_t tiles, int splits, int64_t floats_per_tile_and_split, size_t offset_bytes, hipStream_t stream,
                              int* ticket_set) {
  if (split_atomics_forced() || tiles > SPLIT_TICKET_TILES || splits < 2) return nullptr;
  const int set = workspace_index(stream);
  if (set < 0 || set >= SPLIT_TICKET_SETS) return nullptr;
  offset_bytes = (offset_bytes + 255) & ~(size_t)255;
  const size_t bytes = offset_bytes + (size_t)tiles * splits * floats_per_tile_and_split * sizeof(float);
  float* base = partial_workspace(bytes, stream);
  if (!base) return nullptr;
  *ticket_set = set;
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + offset_bytes);
}

// Host side of ordered_row_finish: the [rows][parts][V] region and the launch's ticket row (tickets: a device array of
// SPLIT_TICKET_SETS * ROW_FINISH_ROWS of the caller's translation unit), or nullptr (no workspace / too many rows / atomics forced).
template <typename Symbol>
inline float* row_finish_workspace(int64_t rows, int parts, int values, const Symbol& tickets, hipStream_t stream, unsigned int** ticket_row) {
  if (split_atomics_forced() || rows > ROW_FINISH_ROWS || parts < 1) return nullptr;
  const int set = workspace_index(stream);
  if (set < 0 || set >= SPLIT_TICKET_SETS) return nullptr;
  float* base = partial_workspace((size_t)rows * parts * values * sizeof(float), stream);
  unsigned int* all = base ? device_tickets(tickets) : nullptr;
  if (!base || !all) return nullptr;
  *ticket_row = all + (size_t)set * ROW_FINISH_ROWS;
  return base;
}

}  // namespace srgan
