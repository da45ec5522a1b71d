// kernels_util.hip -- small plumbing kernels: exclusive prefix sums that turn
// per-read counts into offsets (count -> scan -> fill), u32 -> u64 widening.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace thm {
namespace dev {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// exclusive scan of `v` over the block (256 threads); returns the thread's prefix, total in *tot
__device__ uint64_t block_excl_scan(uint64_t v, uint64_t* sh, uint64_t* tot) {
  const int t = (int)threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 1; o < SCAN_THREADS; o <<= 1) {
    uint64_t a = (t >= o) ? sh[t - o] : 0;
    __syncthreads();
    sh[t] += a;
    __syncthreads();
  }
  const uint64_t incl = sh[t];
  *tot = sh[SCAN_THREADS - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const uint64_t* in, uint64_t* out, uint64_t n,
                                                                  uint64_t* tile_sums) {
  __shared__ uint64_t sh[SCAN_THREADS];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  uint64_t tot;
  uint64_t pre = block_excl_scan(s, sh, &tot);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of the tile sums in place, grand total to *total_out
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint64_t* tile_sums, uint64_t n_tiles,
                                                                 uint64_t* total_out) {
  __shared__ uint64_t sh[SCAN_THREADS];
  uint64_t carry = 0;
  for (uint64_t b0 = 0; b0 < n_tiles; b0 += SCAN_THREADS) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = (i < n_tiles) ? tile_sums[i] : 0;
    uint64_t tot;
    const uint64_t pre = block_excl_scan(v, sh, &tot);
    if (i < n_tiles) tile_sums[i] = carry + pre;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(uint64_t* out, uint64_t n, const uint64_t* tile_offs) {
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  const uint64_t add = tile_offs[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (base + k < n) out[base + k] += add;
}

__global__ void widen_kernel(const uint32_t* in, uint64_t* out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// Calibration of the memory-side counters (rocprofv3 FETCH_SIZE) on a gather of known size in the access patterns
// of the two hot stages (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"):
//   pattern 0  every thread loads 8 aligned bytes at a pseudo-random offset of the table (index probes);
//   pattern 1  every thread loads 4 aligned bytes at a pseudo-random offset (suffix-array entries);
//   pattern 2  groups of 16 lanes load 256 contiguous bytes (16 B per lane) at a pseudo-random 16-byte-aligned
//              offset (window staging of the extend kernel).
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
__global__ __launch_bounds__(256) void calib_gather_kernel(const uint8_t* table, uint64_t span, uint64_t n_threads, int pattern,
                                                           unsigned long long* sink) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= n_threads) return;
  unsigned long long acc = 0;
  if (pattern == 0) {
    const uint64_t o = (mix64(tid) % (span / 8)) * 8;
    acc = *(const unsigned long long*)(table + o);
  } else if (pattern == 1) {
    const uint64_t o = (mix64(tid) % (span / 4)) * 4;
    acc = *(const unsigned*)(table + o);
  } else {
    const uint64_t o = (mix64(tid >> 4) % ((span - 256) / 16)) * 16 + (tid & 15) * 16;
    const uint4 v = *(const uint4*)(table + o);
    acc = v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x7f4a7c15ull) *sink = acc;  // keeps the loads alive (a value all three patterns can produce)
}

}  // namespace dev

hipError_t launch_calib_gather(const uint8_t* table, uint64_t span, uint64_t n_threads, int pattern, unsigned long long* sink,
                               hipStream_t s) {
  if (n_threads == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::calib_gather_kernel, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, s, table, span, n_threads,
                     pattern, sink);
  return hipGetLastError();
}

size_t scan_tmp_entries(uint64_t n) { return (size_t)((n + dev::SCAN_TILE - 1) / dev::SCAN_TILE) + 1; }

// out has n+1 entries: out[i] = sum(in[0..i)), out[n] = total
hipError_t launch_exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s) {
  if (n == 0) return hipMemsetAsync(out, 0, 8, s);
  const uint64_t tiles = (n + dev::SCAN_TILE - 1) / dev::SCAN_TILE;
  hipLaunchKernelGGL(dev::scan_tiles_kernel, dim3((unsigned)tiles), dim3(dev::SCAN_THREADS), 0, s, in, out, n, tmp);
  hipLaunchKernelGGL(dev::scan_sums_kernel, dim3(1), dim3(dev::SCAN_THREADS), 0, s, tmp, tiles, out + n);
  hipLaunchKernelGGL(dev::scan_add_kernel, dim3((unsigned)tiles), dim3(dev::SCAN_THREADS), 0, s, out, n, tmp);
  return hipGetLastError();
}

hipError_t launch_widen_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

}  // namespace thm
