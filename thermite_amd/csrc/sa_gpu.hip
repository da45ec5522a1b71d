// sa_gpu.hip -- the suffix array of the index text on the GPU (offline index construction; the reference calls
// libdivsufsort64, src/index.rs:103-105).  Prefix doubling over rocPRIM's radix sort:
//
//   round 0   key = the first 8 bytes of the suffix (big-endian, zero behind the end of the text); sort (key, position)
//   ranks     rank of a suffix = index of the first entry of its group of equal keys (flags, inclusive max-scan,
//             scatter to text order)
//   round h   key = (rank[i], rank[i + h] + 1, or 0 behind the end): equal prefixes of length h are ordered by the h
//             symbols that follow; h doubles until every rank is distinct
//
// The order is that of csrc/sais.cpp -- plain byte order, a suffix that ends sorts before one that goes on -- and a
// text has one suffix array, so the result equals the host builder's (tests/test_gpu_index.py compares them; the
// library also accepts only verified arrays from outside).  Every round sorts all n pairs, no group bookkeeping:
// log2(longest repeat / 8) + 1 rounds of one 64-bit radix sort (a run of 118 M `N` makes it 25 rounds).  Measured:
// 93 M symbols 0.2 s, 600 M 1.6 s, 2.2 G 6.7 s with the transfers, against 7 s, 55 s and 190 s of single-threaded
// induced sorting on the host.  36 bytes of device memory per symbol while it runs (41 in the wide builder below, for
// texts of 2^32 - 2 symbols and more: 4.4 G symbols in 35 s); hosts without a device and devices without that much free
// memory take the host builder.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdlib>
#include <utility>

#include "thermite_internal.h"

namespace thm {
namespace {

__global__ void sa_first_keys(const uint8_t* text, uint64_t n, uint64_t* key, uint32_t* val) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    uint64_t k = 0;
    for (int b = 0; b < 8; b++) k = (k << 8) | (i + b < n ? (uint64_t)text[i + b] : 0ull);
    key[i] = k;
    val[i] = (uint32_t)i;
  }
}
// head[i] = i where a new group of equal keys starts, else 0; distinct += number of groups
__global__ void sa_group_heads(const uint64_t* key, uint64_t n, uint32_t* head, unsigned long long* distinct) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const bool h = i == 0 || key[i] != key[i - 1];
    head[i] = h ? (uint32_t)i : 0u;
    mine += h;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(distinct, mine);
}
__global__ void sa_scatter_ranks(const uint32_t* val, const uint32_t* rank_sorted, uint64_t n, uint32_t* rank_at) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) rank_at[val[i]] = rank_sorted[i];
}
__global__ void sa_next_keys(const uint32_t* rank_at, uint64_t n, uint64_t h, uint64_t* key, uint32_t* val) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const uint64_t second = i + h < n ? (uint64_t)rank_at[i + h] + 1 : 0ull;
    key[i] = ((uint64_t)rank_at[i] << 32) | second;
    val[i] = (uint32_t)i;
  }
}
__global__ void sa_widen(const uint32_t* in, uint64_t n, uint64_t* out) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) out[i] = in[i];
}

// ---- texts of 2^32 - 2 symbols and more: 64-bit positions, ranks of 33 bits and more.  The pair (rank[i], rank[i + h])
// no longer fits one 64-bit key, so a round is two stable sorts -- by the second component, then by the first.
__global__ void saw_first_keys(const uint8_t* text, uint64_t n, uint64_t* key, uint64_t* val) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    uint64_t k = 0;
    for (int b = 0; b < 8; b++) k = (k << 8) | (i + b < n ? (uint64_t)text[i + b] : 0ull);
    key[i] = k;
    val[i] = i;
  }
}
// head[i] = i where (a[i], b[i]) differs from the entry before (b may be null: one component), else 0
__global__ void saw_group_heads(const uint64_t* a, const uint64_t* b, uint64_t n, uint64_t* head, unsigned long long* distinct) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const bool h = i == 0 || a[i] != a[i - 1] || (b && b[i] != b[i - 1]);
    head[i] = h ? i : 0ull;
    mine += h;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(distinct, mine);
}
__global__ void saw_scatter_ranks(const uint64_t* pos, const uint64_t* rank_sorted, uint64_t n, uint64_t* rank_at) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) rank_at[pos[i]] = rank_sorted[i];
}
// key[i] = rank behind position pos[i] + h (pos == null: position i), 0 behind the end; val[i] = i when asked for
__global__ void saw_second_keys(const uint64_t* rank_at, const uint64_t* pos, uint64_t n, uint64_t h, uint64_t* key, uint64_t* val) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const uint64_t q = (pos ? pos[i] : i) + h;
    key[i] = q < n ? rank_at[q] + 1 : 0ull;
    if (val) val[i] = i;
  }
}
__global__ void saw_first_of(const uint64_t* rank_at, const uint64_t* pos, uint64_t n, uint64_t* key) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) key[i] = rank_at[pos[i]];
}

struct DeviceArrays {  // freed whatever way the builder leaves
  void* p[10] = {nullptr};
  int n = 0;
  template <class T>
  hipError_t alloc(T** q, size_t bytes) {
    const hipError_t e = hipMalloc((void**)q, bytes);
    if (e == hipSuccess) p[n++] = *q;
    return e;
  }
  ~DeviceArrays() {
    for (int i = 0; i < n; i++) (void)hipFree(p[i]);
  }
};

}  // namespace

// the wide builder: 64-bit entries out, 41 bytes of device memory per symbol (four work arrays and the ranks; the sorts
// ping-pong between the work arrays -- hipcub::DoubleBuffer -- instead of taking a copy of keys and values as scratch)
static int build_suffix_array_gpu_wide(const uint8_t* text, uint64_t n, uint64_t* out) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < n * 42 + (1024u << 20)) return -3;
#define SA_CK(x)                    \
  do {                              \
    if ((x) != hipSuccess) {        \
      (void)hipGetLastError();      \
      return -4;                    \
    }                               \
  } while (0)
  DeviceArrays mem;
  uint8_t* d_text = nullptr;
  uint64_t* w[4] = {nullptr, nullptr, nullptr, nullptr};  // work arrays: their roles change from step to step
  uint64_t* rank_at = nullptr;
  unsigned long long* d_distinct = nullptr;
  void* tmp = nullptr;
  SA_CK(mem.alloc(&d_text, n + 8));
  for (int i = 0; i < 4; i++) SA_CK(mem.alloc(&w[i], n * 8));
  SA_CK(mem.alloc(&rank_at, n * 8));
  SA_CK(mem.alloc(&d_distinct, 8));
  SA_CK(hipMemcpy(d_text, text, n, hipMemcpyHostToDevice));
  int rank_bits = 1;
  while (rank_bits < 64 && (n >> rank_bits) != 0) rank_bits++;
  rank_bits++;  // (ranks go up to n - 1, second components up to n)
  size_t sort_bytes = 0, scan_bytes = 0;
  {
    hipcub::DoubleBuffer<uint64_t> dk(w[0], w[1]), dv(w[2], w[3]);
    SA_CK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, n, 0, 64));
  }
  SA_CK(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, w[0], w[1], hipcub::Max(), n));
  const size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  SA_CK(mem.alloc((uint8_t**)&tmp, tmp_bytes));
  const dim3 grid(256 * 16), block(256);
  auto distinct_after = [&](const uint64_t* a, const uint64_t* b, uint64_t* head, unsigned long long& distinct) -> hipError_t {
    hipError_t e = hipMemsetAsync(d_distinct, 0, 8, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(saw_group_heads, grid, block, 0, 0, a, b, n, head, d_distinct);
    return hipMemcpy(&distinct, d_distinct, 8, hipMemcpyDeviceToHost);
  };
  // sort (keys in ka, values in va) with kb / vb as the other halves of the buffers: afterwards ka / va hold the result
  auto sort_pairs = [&](uint64_t*& ka, uint64_t*& kb, uint64_t*& va, uint64_t*& vb, int bits) -> hipError_t {
    hipcub::DoubleBuffer<uint64_t> dk(ka, kb), dv(va, vb);
    size_t sb = tmp_bytes;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, sb, dk, dv, n, 0, bits);
    ka = dk.Current();
    kb = dk.Alternate();
    va = dv.Current();
    vb = dv.Alternate();
    return e;
  };
  // round 0: by the first 8 bytes
  uint64_t *ka = w[0], *kb = w[1], *va = w[2], *vb = w[3];
  hipLaunchKernelGGL(saw_first_keys, grid, block, 0, 0, d_text, n, ka, va);
  SA_CK(sort_pairs(ka, kb, va, vb, 64));
  uint64_t* pos = va;    // positions in the current order
  uint64_t* heads = kb;  // group heads of the current order
  unsigned long long distinct = 0;
  SA_CK(distinct_after(ka, nullptr, heads, distinct));
  for (uint64_t h = 8; distinct != n; h *= 2) {
    if (h >= n) return -5;
    // ranks of the current order (max-scan of the heads), scattered to text order; after that only rank_at counts
    uint64_t* rs = vb;
    size_t sb = tmp_bytes;
    SA_CK(hipcub::DeviceScan::InclusiveScan(tmp, sb, heads, rs, hipcub::Max(), n));
    hipLaunchKernelGGL(saw_scatter_ranks, grid, block, 0, 0, pos, rs, n, rank_at);
    // A: by the second component, from text order
    hipLaunchKernelGGL(saw_second_keys, grid, block, 0, 0, rank_at, (const uint64_t*)nullptr, n, h, ka, va);
    SA_CK(sort_pairs(ka, kb, va, vb, rank_bits));  // ka: second components in order, va: their positions
    // B: stable, by the first component (the sorted second components are not needed any more: kb takes the keys)
    hipLaunchKernelGGL(saw_first_of, grid, block, 0, 0, rank_at, va, n, kb);
    std::swap(ka, kb);
    SA_CK(sort_pairs(ka, kb, va, vb, rank_bits));  // ka: first components in order, va: positions
    pos = va;
    // groups of equal (first, second): the second components once more, in the new order (-> kb), heads -> vb
    hipLaunchKernelGGL(saw_second_keys, grid, block, 0, 0, rank_at, pos, n, h, kb, (uint64_t*)nullptr);
    SA_CK(distinct_after(ka, kb, vb, distinct));
    // roles for the next round: heads in vb, positions in va; ka and kb are free (the scan writes into one of them)
    heads = vb;
    std::swap(kb, vb);  // now heads == kb, and vb (the old kb) is free for the scan's output
  }
  SA_CK(hipMemcpy(out, pos, n * 8, hipMemcpyDeviceToHost));
  SA_CK(hipDeviceSynchronize());
#undef SA_CK
  return 0;
}

// 0: out[0, n) holds the suffix array (entries of elem_bytes = 4 or 8 bytes).  Anything else: nothing was written that
// counts -- no device, not enough device memory, a text too long for 32-bit ranks, a HIP error -- the caller sorts on the host.
int build_suffix_array_gpu(const uint8_t* text, uint64_t n, void* out, int elem_bytes) {
  if (n == 0) return 0;
  if (elem_bytes != 4 && elem_bytes != 8) return -2;
  if (n >= 0xFFFFFFFEull && elem_bytes != 8) return -2;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    (void)hipGetLastError();
    return -1;
  }
  // (THM_SA_WIDE_SORT=1: the wide builder on a small text too -- how the tests reach it)
  if (n >= 0xFFFFFFFEull || (elem_bytes == 8 && getenv("THM_SA_WIDE_SORT"))) return build_suffix_array_gpu_wide(text, n, (uint64_t*)out);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < n * 38 + (256u << 20)) return -3;
#define SA_CK(x)                    \
  do {                              \
    if ((x) != hipSuccess) {        \
      (void)hipGetLastError();      \
      return -4;                    \
    }                               \
  } while (0)
  DeviceArrays mem;
  uint8_t* d_text = nullptr;
  uint64_t *k0 = nullptr, *k1 = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr, *rank_at = nullptr, *head = nullptr, *rs = nullptr;
  unsigned long long* d_distinct = nullptr;
  void* tmp = nullptr;
  SA_CK(mem.alloc(&d_text, n + 8));
  SA_CK(mem.alloc(&k0, n * 8));
  SA_CK(mem.alloc(&k1, n * 8));
  SA_CK(mem.alloc(&v0, n * 4));
  SA_CK(mem.alloc(&v1, n * 4));
  SA_CK(mem.alloc(&rank_at, n * 4));
  SA_CK(mem.alloc(&head, n * 4));
  SA_CK(mem.alloc(&rs, n * 4));
  SA_CK(mem.alloc(&d_distinct, 8));
  SA_CK(hipMemcpy(d_text, text, n, hipMemcpyHostToDevice));
  size_t sort_bytes = 0, scan_bytes = 0;
  SA_CK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k0, k1, v0, v1, n, 0, 64));
  SA_CK(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, head, rs, hipcub::Max(), n));
  const size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  SA_CK(mem.alloc((uint8_t**)&tmp, tmp_bytes));
  const dim3 grid(256 * 16), block(256);
  hipLaunchKernelGGL(sa_first_keys, grid, block, 0, 0, d_text, n, k0, v0);
  for (uint64_t h = 8;; h *= 2) {
    size_t sb = tmp_bytes;
    SA_CK(hipcub::DeviceRadixSort::SortPairs(tmp, sb, k0, k1, v0, v1, n, 0, 64));
    SA_CK(hipMemsetAsync(d_distinct, 0, 8, 0));
    hipLaunchKernelGGL(sa_group_heads, grid, block, 0, 0, k1, n, head, d_distinct);
    unsigned long long distinct = 0;
    SA_CK(hipMemcpy(&distinct, d_distinct, 8, hipMemcpyDeviceToHost));
    if (distinct == n) break;
    if (h >= n) return -5;  // (cannot happen: prefixes of n symbols are all distinct)
    sb = tmp_bytes;
    SA_CK(hipcub::DeviceScan::InclusiveScan(tmp, sb, head, rs, hipcub::Max(), n));
    hipLaunchKernelGGL(sa_scatter_ranks, grid, block, 0, 0, v1, rs, n, rank_at);
    hipLaunchKernelGGL(sa_next_keys, grid, block, 0, 0, rank_at, n, h, k0, v0);
  }
  if (elem_bytes == 4) {
    SA_CK(hipMemcpy(out, v1, n * 4, hipMemcpyDeviceToHost));
  } else {
    hipLaunchKernelGGL(sa_widen, grid, block, 0, 0, v1, n, k0);
    SA_CK(hipMemcpy(out, k0, n * 8, hipMemcpyDeviceToHost));
  }
  SA_CK(hipDeviceSynchronize());
#undef SA_CK
  return 0;
}

}  // namespace thm

// C ABI (include/thermite.h): the suffix array of `text` on the current HIP device
extern "C" int32_t thm_build_suffix_array_gpu(const uint8_t* text, uint64_t n, void* sa_out, uint32_t elem_bytes) {
  if ((!text && n) || !sa_out || (elem_bytes != 4 && elem_bytes != 8)) return THM_ERR_INVALID_ARG;
  const int rc = thm::build_suffix_array_gpu(text, n, sa_out, (int)elem_bytes);
  if (rc == 0) return THM_OK;
  if (rc == -1) {
    thm::set_global_error("thm_build_suffix_array_gpu: no HIP device");
    return THM_ERR_NO_DEVICE;
  }
  if (rc == -2) {
    thm::set_global_error("thm_build_suffix_array_gpu: a text of 2^32 - 2 symbols and more needs 8-byte entries");
    return THM_ERR_UNSUPPORTED;
  }
  if (rc == -3) {
    thm::set_global_error("thm_build_suffix_array_gpu: not enough free device memory (38 bytes per symbol)");
    return THM_ERR_OOM;
  }
  thm::set_global_error("thm_build_suffix_array_gpu: HIP error");
  return THM_ERR_INTERNAL;
}
