// kernels_seed.hip -- seed lookup: all supermaximal exact matches (SMEMs) of
// length >= min_seed_len of each read against the both-strand text, i.e. the
// result of Index::all_smems (reference src/index.rs:228-255).
//
// The reference walks an FMD index (bio 0.37.1, ~2L dependent Occ lookups per
// read).  Here a read position is searched by one thread (ms_search):
//     kt-mer prefix table  ->  suffix-array interval  ->  refine by binary search
//     on (sa, text)  ->  once one suffix is left, 8-byte compares along the text.
// That yields the matching statistic MS[i]; position i starts an SMEM iff
// i + MS[i] > (i-1) + MS[i-1], and only matches of
// length >= k are needed, so positions whose kt-mer is absent stop after one
// table probe.  Because the match end i + MS[i] never decreases with i, most
// positions need no probe at all: the kernels below probe position 0, then a
// stride-8 grid, then only the grid cells around a jump of the end (work lists
// keep those launches dense).  The SMEMs are then put in the order the reference's
// `mems.sort_by_key(len); mems.reverse()` produces:
// length descending, ties by reverse emission order of FMDIndex::all_smems;
// occurrences of one SMEM are sa[hi-1], ..., sa[lo] (descending rank).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "launch.h"
#include "swg_device.h"

namespace thm {
namespace dev {

__device__ __forceinline__ int base_code(uint8_t c) {
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}
// upper-case (reference src/aligner.rs:125) and map every byte that cannot
// occur in the text to 0, which matches nothing (text symbols are $ACGNT and a
// read never legitimately holds '$')
__device__ __forceinline__ uint8_t sanitize_base(uint8_t c) {
  if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
  return (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') ? c : (uint8_t)0;
}
__device__ __forceinline__ uint64_t load8_global(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);  // one global_load_dwordx2; gfx950 serves unaligned global loads
  return v;
}
// 8 bytes from an LDS byte array at any offset (two aligned reads + funnel shift)
__device__ __forceinline__ uint64_t load8_lds(const uint8_t* base16, int off) {
  const uint64_t* q = (const uint64_t*)(base16 + (off & ~7));
  const unsigned sh = (unsigned)(off & 7) * 8u;
  const uint64_t w0 = q[0], w1 = q[1];
  return sh ? ((w0 >> sh) | (w1 << (64u - sh))) : w0;
}

// LCP of the query tail q[0..cap) with the text at tp; *less tells whether the
// text suffix sorts before the query (a suffix that has the whole query tail as a
// prefix is >= it).  The text bytes are fetched 64 at a time with all eight loads
// in flight together (one memory round trip per 64 characters instead of one per
// 8); PROBE first looks at 8 bytes only, which settles most comparisons of a
// binary search.  Text and read buffers carry 128 bytes of padding, so the
// speculative tail of a batch stays inside the allocations.
template <bool PROBE>
__device__ __forceinline__ int lcp_cmp(const uint8_t* tp, const uint8_t* q, int cap, bool* less) {
  *less = false;
  int o = 0;
  if (PROBE && cap > 0) {
    const uint64_t tw = load8_global(tp), qw = load8_global(q);
    const uint64_t x = tw ^ qw;
    if (x) {
      const int idx = __builtin_ctzll(x) >> 3;
      if (idx < cap) {
        *less = ((tw >> (8 * idx)) & 0xff) < ((qw >> (8 * idx)) & 0xff);
        return idx;
      }
      return cap;
    }
    o = 8;
  }
  while (o < cap) {
    uint64_t tw[8];
#pragma unroll
    for (int u = 0; u < 8; u++) tw[u] = load8_global(tp + o + 8 * u);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int oo = o + 8 * u;
      if (oo < cap) {
        const uint64_t qw = load8_global(q + oo);
        const uint64_t x = tw[u] ^ qw;
        if (x) {
          const int idx = __builtin_ctzll(x) >> 3;
          if (oo + idx < cap) {
            *less = ((tw[u] >> (8 * idx)) & 0xff) < ((qw >> (8 * idx)) & 0xff);
            return oo + idx;
          }
          return cap;
        }
      }
    }
    o += 64;
  }
  return cap;
}

// longest match of rd[pos..L) in the text: length d and suffix-array interval.
//   1. kt-mer table: interval of the first kt characters in one probe;
//   2. one suffix left -> compare along the text;
//   3. up to 8 suffixes: fetch all their positions, then all their next 8 bytes,
//      with the loads in flight together; only suffixes that agree on those 8 bytes
//      are compared further.  The interval of the longest match is the run of
//      suffixes attaining the maximum (they are adjacent in suffix order);
//   4. larger intervals (repeats): lower bound of the whole query tail by binary
//      search, the better of the two neighbours of the insertion point gives the
//      match length, two more binary searches give the interval.
template <class C>
__device__ void ms_search(const DeviceIndexT<C>& ix, const uint8_t* rd, int L, int pos, int k, int& out_d, C& out_lo, C& out_hi) {
  C lo = 0, hi = (C)ix.n;
  int d = 0;
  const int kt = (int)ix.kt;
  if (kt <= k) {
    uint32_t code = 0;
    bool acgt = true;
    const uint64_t w0 = load8_global(rd + pos), w1 = load8_global(rd + pos + 8);  // kt <= 14 characters
    for (int t = 0; t < kt; t++) {
      const int c = base_code((uint8_t)((t < 8 ? (w0 >> (8 * t)) : (w1 >> (8 * (t - 8)))) & 0xff));
      acgt = acgt && (c >= 0);
      code = (code << 2) | (uint32_t)(c & 3);
    }
    if (acgt) {
      const LutEntryT<C> e = ix.lut[code];
      lo = e.lo;
      hi = e.hi;
      d = kt;
    }
  }
  if (lo < hi && pos + d < L) {
    const int cap = L - (pos + d);
    const uint8_t* q = rd + pos + d;
    bool less;
    const C sz = hi - lo;
    if (sz == 1) {
      d += lcp_cmp<false>(ix.text + ix.sa[lo] + d, q, cap, &less);
    } else if (q[0] != 0) {  // a byte outside ACGTN matches nothing: the interval stays at depth d
      if (sz <= 8) {
        C sav[8];
        uint64_t tw[8];
        int lc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) sav[u] = (lo + u < hi) ? ix.sa[lo + u] : (C)0;
#pragma unroll
        for (int u = 0; u < 8; u++) tw[u] = load8_global(ix.text + sav[u] + d);
        const uint64_t qw = load8_global(q);
        int ms = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const uint64_t x = tw[u] ^ qw;
          int l = x ? (__builtin_ctzll(x) >> 3) : 8;
          l = min(l, cap);
          if (lo + u >= hi) l = -1;
          lc[u] = l;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (lc[u] == 8 && cap > 8) lc[u] = 8 + lcp_cmp<false>(ix.text + sav[u] + d + 8, q + 8, cap - 8, &less);
          ms = max(ms, lc[u]);
        }
        if (ms > 0) {
          C nlo = hi, nhi = lo;
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (lc[u] == ms) {
              nlo = min(nlo, (C)(lo + u));
              nhi = max(nhi, (C)(lo + u + 1));
            }
          }
          lo = nlo;
          hi = nhi;
          d += ms;
        }
      } else {
        C a = lo, b = hi;
        while (a < b) {
          const C m = a + ((b - a) >> 1);
          (void)lcp_cmp<true>(ix.text + ix.sa[m] + d, q, cap, &less);
          if (less)
            a = m + 1;
          else
            b = m;
        }
        int l1 = -1, l2 = -1;
        if (a > lo) l1 = lcp_cmp<true>(ix.text + ix.sa[a - 1] + d, q, cap, &less);
        if (a < hi) l2 = lcp_cmp<true>(ix.text + ix.sa[a] + d, q, cap, &less);
        const int ms = max(l1, l2);
        if (ms > 0) {
          C nlo = a, nhi = a;
          if (l1 >= ms) {  // leftmost suffix in [lo, a) that still shares ms characters
            C x = lo, y = a - 1;
            while (x < y) {
              const C m = x + ((y - x) >> 1);
              if (lcp_cmp<true>(ix.text + ix.sa[m] + d, q, ms, &less) >= ms)
                y = m;
              else
                x = m + 1;
            }
            nlo = x;
          }
          if (l2 >= ms) {  // one past the rightmost suffix in [a, hi) that shares ms characters
            C x = a + 1, y = hi;
            while (x < y) {
              const C m = x + ((y - x) >> 1);
              if (lcp_cmp<true>(ix.text + ix.sa[m] + d, q, ms, &less) >= ms)
                x = m + 1;
              else
                y = m;
            }
            nhi = x;
          }
          lo = nlo;
          hi = nhi;
          d += ms;
        }
      }
    }
  }
  out_d = (lo < hi) ? d : 0;
  out_lo = lo;
  out_hi = hi;
}

// upper-case + sanitise the batch once (reference src/aligner.rs:125); both the
// probe kernel and the extend kernel read this copy
__global__ void sanitize_kernel(const uint8_t* in, uint8_t* out, uint64_t n, uint64_t n_padded) {
  // 16 bytes per thread (both buffers come from hipMalloc and are padded past n_padded)
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= n_padded) return;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (i < n) v = *(const uint4*)(in + i);  // may read up to 15 bytes past n: inside the allocation's slack
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint64_t at = i + (uint64_t)(4 * j + b);
      const uint8_t c = (at < n) ? sanitize_base((uint8_t)(w[j] >> (8 * b))) : (uint8_t)0;
      r |= (uint32_t)c << (8 * b);
    }
    w[j] = r;
  }
  *(uint4*)(out + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// Matching statistics by probing fewer and fewer positions.  E[i] = i + MS[i], the end of
// the longest match from i, never decreases with i (MS[i+1] >= MS[i] - 1), and a position
// starts an SMEM iff MS[i] >= k and E[i] > E[i-1]:
//   seed_first_kernel  position 0 of every read.  If that match spans the read (the common
//                      case for error-free reads) it is the read's only SMEM: it is emitted
//                      here and the read leaves the seed stage after one probe.  The other
//                      reads go on a work list (one for reads of at most SHORT_READ_MAX bases,
//                      one for longer reads: a few long reads in a batch of short ones must not
//                      size the launches of the short ones).
//   seed_grid_kernel   every PROBE_STRIDE-th position and the last one, for listed reads.
//   seed_cells_kernel  one thread per grid cell (the positions between two grid points): when
//                      both ends of the cell have MS >= k and the same end, every position
//                      inside has that end too (and MS >= k), so nothing starts there and
//                      the end is recorded without probing; other cells go on a work list.
//   seed_fill_kernel   the inner positions of listed cells: the cells around a jump of E.
// The work lists keep the probing launches dense (whole waves of real probes).  The per-position
// arrays are ragged: the row of read r starts at slot ms_row(offsets[r], r) (launch.h).
constexpr int PROBE_STRIDE = 8;

// append `value` to a list for every thread with `flag`: one atomic per 256-thread workgroup
// (every thread must call it; the trailing barrier makes back-to-back calls safe)
__device__ __forceinline__ void block_append(bool flag, unsigned long long value, unsigned long long* list,
                                             unsigned long long* count) {
  __shared__ unsigned w_cnt[4];
  __shared__ unsigned long long b_base;
  const unsigned long long m = __ballot(flag);
  const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
  if (lane == 0) w_cnt[wv] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = w_cnt[0] + w_cnt[1] + w_cnt[2] + w_cnt[3];
    b_base = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
  }
  __syncthreads();
  unsigned before = 0;
  for (int w = 0; w < wv; w++) before += w_cnt[w];
  if (flag) list[b_base + before + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = value;
  __syncthreads();
}

template <class C>
__global__ __launch_bounds__(256) void seed_first_kernel(SeedParamsT<C> p) {
  const uint64_t read = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = lane_id();
  const bool active = read < p.reads.n_reads;
  const int k = (int)p.min_seed_len;
  int d = 0, L = 0;
  C lo = 0, hi = 0;
  bool too_long = false;
  if (active) {
    const uint64_t r0 = p.reads.offsets[read];
    const uint64_t Lfull = p.reads.offsets[read + 1] - r0;
    too_long = Lfull > MAX_READ_LEN;
    L = too_long ? 0 : (int)Lfull;
    if (too_long) p.read_status[read] = THM_ERR_UNSUPPORTED;
    if (k <= L) ms_search(p.ix, p.reads.bases + r0, L, 0, k, d, lo, hi);
    if (L > 0) {
      const uint64_t item = ms_row(r0, read);
      p.ms_end[item] = (uint16_t)((d >= k) ? d : 0);
      p.ms_lo[item] = lo;
      p.ms_hi[item] = hi;
    }
  }
  const bool covered = active && d >= k && d == L;
  // reads that are not finished here: more positions to probe, or (a single position) left to the selection kernels
  const bool more = active && !covered && L - k + 1 >= 1;
  block_append(more && L <= (int)SHORT_READ_MAX, read, p.work_short, &p.work_counts[0]);
  block_append(more && L > (int)SHORT_READ_MAX, read, p.work_long, &p.work_counts[4]);
  if (active && !covered && !more) {  // shorter than a seed (or beyond MAX_READ_LEN): no SMEMs
    p.read_smem_off[read] = 0;
    p.read_smem_cnt[read] = 0;
    p.read_hits[read] = 0;
  }
  // a covered read is done: its one SMEM goes to the pool here (the selection kernels never see it)
  const unsigned long long m = __ballot(covered);
  if (m) {
    const int leader = (int)__builtin_ctzll(m);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(p.cursor, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader) << 32) | (unsigned)__shfl((int)(base & 0xffffffffu), leader);
    const bool fits = base + (unsigned long long)__popcll(m) <= p.smem_cap;
    unsigned long long hits = 0;
    if (covered) {
      const unsigned long long slot = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
      if (fits) {
        SmemT<C> sm;
        sm.lo = lo;
        sm.hi = hi;
        sm.qpos = 0;
        sm.len = (uint16_t)L;
        p.smems[slot] = sm;
      }
      hits = (unsigned long long)(hi - lo);
      p.read_smem_off[read] = slot;
      p.read_smem_cnt[read] = 1u;
      p.read_hits[read] = hits;
    }
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
    if (lane == leader) {
      if (!fits) atomicExch(p.fault, 1);
      atomicAdd(&p.counters[THM_CNT_SMEMS], (unsigned long long)__popcll(m));
      atomicAdd(&p.counters[THM_CNT_HITS], hits);
    }
  }
}

// `list[0 .. *count)`: the reads of one length class; G = grid slots per read for the longest read of the class
template <class C>
__global__ __launch_bounds__(256) void seed_grid_kernel(SeedParamsT<C> p, const unsigned long long* list,
                                                        const unsigned long long* count, uint32_t G) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t wi = tid / G;
  if (wi >= *count) return;
  const uint64_t read = list[wi];
  const int g = (int)(tid - wi * G);
  const uint64_t r0 = p.reads.offsets[read];
  const int L = (int)(p.reads.offsets[read + 1] - r0);
  const int k = (int)p.min_seed_len;
  const int npos = L - k + 1;  // >= 1 for listed reads
  const int pos = (g + 1 == (int)G) ? npos - 1 : g * PROBE_STRIDE;
  // position 0 was probed by seed_first_kernel; the last position is probed once (by the extra slot)
  if (pos <= 0 || pos >= npos || (g + 1 != (int)G && pos == npos - 1)) return;
  int d = 0;
  C lo = 0, hi = 0;
  ms_search(p.ix, p.reads.bases + r0, L, pos, k, d, lo, hi);
  const uint64_t item = ms_row(r0, read) + (uint64_t)pos;
  p.ms_end[item] = (uint16_t)((d >= k) ? pos + d : 0);
  p.ms_lo[item] = lo;
  p.ms_hi[item] = hi;
}

template <class C>
__global__ __launch_bounds__(256) void seed_cells_kernel(SeedParamsT<C> p, const unsigned long long* list,
                                                         const unsigned long long* count, uint32_t NC) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t wi = tid / NC;
  bool todo = false;
  uint64_t read = 0;
  int c = 0;
  if (wi < *count) {
    read = list[wi];
    c = (int)(tid - wi * NC);
    const uint64_t r0 = p.reads.offsets[read];
    const int L = (int)(p.reads.offsets[read + 1] - r0);
    const int npos = L - (int)p.min_seed_len + 1;
    const int a = c * PROBE_STRIDE, b = min(a + PROBE_STRIDE, npos - 1);
    if (b - a > 1) {  // the cell has inner positions
      const uint64_t item0 = ms_row(r0, read);
      const int ea = p.ms_end[item0 + (uint64_t)a], eb = p.ms_end[item0 + (uint64_t)b];
      if (ea != 0 && ea == eb) {
        for (int q = a + 1; q < b; q++) p.ms_end[item0 + (uint64_t)q] = (uint16_t)ea;  // same end: nothing starts here
      } else {
        todo = true;
      }
    }
  }
  block_append(todo, (read << 16) | (unsigned long long)c, p.work_cells, &p.work_counts[1]);
}

// slot `tid` of the fill stage -> (read, position); false: nothing to probe there
template <class C>
__device__ __forceinline__ bool fill_item(const SeedParamsT<C>& p, uint64_t tid, uint64_t& read, int& pos, uint64_t& r0, int& L) {
  const int j = (int)(tid % PROBE_STRIDE);
  if (j == 0) return false;  // the cell's left end is a grid position
  const unsigned long long w = p.work_cells[tid / PROBE_STRIDE];
  read = w >> 16;
  const int c = (int)(w & 0xffffu);
  r0 = p.reads.offsets[read];
  L = (int)(p.reads.offsets[read + 1] - r0);
  const int npos = L - (int)p.min_seed_len + 1;
  pos = c * PROBE_STRIDE + j;
  return pos < min(c * PROBE_STRIDE + PROBE_STRIDE, npos - 1);
}
template <class C>
__device__ __forceinline__ void fill_probe(const SeedParamsT<C>& p, uint64_t read, int pos, uint64_t r0, int L) {
  const int k = (int)p.min_seed_len;
  int d = 0;
  C lo = 0, hi = 0;
  ms_search(p.ix, p.reads.bases + r0, L, pos, k, d, lo, hi);
  const uint64_t item = ms_row(r0, read) + (uint64_t)pos;
  p.ms_end[item] = (uint16_t)((d >= k) ? pos + d : 0);
  p.ms_lo[item] = lo;
  p.ms_hi[item] = hi;
}

template <class C>
__global__ __launch_bounds__(256) void seed_fill_kernel(SeedParamsT<C> p) {
  const uint64_t total = p.work_counts[1] * PROBE_STRIDE;
  const uint64_t step = (uint64_t)gridDim.x * 256;  // (the worst-case grid covers every slot: one pass)
  for (uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x; tid < total; tid += step) {
    uint64_t read, r0;
    int pos, L;
    if (fill_item(p, tid, read, pos, r0, L)) fill_probe(p, read, pos, r0, L);
  }
}

// ---- the probes in bucket order (fill_mode 2) ----
template <class C>
__global__ __launch_bounds__(256) void seed_fill_keys_kernel(SeedParamsT<C> p) {
  const uint64_t total = p.work_counts[1] * PROBE_STRIDE;
  const uint64_t step = (uint64_t)gridDim.x * 256;
  for (uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x; tid < total; tid += step) {
    uint64_t read, r0;
    int pos, L;
    uint16_t key = 0xFFFF;
    if (fill_item(p, tid, read, pos, r0, L)) {
      const uint64_t w0 = load8_global(p.reads.bases + r0 + pos);
      uint32_t code = 0;
      bool acgt = true;
      for (int t = 0; t < FILL_KEY_BASES; t++) {
        const int c = base_code((uint8_t)((w0 >> (8 * t)) & 0xff));
        acgt = acgt && (c >= 0);
        code = (code << 2) | (uint32_t)(c & 3);
      }
      key = (uint16_t)(acgt ? code : FILL_BUCKETS);
      atomicAdd(&p.fill_hist[key], 1u);
    }
    p.fill_keys[tid] = key;
  }
}
// cursors = exclusive scan of the counts (one workgroup of 1024 threads), total behind them
__global__ __launch_bounds__(1024) void seed_fill_scan_kernel(unsigned int* hist) {
  __shared__ unsigned part[1024];
  constexpr unsigned NB = FILL_BUCKETS + 1, PER = (NB + 1023) / 1024;
  const unsigned t = threadIdx.x, b0 = t * PER;
  unsigned sum = 0;
  for (unsigned k = 0; k < PER; k++)
    if (b0 + k < NB) sum += hist[b0 + k];
  part[t] = sum;
  __syncthreads();
  for (unsigned o = 1; o < 1024; o <<= 1) {  // inclusive scan of the per-thread sums
    const unsigned v = t >= o ? part[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned run = part[t] - sum;
  unsigned int* cursor = hist + NB;
  for (unsigned k = 0; k < PER; k++)
    if (b0 + k < NB) {
      cursor[b0 + k] = run;
      run += hist[b0 + k];
    }
  if (t == 1023) hist[2 * NB] = part[1023];
}
template <class C>
__global__ __launch_bounds__(256) void seed_fill_scatter_kernel(SeedParamsT<C> p) {
  const uint64_t total = p.work_counts[1] * PROBE_STRIDE;
  const uint64_t step = (uint64_t)gridDim.x * 256;
  unsigned int* cursor = p.fill_hist + (FILL_BUCKETS + 1);
  for (uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x; tid < total; tid += step) {
    const uint16_t key = p.fill_keys[tid];
    if (key != 0xFFFF) p.fill_perm[atomicAdd(&cursor[key], 1u)] = (uint32_t)tid;
  }
}
template <class C>
__global__ __launch_bounds__(256) void seed_fill_bucketed_kernel(SeedParamsT<C> p) {
  const uint64_t total = p.fill_hist[2 * (FILL_BUCKETS + 1)];
  const uint64_t step = (uint64_t)gridDim.x * 256;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += step) {
    const uint64_t tid = p.fill_perm[s];
    uint64_t read, r0;
    int pos, L;
    if (fill_item(p, tid, read, pos, r0, L)) fill_probe(p, read, pos, r0, L);
  }
}

// per read: SMEM selection and ordering, one read per wavefront.  The per-read lists (one entry per
// position at most) live in LDS, or -- GS, reads of thousands of bases -- in a wave-private scratch in
// global memory.  `lcap` = list capacity (longest read of the class, rounded).
template <class C, bool GS>
__global__ __launch_bounds__(256) void seed_select_kernel(SeedParamsT<C> p, const unsigned long long* sel_list,
                                                          const unsigned long long* sel_count, unsigned int* queue, uint32_t lcap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = bcast_first((int)(threadIdx.x >> 6));  // wave-uniform: LDS bases stay on the scalar unit
  const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
  const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + (unsigned)wave;
  const size_t per_wave = (size_t)lcap * (2 * sizeof(C) + 8);
  uint8_t* base = GS ? p.sel_scratch + (size_t)wave_global * p.sel_scratch_per_wave : smem + (size_t)wave * per_wave;
  C* s_lo = (C*)base;                          // lcap entries
  C* s_hi = s_lo + lcap;                       // lcap
  uint16_t* a_end = (uint16_t*)(s_hi + lcap);  // lcap * 2 bytes
  uint16_t* s_pos = a_end + lcap;              // lcap * 2
  uint16_t* s_len = s_pos + lcap;              // lcap * 2
  uint16_t* s_em = s_len + lcap;               // lcap * 2
  auto sync = [] {
    if (GS)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  const int k = (int)p.min_seed_len;
  unsigned long long c_smems = 0, c_hits = 0;
  constexpr unsigned QCHUNK = 8;
  unsigned q_next = 0, q_end = 0;
  // Work distribution: most reads are dealt out in fixed interleaved chunks (wave w takes chunks
  // w, w + W, ...: no atomics), the last eighth through the atomic queue so that the waves finish
  // together.  One hot word serves only ~88 M returning atomics per second.
  const unsigned n_work = (unsigned)uload(sel_count);
  const unsigned n_static_chunks = (unsigned)(((unsigned long long)n_work * 7 / 8) / QCHUNK / n_waves) * n_waves;
  unsigned s_chunk = wave_global;
  unsigned long long pool_off = 0;
  unsigned pool_left = 0;
  for (;;) {
    if (q_next == q_end) {
      if (s_chunk < n_static_chunks) {
        q_next = s_chunk * QCHUNK;
        q_end = q_next + QCHUNK;
        s_chunk += n_waves;
      } else {
        unsigned g = 0;
        if (lane == 0) g = atomicAdd(queue, QCHUNK);
        g = n_static_chunks * QCHUNK + (unsigned)bcast_first((int)g);
        if (g >= n_work) break;
        q_next = g;
        q_end = min(g + QCHUNK, n_work);
      }
    }
    const unsigned idx = (unsigned)uload(&sel_list[q_next++]);
    const uint64_t r0 = uload(&p.reads.offsets[idx]);
    const int L = (int)(uload(&p.reads.offsets[idx + 1]) - r0);
    const uint64_t item0 = ms_row(r0, idx);
    const int npos = max(L - k + 1, 0);  // positions that were probed
    // Everything that depends only on the read index is requested at once (one memory round
    // trip): for up to 128 positions the whole row of ends and intervals; longer rows take the
    // intervals of the SMEM starts in a second trip.
    const bool row_in_regs = npos <= 128;
    uint16_t e_r[2] = {0, 0};
    C lo_r[2] = {0, 0}, hi_r[2] = {0, 0};
    if (row_in_regs) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int pos = j * 64 + lane;
        if (pos < npos) {
          e_r[j] = p.ms_end[item0 + pos];
          lo_r[j] = p.ms_lo[item0 + pos];
          hi_r[j] = p.ms_hi[item0 + pos];
          a_end[pos] = e_r[j];
        }
      }
    } else {
#pragma unroll 1
      for (int t = lane; t < npos; t += 64) a_end[t] = p.ms_end[item0 + t];
    }
    sync();

    // SMEM starts: end[i] > end[i-1]; compacted in start order
    int n_sm = 0;
#pragma unroll 1
    for (int b0 = 0; b0 < npos; b0 += 64) {
      const int pos = b0 + lane;
      bool is = false;
      int e = 0;
      if (pos < npos) {
        e = a_end[pos];
        const int prev = pos > 0 ? (int)a_end[pos - 1] : 0;
        is = e > 0 && e > prev;
      }
      const unsigned long long mask = __ballot(is);
      if (is) {
        const int at = n_sm + __popcll(mask & ((1ull << lane) - 1ull));
        s_pos[at] = (uint16_t)pos;
        s_len[at] = (uint16_t)(e - pos);
        s_lo[at] = row_in_regs ? (b0 == 0 ? lo_r[0] : lo_r[1]) : p.ms_lo[item0 + pos];
        s_hi[at] = row_in_regs ? (b0 == 0 ? hi_r[0] : hi_r[1]) : p.ms_hi[item0 + pos];
      }
      n_sm += __popcll(mask);
    }
    sync();

    // emission order of FMDIndex::all_smems: walk i0; the SMEMs covering i0 come
    // out by descending start; i0 jumps to the furthest end (or to the next start)
    if (n_sm > 1) {
      int t = 0, em = 0, i0 = 0;
      while (t < n_sm) {
        const int st = s_pos[t];
        if (st > i0) i0 = st;
        int u = t;
        while (u < n_sm && (int)s_pos[u] <= i0) u++;
        if (lane == 0)
          for (int v = u - 1; v >= t; v--) s_em[v] = (uint16_t)(em + (u - 1 - v));
        em += u - t;
        i0 = (int)s_pos[u - 1] + (int)s_len[u - 1];
        t = u;
      }
    } else if (lane == 0) {
      s_em[0] = 0;
    }
    sync();
    // order: length descending, then emission index descending
    // the SMEM pool is handed out to waves in slices (one atomic per slice: a single
    // hot word serves only ~88 M returning atomics per second)
    unsigned long long base_out = 0;
    if (n_sm > 0) {
      if ((unsigned)n_sm > pool_left) {
        const unsigned grab = max(256u, (unsigned)n_sm);
        unsigned long long g = 0;
        if (lane == 0) g = atomicAdd(p.cursor, (unsigned long long)grab);
        g = ((unsigned long long)(unsigned)bcast_first((int)(g >> 32)) << 32) | (unsigned)bcast_first((int)(g & 0xffffffffu));
        pool_off = g;
        pool_left = (g + grab <= p.smem_cap) ? grab : 0u;
        if (pool_left == 0) pool_off = p.smem_cap;  // forces !fits below
      }
      base_out = pool_off;
      if (pool_left >= (unsigned)n_sm) {
        pool_off += (unsigned)n_sm;
        pool_left -= (unsigned)n_sm;
      }
    }
    const bool fits = base_out + (unsigned long long)n_sm <= p.smem_cap;
    unsigned long long hits = 0;
    for (int t0 = 0; t0 < n_sm; t0 += 64) {
      const int t = t0 + lane;
      if (t < n_sm) {
        const int len = s_len[t], em = s_em[t];
        int rank = 0;
        for (int u = 0; u < n_sm; u++) {
          const int lu = s_len[u], eu = s_em[u];
          rank += (lu > len || (lu == len && eu > em)) ? 1 : 0;
        }
        if (fits) {
          SmemT<C> sm;
          sm.lo = s_lo[t];
          sm.hi = s_hi[t];
          sm.qpos = s_pos[t];
          sm.len = (uint16_t)len;
          p.smems[base_out + rank] = sm;
        }
        hits += (unsigned long long)(s_hi[t] - s_lo[t]);
      }
    }
    // wave sum of hits
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
    if (lane == 0) {
      if (!fits) atomicExch(p.fault, 1);
      p.read_smem_off[idx] = base_out;
      p.read_smem_cnt[idx] = (uint32_t)n_sm;
      p.read_hits[idx] = hits;
    }
    c_smems += (unsigned long long)n_sm;
    c_hits += hits;
    sync();
  }
  if (lane == 0 && (c_smems | c_hits)) {
    atomicAdd(&p.counters[THM_CNT_SMEMS], c_smems);
    atomicAdd(&p.counters[THM_CNT_HITS], c_hits);
  }
}

// SMEM selection and ordering with one read per THREAD, for the usual case of a handful of SMEMs in
// a read of at most SHORT_READ_MAX bases: the work per read is a few hundred instructions behind a chain of
// dependent loads, so the number of reads in flight is what counts (a wavefront per read leaves the
// machine waiting).  A read with more than SEL_CAP SMEMs goes on a list for seed_select_kernel.
constexpr int SEL_CAP = 6;

template <class C>
__global__ __launch_bounds__(256) void seed_select_thread_kernel(SeedParamsT<C> p, unsigned long long* over_list,
                                                                 unsigned long long* over_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const unsigned T = blockDim.x, tid = threadIdx.x;
  C* l_lo = (C*)smem;                    // [SEL_CAP][T]: conflict-free for a wave
  C* l_hi = l_lo + (size_t)SEL_CAP * T;  // [SEL_CAP][T]
  uint32_t* l_pl = (uint32_t*)(l_hi + (size_t)SEL_CAP * T);  // [SEL_CAP][T]: pos | len << 8 | emission index << 16
  const int lane = lane_id();
  const uint64_t wi = (uint64_t)blockIdx.x * T + tid;
  const bool active = wi < p.work_counts[0];
  const int k = (int)p.min_seed_len;
  int n_sm = 0;
  bool overflow = false;
  uint64_t read = 0;
  if (active) {
    read = p.work_short[wi];
    const uint64_t r0 = p.reads.offsets[read];
    const int L = (int)(p.reads.offsets[read + 1] - r0);
    const int npos = max(L - k + 1, 0);
    const uint64_t item0 = ms_row(r0, read);
    int prev = 0;
    // the row of ends, eight positions (16 bytes) per load: rows start on a multiple of 8 slots
    const uint4* row = (const uint4*)(p.ms_end + item0);
    for (int c0 = 0; c0 < npos; c0 += 8) {
      const uint4 v = row[c0 >> 3];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int pos = c0 + j;
        const int e = (int)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        if (pos < npos && e > 0 && e > prev) {  // an SMEM starts here
          if (n_sm < SEL_CAP) {
            l_lo[(size_t)n_sm * T + tid] = p.ms_lo[item0 + (uint64_t)pos];
            l_hi[(size_t)n_sm * T + tid] = p.ms_hi[item0 + (uint64_t)pos];
            l_pl[(size_t)n_sm * T + tid] = (uint32_t)pos | ((uint32_t)(e - pos) << 8);
          } else {
            overflow = true;
          }
          n_sm++;
        }
        prev = e;
      }
    }
    if (overflow) n_sm = 0;
  }
  block_append(overflow, read, over_list, over_count);

  // emission order of FMDIndex::all_smems over the read's SMEMs (in start order), as in seed_select_kernel
  if (n_sm > 1) {
    int t = 0, em = 0, i0 = 0;
    while (t < n_sm) {
      const int st = (int)(l_pl[(size_t)t * T + tid] & 0xffu);
      if (st > i0) i0 = st;
      int u = t;
      while (u < n_sm && (int)(l_pl[(size_t)u * T + tid] & 0xffu) <= i0) u++;
      for (int v = u - 1; v >= t; v--) l_pl[(size_t)v * T + tid] |= (uint32_t)(em + (u - 1 - v)) << 16;
      em += u - t;
      const uint32_t z = l_pl[(size_t)(u - 1) * T + tid];
      i0 = (int)(z & 0xffu) + (int)((z >> 8) & 0xffu);
      t = u;
    }
  }
  // pool entries of the wave: exclusive prefix sum of n_sm over the lanes, one atomic for the total
  int incl = n_sm;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63);
  unsigned long long base = 0;
  if (total > 0) {
    if (lane == 0) base = atomicAdd(p.cursor, (unsigned long long)total);
    base = ((unsigned long long)(unsigned)bcast_first((int)(base >> 32)) << 32) | (unsigned)bcast_first((int)(base & 0xffffffffu));
  }
  const bool fits = base + (unsigned long long)total <= p.smem_cap;
  const unsigned long long mine = base + (unsigned long long)(incl - n_sm);
  unsigned long long hits = 0;
  for (int t = 0; t < n_sm; t++) {
    const uint32_t a = l_pl[(size_t)t * T + tid];
    const int len = (int)((a >> 8) & 0xffu), em = (int)(a >> 16);
    int rank = 0;
    for (int u = 0; u < n_sm; u++) {
      const uint32_t b = l_pl[(size_t)u * T + tid];
      const int lu = (int)((b >> 8) & 0xffu), eu = (int)(b >> 16);
      rank += (lu > len || (lu == len && eu > em)) ? 1 : 0;
    }
    const C slo = l_lo[(size_t)t * T + tid], shi = l_hi[(size_t)t * T + tid];
    if (fits) {
      SmemT<C> sm;
      sm.lo = slo;
      sm.hi = shi;
      sm.qpos = (uint16_t)(a & 0xffu);
      sm.len = (uint16_t)len;
      p.smems[mine + rank] = sm;
    }
    hits += (unsigned long long)(shi - slo);
  }
  if (active && !overflow) {
    p.read_smem_off[read] = mine;
    p.read_smem_cnt[read] = (uint32_t)n_sm;
    p.read_hits[read] = hits;
  }
  int fault = (total > 0 && !fits) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) {
    hits += __shfl_xor(hits, o);
    fault |= __shfl_xor(fault, o);
  }
  if (lane == 0) {
    if (fault) atomicExch(p.fault, 1);
    if (total > 0) {
      atomicAdd(&p.counters[THM_CNT_SMEMS], (unsigned long long)total);
      atomicAdd(&p.counters[THM_CNT_HITS], hits);
    }
  }
}

// After the seed stage: the extend stage's lists.  Reads of the fast class with many seed hits (longest
// jobs first), reads of the slow class (band or length beyond the register-resident kernels), and the
// status of reads no kernel takes.
__global__ __launch_bounds__(256) void plan_kernel(PlanParams p) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = r < p.n_reads;
  uint64_t L = 0, hits = 0;
  if (active) {
    L = p.offsets[r + 1] - p.offsets[r];
    hits = p.read_hits[r];
  }
  const bool fast = active && L <= p.fast_max_len;
  const bool slow = active && !fast && L <= p.slow_max_len;
  if (active && !fast && !slow) {
    p.read_status[r] = THM_ERR_UNSUPPORTED;
    p.read_n_alns[r] = 0;
    p.read_op_bytes[r] = 0;
  }
  // (with the problem-parallel path in front, the wave-per-read kernel has little else to do: every read that path does
  // not take and the team kernel can, is the team's)
  const uint64_t team_thr = p.tpr_max_hits ? (uint64_t)max(p.tpr_max_hits, TEAM_MIN_HITS)
                                           : min((uint64_t)TEAM_HITS, max((uint64_t)TEAM_MIN_HITS, *p.total_hits / max(p.team_div, 1u)));
  const bool team = fast && p.team_ok && hits >= team_thr && hits <= TEAM_MAX_HITS;
  block_append(fast && !team && hits >= (p.tpr_max_hits ? (uint64_t)p.tpr_max_hits : (uint64_t)HEAVY_HITS), r, p.heavy, &p.counts[2]);
  block_append(slow, r, p.slow, &p.counts[5]);
  block_append(team, r, p.team, &p.counts[7]);
}

// One record per read for the extend kernel (launch.h, ReadRecT): thread per read
template <class C>
__global__ __launch_bounds__(256) void pack_reads_kernel(PackParamsT<C> p) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= p.n_reads) return;
  ReadRecT<C> rec;
  const uint64_t b0 = p.offsets[r], L = p.offsets[r + 1] - b0;
  rec.base_off = b0;
  rec.len = L > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)L;
  rec.smem_off = p.read_smem_off[r];
  rec.smem_cnt = p.read_smem_cnt[r];
  const uint64_t c0 = p.read_cand_off[r], nh = p.read_cand_off[r + 1] - c0;
  rec.cand_off = c0;
  rec.n_hits = nh > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nh;
  rec.qpos0 = rec.len0 = 0;
  rec.lo0 = rec.hi0 = rec.sa0 = 0;
  if (rec.smem_cnt > 0 && *p.fault_seed == 0) {  // after a pool overflow the runs are incomplete (the batch is replayed)
    const SmemT<C> sm = p.smems[rec.smem_off];
    rec.qpos0 = sm.qpos;
    rec.len0 = sm.len;
    rec.lo0 = sm.lo;
    rec.hi0 = sm.hi;
    if (sm.hi > sm.lo) rec.sa0 = p.sa[sm.hi - 1];
  }
  p.recs[r] = rec;
}

// Mem list of Index::all_smems for thm_smems_batch: one wave per read
template <class C>
__global__ __launch_bounds__(256) void expand_kernel(ExpandParamsT<C> p) {
  const int lane = lane_id();
  const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.n_reads) return;
  const uint64_t s0 = p.read_smem_off[r];
  const uint32_t ns = p.read_smem_cnt[r];
  uint64_t out = p.read_mem_off[r];
  for (uint32_t s = 0; s < ns; s++) {
    const SmemT<C> sm = p.smems[s0 + s];
    const uint64_t cnt = (uint64_t)(sm.hi - sm.lo);
    for (uint64_t t = lane; t < cnt; t += 64) {
      thm_mem m;
      m.ref_idx = p.ix.sa[sm.hi - 1 - t];
      m.query_idx = sm.qpos;
      m.len = sm.len;
      p.mems[out + t] = m;
    }
    out += cnt;
  }
}

}  // namespace dev

static uint32_t sel_lcap(uint32_t max_read_len) { return (max_read_len + 31u) & ~15u; }
// lists of the wavefront-per-read selection: per position 2 intervals bounds + 4 u16 (sized for the wide coordinates)
size_t seed_select_scratch_bytes(uint32_t max_read_len) { return (size_t)sel_lcap(max_read_len) * (2 * 8 + 8); }
size_t seed_select_lds_bytes(uint32_t max_read_len) { return 4 * seed_select_scratch_bytes(max_read_len); }

hipError_t launch_sanitize(const uint8_t* in, uint8_t* out, uint64_t n, uint64_t n_padded, hipStream_t s) {
  if (n_padded == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::sanitize_kernel, dim3((unsigned)((n_padded / 16 + 255) / 256 + 1)), dim3(256), 0, s, in, out, n, n_padded);
  return hipGetLastError();
}

template <class C>
static hipError_t launch_seed_t(const SeedParamsT<C>& p, int n_blocks, hipStream_t s) {
  const uint64_t n = p.reads.n_reads;
  if (n == 0) return hipSuccess;
  const uint32_t k = p.min_seed_len;
  auto blocks = [](uint64_t threads) { return dim3((unsigned)((threads + 255) / 256)); };
  auto cells_of = [&](uint32_t max_len) -> uint64_t {  // grid cells per read of a class
    const uint32_t P = (max_len >= k) ? max_len - k + 1 : 1;
    return (P + dev::PROBE_STRIDE - 1) / dev::PROBE_STRIDE;
  };
  hipError_t e;
  hipLaunchKernelGGL(dev::seed_first_kernel<C>, blocks(n), dim3(256), 0, s, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  const uint64_t n_short = n - p.n_long;
  // grids sized for the worst case; threads past the work-list counts (device memory) leave at once
  struct Cls {
    uint64_t n;
    uint32_t max_len;
    const unsigned long long* list;
    const unsigned long long* count;
  } cls[2] = {{n_short, p.max_len_short, p.work_short, p.work_counts + 0}, {p.n_long, p.max_len_long, p.work_long, p.work_counts + 4}};
  uint64_t fill_cells = 0;
  for (const Cls& c : cls) {
    if (c.n == 0 || c.max_len < k + 1) continue;  // a single position was probed by seed_first_kernel
    const uint64_t NC = cells_of(c.max_len), G = NC + 1;
    hipLaunchKernelGGL(dev::seed_grid_kernel<C>, blocks(c.n * G), dim3(256), 0, s, p, c.list, c.count, (uint32_t)G);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(dev::seed_cells_kernel<C>, blocks(c.n * NC), dim3(256), 0, s, p, c.list, c.count, (uint32_t)NC);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    fill_cells += c.n * NC;
  }
  if (fill_cells) {
    const dim3 worst = blocks(fill_cells * dev::PROBE_STRIDE);
    const dim3 fixed(std::min<unsigned>(worst.x, 256u * 8u * 4u));  // (8 workgroups per CU, four times over)
    if (p.fill_mode == 2 && fill_cells * dev::PROBE_STRIDE < (1ull << 32)) {
      hipLaunchKernelGGL(dev::seed_fill_keys_kernel<C>, fixed, dim3(256), 0, s, p);
      hipLaunchKernelGGL(dev::seed_fill_scan_kernel, dim3(1), dim3(1024), 0, s, p.fill_hist);
      hipLaunchKernelGGL(dev::seed_fill_scatter_kernel<C>, fixed, dim3(256), 0, s, p);
      hipLaunchKernelGGL(dev::seed_fill_bucketed_kernel<C>, fixed, dim3(256), 0, s, p);
    } else {
      hipLaunchKernelGGL(dev::seed_fill_kernel<C>, p.fill_mode == 1 ? fixed : worst, dim3(256), 0, s, p);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  // SMEM selection: short reads one per thread, its overflow list and the long reads through the
  // wavefront-per-read kernel
  auto select_waves = [&](const unsigned long long* list, const unsigned long long* count, unsigned int* queue, uint32_t max_len,
                          int nb) -> hipError_t {
    const uint32_t lcap = sel_lcap(max_len);
    const size_t lds = 4 * (size_t)lcap * (2 * sizeof(C) + 8);
    if (lds <= SEED_SELECT_LDS_LIMIT) {
      hipLaunchKernelGGL((dev::seed_select_kernel<C, false>), dim3(nb), dim3(256), lds, s, p, list, count, queue, lcap);
    } else {
      hipLaunchKernelGGL((dev::seed_select_kernel<C, true>), dim3(nb), dim3(256), 0, s, p, list, count, queue, lcap);
    }
    return hipGetLastError();
  };
  if (n_short) {
    unsigned long long* over_list = p.work_cells;  // free again: seed_fill_kernel is done with it
    unsigned long long* over_count = p.work_counts + 3;
    hipLaunchKernelGGL(dev::seed_select_thread_kernel<C>, dim3((unsigned)((n_short + 255) / 256)), dim3(256),
                       (size_t)256 * (2 * sizeof(C) + 4) * dev::SEL_CAP, s, p, over_list, over_count);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // the overflow list is short: every wave costs an atomic just to find it empty
    if ((e = select_waves(over_list, over_count, p.queue, p.max_len_short, std::min(n_blocks, 256))) != hipSuccess) return e;
  }
  if (p.n_long) {
    const int nb = (int)std::min<uint64_t>((uint64_t)n_blocks, (p.n_long + 3) / 4);
    if ((e = select_waves(p.work_long, p.work_counts + 4, p.queue + 1, p.max_len_long, std::max(nb, 1))) != hipSuccess) return e;
  }
  return hipSuccess;
}
hipError_t launch_seed(const SeedParamsT<uint32_t>& p, int n_blocks, hipStream_t s) { return launch_seed_t(p, n_blocks, s); }
hipError_t launch_seed(const SeedParamsT<uint64_t>& p, int n_blocks, hipStream_t s) { return launch_seed_t(p, n_blocks, s); }

hipError_t launch_plan(const PlanParams& p, hipStream_t s) {
  if (p.n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::plan_kernel, dim3((unsigned)((p.n_reads + 255) / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}

template <class C>
static hipError_t launch_pack_reads_t(const PackParamsT<C>& p, hipStream_t s) {
  if (p.n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::pack_reads_kernel<C>, dim3((unsigned)((p.n_reads + 255) / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_pack_reads(const PackParamsT<uint32_t>& p, hipStream_t s) { return launch_pack_reads_t(p, s); }
hipError_t launch_pack_reads(const PackParamsT<uint64_t>& p, hipStream_t s) { return launch_pack_reads_t(p, s); }

template <class C>
static hipError_t launch_expand_t(const ExpandParamsT<C>& p, hipStream_t s) {
  const unsigned blocks = (unsigned)((p.n_reads + 3) / 4);
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::expand_kernel<C>, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_expand(const ExpandParamsT<uint32_t>& p, hipStream_t s) { return launch_expand_t(p, s); }
hipError_t launch_expand(const ExpandParamsT<uint64_t>& p, hipStream_t s) { return launch_expand_t(p, s); }

}  // namespace thm
