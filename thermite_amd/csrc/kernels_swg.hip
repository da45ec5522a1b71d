// kernels_swg.hip -- operator-level kernel: a batch of independent
// SwgExtend::extend problems (reference src/swg.rs:31-167), one problem per
// wavefront, pulled from a device work queue.  This is the surface the
// reference's known-answer tests pin (src/swg.rs:249-317).
#include <hip/hip_runtime.h>

#include "launch.h"
#include "swg_device.h"

namespace thm {
namespace dev {

// per-wave layout of the any-width variant's slice of global memory (CPL == 0)
struct SwgSlowLayout {
  uint64_t xs, ys, ops, dp, trace, total;
  uint32_t dp_stride, ops_cap;
};
__host__ __device__ inline SwgSlowLayout swg_slow_layout(uint32_t x_cap, uint32_t y_cap, uint32_t max_bw) {
  SwgSlowLayout s;
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) {
    const uint64_t at = o;
    o += (bytes + 63u) & ~63ull;
    return at;
  };
  s.ops_cap = x_cap + y_cap + 16;
  s.xs = take(x_cap);
  s.ys = take(y_cap);
  s.ops = take(s.ops_cap);
  const uint32_t tiles = (2u * max_bw + 1u + 63u) / 64u;
  s.dp_stride = tiles * 64u + 64u;
  s.dp = take((uint64_t)s.dp_stride * 16u);
  s.trace = take((uint64_t)(y_cap + 2u) * tiles * 16u);
  s.total = o;
  return s;
}

// CPL = band slots per lane (1..4), or 0: band of any width (swg_extend_tiled, buffers in global memory)
template <int CPL>
__global__ __launch_bounds__(256) void swg_batch_kernel(SwgBatchParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = bcast_first((int)(threadIdx.x >> 6));  // wave-uniform: LDS bases stay on the scalar unit
  uint8_t *xs, *ys, *opsb;
  unsigned long long* trace;
  int* dp = nullptr;
  int dp_stride = 0;
  uint32_t ops_cap;
  if constexpr (CPL == 0) {
    const SwgSlowLayout sl = swg_slow_layout(p.x_cap, p.y_cap, p.max_bw);
    uint8_t* base = p.scratch + (size_t)(blockIdx.x * (blockDim.x >> 6) + (unsigned)wave) * p.scratch_per_wave;
    xs = base + sl.xs;
    ys = base + sl.ys;
    opsb = base + sl.ops;
    trace = (unsigned long long*)(base + sl.trace);
    dp = (int*)(base + sl.dp);
    dp_stride = (int)sl.dp_stride;
    ops_cap = sl.ops_cap;
  } else {
    const uint32_t tr_bytes = (p.y_cap + 1) * CPL * 16;
    ops_cap = p.x_cap + p.y_cap + 16;
    const uint32_t per_wave = p.x_cap + p.y_cap + tr_bytes + ops_cap;
    uint8_t* base = smem + (size_t)wave * per_wave;
    xs = base;
    ys = xs + p.x_cap;
    trace = (unsigned long long*)(ys + p.y_cap);
    opsb = (uint8_t*)trace + tr_bytes;
  }
  auto sync = [] {
    if (CPL == 0)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  unsigned long long n_cells = 0, n_cols = 0, n_calls = 0;
  constexpr unsigned QCHUNK = 8;  // problems per queue atomic (a single hot word serves ~88 M atomics/s)
  unsigned q_next = 0, q_end = 0;
  for (;;) {
    if (q_next == q_end) {
      unsigned g = 0;
      if (lane == 0) g = atomicAdd(p.queue, QCHUNK);
      g = (unsigned)bcast_first((int)g);
      if (g >= p.n) break;
      q_next = g;
      q_end = (unsigned)min((uint64_t)g + QCHUNK, p.n);
    }
    const unsigned idx = q_next++;
    const uint64_t x0 = p.xo[idx], y0 = p.yo[idx];
    const int xlen = (int)(p.xo[idx + 1] - x0);
    const int ylen_full = (int)(p.yo[idx + 1] - y0);
    const int bw = (int)p.bw[idx];
    const int xd = p.xd[idx];
    // only the first xlen+bw+1 columns are reachable
    const int ylen = min(ylen_full, xlen + bw + 1);
    #pragma unroll 1
    for (int t = lane; t < xlen; t += 64) xs[t] = p.xb[x0 + t];
    #pragma unroll 1
    for (int t = lane; t < ylen; t += 64) ys[t] = p.yb[y0 + t];
    sync();

    SwgResult r;
    int nops;
#ifndef THM_NO_SHORTCUT
    if (swg_one_mismatch_shortcut(xs, 1, xlen, ys, 1, ylen, xd, opsb + ops_cap - 1, -1, (int)ops_cap, r, nops)) {
      // result known without DP (swg_device.h)
    } else
#endif
    if constexpr (CPL == 0) {
      r = swg_extend_tiled(xs, 1, xlen, ys, 1, ylen, bw, xd, trace, dp, dp_stride);
      sync();
      nops = swg_traceback_tiled(trace, r.xend, r.yend, bw, opsb + ops_cap - 1, -1, (int)ops_cap);
    } else {
      r = swg_extend_wave<(CPL > 0 ? CPL : 1)>(xs, 1, xlen, ys, 1, ylen, bw, xd, trace);
      sync();
      // path from the max cell back to the origin, laid out so that it reads forward
      nops = swg_traceback_wave<(CPL > 0 ? CPL : 1)>(trace, r.xend, r.yend, bw, opsb + ops_cap - 1, -1, (int)ops_cap);
    }
    sync();
    if (nops < 0) {
      if (lane == 0) atomicExch(p.fault, 1);
      nops = 0;
    }
    uint8_t* out = p.ops + p.ops_off[idx];
    const uint8_t* src = opsb + ops_cap - nops;
    #pragma unroll 1
    for (int t = lane; t < nops; t += 64) out[t] = src[t];
    uint32_t total = (uint32_t)nops;
    if (r.xend < xlen) {  // reference :178-180 Xclip(len - i), last after the reverse
      if (lane == 0) {
        const uint32_t clip = (uint32_t)(xlen - r.xend);
        out[nops] = THM_OP_XCLIP;
        out[nops + 1] = (uint8_t)(clip);
        out[nops + 2] = (uint8_t)(clip >> 8);
        out[nops + 3] = (uint8_t)(clip >> 16);
        out[nops + 4] = (uint8_t)(clip >> 24);
      }
      total += 5;
    }
    if (lane == 0) {
      thm_swg_aln a;
      a.ops_off = p.ops_off[idx];
      a.ops_len = total;
      a.score = r.score;
      a.xend = (uint32_t)r.xend;
      a.yend = (uint32_t)r.yend;
      p.out[idx] = a;
    }
    n_cells += r.cells;
    n_cols += r.cols;
    n_calls += 1;
  }
  if (lane == 0 && n_calls) {
    atomicAdd(&p.counters[THM_CNT_SWG_CALLS], n_calls);
    atomicAdd(&p.counters[THM_CNT_DP_CELLS], n_cells);
    atomicAdd(&p.counters[THM_CNT_DP_COLS], n_cols);
  }
}

// wave primitive self-test (checked against numpy on the GPU box)
__global__ void wave_prims_kernel(const int* in, int* out) {
  const int l = lane_id();
  const int v = in[l];
  out[l] = wave_incl_max_scan(v);
  out[64 + l] = wave_excl_max_scan(v);
  out[128 + l] = wave_max(v);
  out[192 + l] = wave_min(v);
  out[256 + l] = wave_shr1(v, -7);
  out[320 + l] = wave_shl1(v, -9);
}

}  // namespace dev

size_t swg_batch_scratch_bytes(const SwgBatchParams& p) { return (size_t)dev::swg_slow_layout(p.x_cap, p.y_cap, p.max_bw).total; }

size_t swg_batch_lds_bytes(const SwgBatchParams& p, int cpl) {
  if (cpl == 0) return 0;
  const size_t tr = (size_t)(p.y_cap + 1) * cpl * 16;
  return 4 * ((size_t)p.x_cap + p.y_cap + tr + p.x_cap + p.y_cap + 16);
}

hipError_t launch_swg_batch(const SwgBatchParams& p, int cpl, int n_blocks, hipStream_t s) {
  const size_t lds = swg_batch_lds_bytes(p, cpl);
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(256), lds, s, p);
    return hipGetLastError();
  };
  switch (cpl) {
    case 0: return go(dev::swg_batch_kernel<0>);
    case 1: return go(dev::swg_batch_kernel<1>);
    case 2: return go(dev::swg_batch_kernel<2>);
    case 3: return go(dev::swg_batch_kernel<3>);
    case 4: return go(dev::swg_batch_kernel<4>);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_wave_prims(const int* in, int* out, hipStream_t s) {
  hipLaunchKernelGGL(dev::wave_prims_kernel, dim3(1), dim3(64), 0, s, in, out);
  return hipGetLastError();
}

}  // namespace thm
