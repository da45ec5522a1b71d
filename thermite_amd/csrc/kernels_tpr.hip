// kernels_tpr.hip -- align_read with extension problems, not reads, as the unit of wavefront work.
//
// align_read (reference src/aligner.rs:123-190) is control flow and table lookups around SwgExtend::extend calls
// (src/swg.rs:31-167).  The wave-per-read kernel (kernels_extend.hip) gives a read a whole wavefront for all of it:
// the control flow runs wave-uniform in 64 lanes, a narrow band leaves most lanes idle, and a launch ends when its
// slowest read does.  Here the two kinds of work are taken apart:
//
//   extend_ctl_kernel   one read per THREAD: contig of a hit (Index::idx_to_ref, src/index.rs:287-290), the exon
//                       intervals over the seed (exon_to_tx.find, src/aligner.rs:232-236), lift_mem_to_tx
//                       (src/txome.rs:82-103), extend_seed_match (src/aligner.rs:410-426), the score rules of the hit loop
//                       (:146-174), lift_tx_to_gx (src/txome.rs:110-160), concat_to_chr_aln (:429-449), retain /
//                       filter_overlapping / sort (:177-187), serialisation.  An extend() call whose result is known in
//                       closed form (empty x or y, src/swg.rs:39-55; one mismatch next to the seed and an exact match
//                       behind it, swg_device.h::swg_one_mismatch_shortcut; a single mismatching base) is answered in
//                       place.  Any other call becomes a REQUEST record.
//   extend_dp_kernel    one request per wavefront: stage x and y in LDS, swg_extend_wave + swg_traceback_wave
//                       (swg_device.h), result and op list back into the record.
//
// The hits of a read must be taken in order (band, X-drop and best score are carried from hit to hit,
// src/aligner.rs:143-175), so a read needs the results of hit k before it can state the problems of hit k + 1: the
// two kernels alternate in ROUNDS.  The control kernel keeps no state between rounds but the results themselves: in
// every round it replays the read from its first hit (thread-level work: cheap), taking the DP results of earlier
// rounds from the read's memo, until it meets a hit whose results are missing; it requests the extension problems
// of that hit (genome window and every transcript target) and goes to sleep.  The carried state changes only when a
// hit beats the best score so far -- a handful of times per read, early -- so once a read has an accepted candidate the
// request covers ALL its remaining hits under the state in force (the scheme of the team kernel, kernels_extend.hip):
// the next replay takes those results for as long as the state really stays what it was and asks again, under the
// new state, from the first hit behind a change.  Exact: every result that is used was computed under exactly the
// band and X-drop the sequential loop has at that hit.  A read whose replay gets through its last hit is finished:
// final filters, serialisation, done.  Reads that exceed one of the small fixed capacities of this path (rounds,
// candidates, grid entries, introns) go on the lists of the wave-per-read kernels, which remain the general path;
// reads with TPR_MAX_HITS hits and more are theirs from the start (a thread replaying thousands of hits would be the
// tail of the launch).
//
// Exactness.  Every extend() call gets the inputs the reference gives it (x, y, band, X-drop; y cut to the
// |x| + bw + 1 reachable columns) and is computed by the same device code as before.  Two
// calls are not computed: (a) a transcript target whose x and whose y bytes equal the genome problem's takes the
// genome result (same inputs, same result); (b) the genome extensions of a hit are dead when some transcript target
// is known, in closed form, to reach the upper bound of the genome score: exonic iff best.score >= gx.score
// (src/aligner.rs:263) and the genome result is used for nothing else.  The bound: an extension whose first pair
// mismatches scores at most max(|x| - 2, 0) unless x == y[1..|x|+1) (one leading deletion, which costs no gap-open:
// |x| - 1) -- the penalty argument of swg_one_mismatch_shortcut.  The SwgExtend::extend calls are counted in the
// reference's terms either way (THM_CNT_SWG_CALLS); THM_CNT_DP_CELLS / _COLS count the work actually done.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.h"
#include "swg_device.h"

namespace thm {
namespace dev {

namespace {

constexpr int TPR_KEEP = 8;      // accepted candidates a thread keeps book of
constexpr int TPR_MAX_ENT = 24;  // exon-grid candidates of one query a thread walks
constexpr int TPR_MAX_MK = 4;    // introns of one alignment
constexpr int TPR_MAX_PEND = 24;  // extension problems one read may ask for in one round
#ifndef THM_TPR_CTL_MINW
#define THM_TPR_CTL_MINW 4
#endif
constexpr int TPR_CTL_MINW = THM_TPR_CTL_MINW;  // waves per SIMD the control kernel's register budget is set for

enum : int { FAULT_OPS_POOL = 1 };  // kernels_extend.hip

template <class C>
struct TCoord {
  typedef int S;
};
template <>
struct TCoord<uint64_t> {
  typedef long long S;
};

__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);  // one global_load_dwordx2 (unaligned access is enabled for global memory)
  return v;
}
struct U16 {
  uint64_t lo, hi;
};
__device__ __forceinline__ U16 ld16(const uint8_t* p) {
  U16 v;
  __builtin_memcpy(&v, p, 16);  // one global_load_dwordx4
  return v;
}
// leading positions t < n with a[t] == b[t]; touches nothing outside [a, a + n) and [b, b + n)
__device__ __forceinline__ int match_fwd(const uint8_t* a, const uint8_t* b, int n) {
  int t = 0;
  while (t + 16 <= n) {
    const U16 x = ld16(a + t), y = ld16(b + t);
    const uint64_t d0 = x.lo ^ y.lo, d1 = x.hi ^ y.hi;
    if (d0) return t + (int)(__builtin_ctzll(d0) >> 3);
    if (d1) return t + 8 + (int)(__builtin_ctzll(d1) >> 3);
    t += 16;
  }
  while (t + 8 <= n) {
    const uint64_t d = ld8(a + t) ^ ld8(b + t);
    if (d) return t + (int)(__builtin_ctzll(d) >> 3);
    t += 8;
  }
  while (t < n && a[t] == b[t]) t++;
  return t;
}
// positions t < n with a[-1 - t] == b[-1 - t], walking backwards; touches nothing outside [a - n, a) and [b - n, b)
__device__ __forceinline__ int match_bwd(const uint8_t* a, const uint8_t* b, int n) {
  int t = 0;
  while (t + 16 <= n) {
    const U16 x = ld16(a - 16 - t), y = ld16(b - 16 - t);
    const uint64_t d0 = x.lo ^ y.lo, d1 = x.hi ^ y.hi;
    if (d1) return t + (int)(__builtin_clzll(d1) >> 3);
    if (d0) return t + 8 + (int)(__builtin_clzll(d0) >> 3);
    t += 16;
  }
  while (t + 8 <= n) {
    const uint64_t d = ld8(a - 8 - t) ^ ld8(b - 8 - t);
    if (d) return t + (int)(__builtin_clzll(d) >> 3);
    t += 8;
  }
  while (t < n && a[-1 - t] == b[-1 - t]) t++;
  return t;
}

// a[0 .. n) == b[0 .. n), and (through `uniform`) whether every a[t] equals the byte c; touches nothing outside the ranges
__device__ __forceinline__ bool equal_and_uniform(const uint8_t* a, const uint8_t* b, int n, uint8_t c, bool& uniform) {
  const uint64_t splat = 0x0101010101010101ull * (uint64_t)c;
  uint64_t diff = 0, nonu = 0;
  int t = 0;
  while (t + 16 <= n) {
    const U16 x = ld16(a + t), y = ld16(b + t);
    diff |= (x.lo ^ y.lo) | (x.hi ^ y.hi);
    nonu |= (x.lo ^ splat) | (x.hi ^ splat);
    if (diff) {
      uniform = false;
      return false;
    }
    t += 16;
  }
  while (t + 8 <= n) {
    const uint64_t x = ld8(a + t), y = ld8(b + t);
    diff |= x ^ y;
    nonu |= x ^ splat;
    t += 8;
  }
  while (t < n) {
    diff |= (uint64_t)(a[t] ^ b[t]);
    nonu |= (uint64_t)(a[t] ^ c);
    t++;
  }
  uniform = nonu == 0;
  return diff == 0;
}

// One SwgExtend::extend call as the control kernel sees it.  rec < 0: the result is known in closed form (ops in
// walk order from the seed outwards: Subst if sp == 0, then Match).  rec >= 0: a DP record.
struct Side {
  int score, xend, yend, n, sp;
  int rec;     // DP record index, -1: closed form
  int ub;      // upper bound of the score (== score when known)
  bool known;  // closed form
};

// Classification of one extension.  x0 / y0: the first symbols as the extension walks them; dir = +1 (right) or -1.
__device__ __forceinline__ Side side_classify(const uint8_t* x0, const uint8_t* y0, int dir, int xlen, long long ylen, int xd) {
  Side s;
  s.score = s.xend = s.yend = s.n = 0;
  s.sp = -1;
  s.rec = -1;
  s.ub = 0;
  s.known = true;
  if (xlen == 0 || ylen <= 0) return s;  // src/swg.rs:39-55 (the clip is implied by xend)
  const bool first_eq = x0[0] == y0[0];
  if (xlen == 1 && !first_eq) return s;  // one base that mismatches: no cell of row 1 exceeds 0 (2 - j at best)
  s.known = false;
  s.ub = xlen;
  if (first_eq) return s;
  // The rest of x and of y as forward ranges: a left extension walks both backwards, so its symbols 1 .. n-1 are the
  // n - 1 bytes BEFORE the first ones.
  const int n1 = xlen - 1;
  const uint8_t* xa = dir > 0 ? x0 + 1 : x0 - n1;
  const uint8_t* ya = dir > 0 ? y0 + 1 : y0 - n1;
  // swg_one_mismatch_shortcut: |x| >= 3, |y| >= |x|, x_drop >= 1, x[0] != y[0], x[1..] == y[1..|x|), x not one repeated base
  if (xlen >= 3 && ylen >= (long long)xlen && xd >= 1) {
    bool uniform;  // x[1..] all equal to x[0]: one repeated base
    if (equal_and_uniform(xa, ya, n1, x0[0], uniform) && !uniform) {
      s.known = true;
      s.score = s.ub = xlen - 2;
      s.xend = s.yend = s.n = xlen;
      s.sp = 0;
      return s;
    }
  }
  // upper bound for a first pair that mismatches: |x| - 1 if x == y[1 .. |x| + 1) (a leading deletion), else max(|x| - 2, 0)
  bool lead_del = false;
  if (ylen >= (long long)xlen + 1) {
    const uint8_t* xf = dir > 0 ? x0 : x0 - n1;
    const uint8_t* yf = dir > 0 ? y0 + 1 : y0 - 1 - n1;
    lead_del = match_fwd(xf, yf, xlen) == xlen;
  }
  s.ub = lead_del ? xlen - 1 : max(xlen - 2, 0);
  return s;
}

template <class S>
struct TPath {
  int score, xstart, xend, nops, nl, len;  // nl ops of the left extension, then `len` Match, then the right extension's
  Side l, r;
  S ystart, yend;
};

template <class C, class IX>
__device__ __forceinline__ void idx_to_ref_thread(const IX& ix, C idx, RefRecT<C>& r, uint32_t& id) {
  uint32_t lo = ix.ref_bin[idx >> GRID_SHIFT];
  r = ix.ref_recs[lo];
  while (r.end <= idx && lo + 1 < ix.n_refs) {
    lo++;
    r = ix.ref_recs[lo];
  }
  id = lo;
}

// an extension problem whose result is missing, until the workgroup's allocation has given it a record
struct Pend {
  const uint8_t* x0;
  const uint8_t* y0;
  uint16_t xlen, ylen;
  int8_t dir;
  uint8_t cls;
  uint16_t pad_;
};

// what the kernel keeps of an accepted candidate until the read is finished
struct TCand {
  uint64_t ch0, ch1, ylen;  // chromosome coordinates
  int score, xstart, xend, nops, nl, len;
  int l_sp, r_sp, l_rec, r_rec;
  uint32_t ref_id, name_rank, type_idx;
  uint32_t mk_k[TPR_MAX_MK], ycl[TPR_MAX_MK];  // introns: path index they precede, length
  int tx_ystart, tx_yend;
  uint32_t tx_ylen;
  uint8_t strand, aln_type, rev, n_y;
};

// op k of a candidate's path: rev(left.ops) ++ Match x len ++ right.ops (src/aligner.rs:388-394).  A DP record holds
// its ops in traceback order (from the end cell back to the seed).
struct PathView {
  int nl, len, nops, l_sp, r_sp;
  const uint8_t* l_ops;  // DP ops of the left extension (null: closed form)
  const uint8_t* r_ops;
  __device__ __forceinline__ uint8_t op(int k) const {
    if (k < nl) return l_ops ? l_ops[k] : (uint8_t)((l_sp >= 0 && k == nl - 1 - l_sp) ? OPK_SUBST : OPK_MATCH);
    if (k < nl + len) return (uint8_t)OPK_MATCH;
    const int t = k - nl - len, nr = nops - nl - len;
    return r_ops ? r_ops[nr - 1 - t] : (uint8_t)((r_sp >= 0 && t == r_sp) ? OPK_SUBST : OPK_MATCH);
  }
};

// One op stream: [Xclip(lead)] path with introns [Xclip(trail)], mirrored as a list when `rev`
// (kernels_extend.hip::emit_alignment).  Paths are Match almost everywhere: zero fill, then patches.
// `o` is 4-byte aligned and the allocation is rounded up to 4 bytes.
__device__ __forceinline__ void emit_stream(uint8_t* o, int total, const PathView& pv, int lead, int trail, bool rev, int n_y,
                                            const uint32_t* mk_k, const uint32_t* ycl) {
  static_assert(OPK_MATCH == 0, "zero fill");
  uint32_t* o4 = (uint32_t*)o;
  for (int t = 0; t * 4 < total; t++) o4[t] = 0u;
  const int lead5 = lead > 0 ? 5 : 0;
  auto put1 = [&](int fpos, uint8_t v) { o[rev ? total - (fpos + 1) : fpos] = v; };
  auto put5 = [&](int fpos, uint8_t kind, uint32_t v) {
    uint8_t* d = o + (rev ? total - (fpos + 5) : fpos);
    d[0] = kind;
    d[1] = (uint8_t)v;
    d[2] = (uint8_t)(v >> 8);
    d[3] = (uint8_t)(v >> 16);
    d[4] = (uint8_t)(v >> 24);
  };
  auto before = [&](int k) {
    int b = 0;
    for (int m = 0; m < TPR_MAX_MK; m++) b += (m < n_y && (int)mk_k[m] <= k) ? 1 : 0;
    return b;
  };
  // the ops that are not Match
  if (!pv.l_ops) {
    if (pv.l_sp >= 0) {
      const int k = pv.nl - 1 - pv.l_sp;
      put1(lead5 + k + 5 * before(k), (uint8_t)OPK_SUBST);
    }
  } else {
    for (int k = 0; k < pv.nl; k++) {
      const uint8_t v = pv.l_ops[k];
      if (v) put1(lead5 + k + 5 * before(k), v);
    }
  }
  const int r0 = pv.nl + pv.len, nr = pv.nops - r0;
  if (!pv.r_ops) {
    if (pv.r_sp >= 0) {
      const int k = r0 + pv.r_sp;
      put1(lead5 + k + 5 * before(k), (uint8_t)OPK_SUBST);
    }
  } else {
    for (int t = 0; t < nr; t++) {
      const uint8_t v = pv.r_ops[nr - 1 - t];
      if (v) put1(lead5 + r0 + t + 5 * before(r0 + t), v);
    }
  }
  for (int m = 0; m < TPR_MAX_MK; m++)
    if (m < n_y) put5(lead5 + (int)mk_k[m] + 5 * m, THM_OP_YCLIP, ycl[m]);
  if (lead > 0) put5(0, THM_OP_XCLIP, (uint32_t)lead);
  if (trail > 0) put5(total - 5, THM_OP_XCLIP, (uint32_t)trail);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Control kernel: one read per thread and pass of the grid-stride loop.
//   round 0: the reads are 0 .. n_reads; later rounds: the list the round before left (tp.act_in).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// quantities a workgroup allocates in one go: cand op bytes, records, DP op bytes, sleepers, bails to the heavy list,
// bails to the team list, queue slots of the four band classes
constexpr int NALLOC = 10;
enum { A_OPS = 0, A_REC = 1, A_DPO = 2, A_ACT = 3, A_BAIL = 4, A_BAILT = 5, A_Q0 = 6 };

// everything one walk over a hit's extension problems produces
template <class S, class C>
struct HitOut {
  TPath<S> gx, best;
  bool have_best, gx_dead;
  uint32_t best_tx, best_ent, e0;  // e0: first exon-grid entry of the query (best_ent counts from it)
  RefRecT<C> ref;
  uint32_t ref_id;
  unsigned calls, win;
  unsigned long long cells, cols;  // DP work of the results the hit used
  int why;
};
}  // namespace

template <class C>
__global__ __launch_bounds__(256, TPR_CTL_MINW) void extend_ctl_kernel(ExtendParamsT<C> p, TprParamsT<C> tp) {
  typedef typename TCoord<C>::S S;
  __shared__ unsigned long long s_cnt[THM_N_COUNTERS];
  __shared__ unsigned s_part[4][NALLOC];
  __shared__ unsigned long long s_base[NALLOC];
  __shared__ unsigned s_stats[8];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  if (threadIdx.x < THM_N_COUNTERS) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 8) s_stats[threadIdx.x] = 0;
  __syncthreads();
  const auto& ix = p.ix;
  const uint64_t n_items = (uint64_t)*tp.n_act_in;  // round 0: the reads of this path by descending hit count (tpr_order_kernel)
  const uint64_t gsz = (uint64_t)gridDim.x * 256;
  const uint64_t n_iter = (n_items + gsz - 1) / gsz;
  const bool dead_run = *p.fault_seed != 0;  // SMEM pool overflow: the host grows the pool and replays the batch

  for (uint64_t it = 0; it < n_iter; it++) {
    const uint64_t item = it * gsz + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    // (a slot of the list whose read went to the wave-per-read kernel after all holds 0xFFFFFFFF)
    const uint64_t idx_raw = item < n_items ? (uint64_t)tp.act_in[item] : 0;
    const bool active = item < n_items && !dead_run && idx_raw != 0xFFFFFFFFull;
    const uint64_t idx = active ? idx_raw : 0;
    bool done = false, sleep = false, bail = false;
    int why = 0;  // statistics only
    TCand cd[TPR_KEEP];
    int res[TPR_KEEP];  // final order: res[t] = index into cd
    uint32_t nres = 0;
    unsigned calls = 0, winbytes = 0;
    unsigned long long dp_cells = 0, dp_cols = 0;
    uint64_t cand0 = 0;
    int L = 0;
    ReadMemo memo;
    memo.n_rounds = 0;
    // replay position in the memo: the round whose records the walk reads, the next record of it, and whether the
    // state in force is still the one the round was requested under
    int k_round = -1;
    unsigned ord_r = 0;
    bool round_ok = false;
    // the frontier (the first hit whose results are missing) and the extension problems requested from it on (the
    // hit's own and, speculatively, those of the hits behind it): collected by the walks, written after the
    // workgroup's allocation
    uint32_t f_hno = 0;
    int f_bw = 0, f_xd = 0;
    Pend pend[TPR_MAX_PEND];
    int n_pend = 0;
    bool pend_over = false;

    ReadRecT<C> rec;
    rec.len = 0xFFFFFFFFu;
    rec.base_off = 0;
    rec.n_hits = 0;
    if (active) rec = p.read_recs[idx];
    const bool mine = active && rec.len <= p.max_read_len && rec.n_hits < tp.max_hits;
    const uint8_t* rd = p.reads.bases + rec.base_off;
    L = mine ? (int)rec.len : 0;

    // the read's hits in align_read's order: SMEMs as listed, occurrences by descending suffix-array rank
    auto load_sm = [&](uint32_t si) -> SmemT<C> {
      SmemT<C> sm;
      if (si == 0) {
        sm.lo = rec.lo0;
        sm.hi = rec.hi0;
        sm.qpos = rec.qpos0;
        sm.len = rec.len0;
      } else {
        sm = p.smems[rec.smem_off + si];
      }
      return sm;
    };

    // One walk over the extension problems of a hit, in the order the reference meets them.
    //   compute  take the DP results from the memo and produce the hit's outcome; returns 1 when they are missing
    //   else     (a hit behind the frontier, asked for speculatively) only collect the problems
    // Problems whose results are missing are collected in pend[].  Order of the records of a hit: the unknown sides of
    // the transcript targets in yield order (right, left), then those of the genome window -- which are requested only
    // if the genome problem is not dead, known by then.
    auto walk_hit = [&](const bool compute, const SmemT<C>& sm, const C rr, const bool first_occ, const int bw, const int xd, HitOut<S, C>& o) -> int {
      const int mode = compute ? 0 : 1;
      const int q = sm.qpos, len = sm.len;
      const C hrc = first_occ ? rec.sa0 : ix.sa[rr - 1];
      const S hr = (S)hrc;
      idx_to_ref_thread<C>(ix, hrc, o.ref, o.ref_id);
      const C qs = hrc, qe = (C)(hrc + (C)len);
      // genome window (:212-215)
      const S seq_start = max((hr > (S)(L + bw)) ? hr - (S)(L + bw) : (S)0, (S)o.ref.start);
      const S seq_end = min(hr + (S)(len + L + bw), (S)o.ref.end - 1);
      bool need = false;
      o.calls = o.win = 0;
      o.cells = o.cols = 0;
      o.why = 0;
      // an extension that needs a DP: its result from the memo, or one more problem to ask for
      auto unknown_side = [&](Side& s, const uint8_t* x0, const uint8_t* y0, int dir, int xlen, long long ylen) {
        const int slots = min(2 * bw + 1, xlen + 1);
        if (compute) {
          const unsigned od = ord_r++;
          if (round_ok && od < (unsigned)memo.cnt[k_round]) {
            const uint32_t ri = memo.base[k_round] + od;
            const DpRec* d = tp.recs + ri;
            s.score = d->score;
            s.xend = d->xend;
            s.yend = d->yend;
            s.n = d->nops;
            s.rec = (int)ri;
            o.cells += d->cells;
            o.cols += d->cols;
            // cannot be: every record of an earlier round was computed, for exactly this problem; and a hit's results
            // are there as a whole or not at all
            if (need || d->done != 1 || d->xlen != (uint16_t)xlen || d->ylen != (uint16_t)ylen || d->bw != (uint16_t)bw || d->dir != (int8_t)dir) {
              o.why = 7;
              atomicOr(p.fault, 2 | 128);  // (diagnosis)
            }
            return;
          }
          need = true;
        }
        if (n_pend < TPR_MAX_PEND) {
          Pend& e = pend[n_pend++];
          e.x0 = x0;
          e.y0 = y0;
          e.xlen = (uint16_t)xlen;
          e.ylen = (uint16_t)ylen;
          e.dir = (int8_t)dir;
          e.cls = (uint8_t)((slots + 63) / 64);
          e.pad_ = 0;
        } else {
          pend_over = true;
        }
      };
      // the two sides of extend_left_right (src/aligner.rs:352-407) for a target spanning [lo_abs, hi_abs)
      struct LrGeom {
        const uint8_t *xr0, *yr0, *xl0, *yl0;
        int xr, xl;
        long long yr, yl;
      };
      auto lr_geom = [&](const uint8_t* ybase, S lo_abs, S hi_abs, S r, int q2, int len2) -> LrGeom {
        LrGeom g;
        g.xr = L - (q2 + len2);
        g.yr = min((long long)(hi_abs - (r + len2)), (long long)(g.xr + bw + 1));
        g.xr0 = rd + q2 + len2;
        g.yr0 = ybase + (r + len2);
        g.xl = q2;
        const S rel = r - lo_abs;
        const S y0 = lo_abs + (rel > (S)(L + bw) ? rel - (S)(L + bw) : (S)0);
        g.yl = min((long long)(r - y0), (long long)(g.xl + bw + 1));
        g.xl0 = rd + q2 - 1;
        g.yl0 = ybase + (r - 1);
        return g;
      };
      auto finish_path = [&](TPath<S>& t, S r, int q2, int len2) {
        t.nl = t.l.n;
        t.len = len2;
        t.nops = t.l.n + len2 + t.r.n;
        t.score = t.l.score + len2 * MATCH_SCORE + t.r.score;
        t.ystart = r - (S)t.l.yend;
        t.yend = r + (S)len2 + (S)t.r.yend;
        t.xstart = q2 - t.l.xend;
        t.xend = q2 + len2 + t.r.xend;
      };
      // ---- genome window: classified now, requested (if at all) behind the transcript targets ----
      o.win += (unsigned)(seq_end - seq_start);
      o.calls += 2;
      const LrGeom gg = lr_geom(ix.text, seq_start, seq_end, hr, q, len);
      o.gx.r = side_classify(gg.xr0, gg.yr0, 1, gg.xr, gg.yr, xd);
      o.gx.l = side_classify(gg.xl0, gg.yl0, -1, gg.xl, gg.yl, xd);
      const bool gx_known = o.gx.l.known && o.gx.r.known;
      const int gx_ub = o.gx.l.ub + len * MATCH_SCORE + o.gx.r.ub;
      // exon_to_tx.find(seed) in yield order (:231-258): by ascending pre-order rank.  A bin's entries are sorted by
      // rank, a seed spans one bin or two: a merge of two sorted lists, skipping what does not overlap and the second
      // copy of an interval listed in both bins.
      const uint32_t b0 = (uint32_t)(qs >> GRID_SHIFT), b1 = (uint32_t)((qe > qs ? qe - 1 : qs) >> GRID_SHIFT);
      const uint32_t e0 = ix.exon_grid_off[b0], e1 = ix.exon_grid_off[b1 + 1];
      const uint32_t emid = (b1 == b0) ? e1 : ix.exon_grid_off[b0 + 1];
      const ExonEntryT<C>* ent = ix.exon_grid;
      o.e0 = e0;
      if (e1 - e0 > (uint32_t)TPR_MAX_ENT || b1 > b0 + 1) {
        o.why = 2;
        return 2;
      }
      uint32_t gi = e0, gj = emid;  // cursors of the two lists
      auto grid_skip = [&](uint32_t& t, uint32_t end) {
        while (t < end) {
          const C es = ent[t].start, ee = ent[t].end;
          const uint32_t home = max(b0, (uint32_t)(es >> GRID_SHIFT));
          if (qs < ee && es < qe && (ent[t].rank & 0xffu) == (home & 0xffu)) break;
          t++;
        }
      };
      o.have_best = false;
      o.best_tx = o.best_ent = 0;
      int known_best = -1;  // best score among the targets known in closed form
      // the first target that is the genome problem again while that is not known yet: its score comes with the genome's
      bool alias = false;
      int alias_pos = 0, best_pos = 0, pos = 0;
      uint32_t alias_tx = 0, alias_ent = 0;
      int alias_tr = 0;
      bool stop = false;
      while (!stop) {
        grid_skip(gi, emid);
        grid_skip(gj, e1);
        uint32_t ei;
        if (gi < emid && (gj >= e1 || ent[gi].rank <= ent[gj].rank))
          ei = gi++;
        else if (gj < e1)
          ei = gj++;
        else
          break;
        const ExonEntryT<C> ge = ent[ei];
        if (!(ge.prev_end <= qs)) {  // lift_mem_to_tx's general case (a seed across a short intron)
          o.why = 3;
          return 2;
        }
        // lift_mem_to_tx (src/txome.rs:82-103)
        const S xs = (S)ge.start, xe = (S)ge.end;
        const int exon_sum = (int)ge.txoff;
        int tr_ = (int)((hr > xs) ? hr - xs : (S)0) + exon_sum;
        const int start_offset = (int)((xs > hr) ? xs - hr : (S)0);
        const int t_end = (int)(min(hr + (S)len, xe) - xs) + exon_sum;
        int t_q = q + start_offset;
        int t_len = t_end - tr_;
        const int tlen = (int)ge.seq_len;
        const int ws = (tr_ > L + bw) ? tr_ - (L + bw) : 0;
        const int we = min(tlen, tr_ + t_len + L + bw + 1);
        o.win += (unsigned)(we - ws);
        o.calls += 2;
        const uint8_t* seq = ix.tx_seq + ge.seq_off;
        // extend_seed_match (src/aligner.rs:410-426)
        {
          int ext = match_fwd(seq + tr_ + t_len, rd + t_q + t_len, min(tlen - (tr_ + t_len), L - (t_q + t_len)));
          t_len += ext;
          ext = match_bwd(seq + tr_, rd + t_q, min(tr_, t_q));
          tr_ -= ext;
          t_q -= ext;
          t_len += ext;
        }
        const LrGeom tg = lr_geom(seq, (S)0, (S)tlen, (S)tr_, t_q, t_len);
        // Same seed on the read and the same y bytes as the genome problem (the hit lies inside one exon that covers
        // both windows): the two extend() calls have the genome calls' inputs, hence its results.
        // (With a genome problem known in closed form the target's sides are simply classified: same bytes, same answer.)
        bool same = false;
        if (!gx_known && t_q == q && t_len == len && tg.yr == gg.yr && tg.yl == gg.yl) {
          same = (gg.xr == 0 || gg.yr <= 0 || match_fwd(tg.yr0, gg.yr0, (int)gg.yr) == (int)gg.yr) &&
                 (gg.xl == 0 || gg.yl <= 0 || match_fwd(tg.yl0 + 1 - (int)gg.yl, gg.yl0 + 1 - (int)gg.yl, (int)gg.yl) == (int)gg.yl);
        }
        TPath<S> pth;
        bool p_known, p_scored = true;  // scored: the target's score is in pth.score now
        if (same) {
          pth.l = o.gx.l;
          pth.r = o.gx.r;
          p_known = false;
          p_scored = false;
          if (!alias) {
            alias = true;
            alias_pos = pos;
            alias_tx = ge.value;
            alias_ent = ei - e0;
            alias_tr = tr_;
          }
        } else {
          pth.r = side_classify(tg.xr0, tg.yr0, 1, tg.xr, tg.yr, xd);
          if (!pth.r.known) unknown_side(pth.r, tg.xr0, tg.yr0, 1, tg.xr, tg.yr);
          pth.l = side_classify(tg.xl0, tg.yl0, -1, tg.xl, tg.yl, xd);
          if (!pth.l.known) unknown_side(pth.l, tg.xl0, tg.yl0, -1, tg.xl, tg.yl);
          p_known = pth.l.known && pth.r.known;
        }
        if (p_known) {
          finish_path(pth, (S)tr_, t_q, t_len);
          known_best = max(known_best, pth.score);
        }
        if (mode == 0 && !need && p_scored) {
          if (!p_known) finish_path(pth, (S)tr_, t_q, t_len);
          if (!o.have_best || pth.score > o.best.score) {  // strictly better (:249)
            o.have_best = true;
            o.best_tx = ge.value;
            o.best_ent = ei - e0;
            o.best = pth;
            best_pos = pos;
          }
        }
        // cannot beat an exact match (:253-257); a target that needs a DP scores below L
        if (p_known && pth.score >= L * MATCH_SCORE) stop = true;
        pos++;
      }
      // ---- the genome problem: dead when a target known in closed form reaches its upper bound and no target takes
      // its result (it is not computed then and takes no records); else its unknown sides are the hit's last requests ----
      o.gx_dead = !gx_known && !alias && known_best >= gx_ub;
      if (!o.gx_dead) {
        if (!o.gx.r.known) unknown_side(o.gx.r, gg.xr0, gg.yr0, 1, gg.xr, gg.yr);
        if (!o.gx.l.known) unknown_side(o.gx.l, gg.xl0, gg.yl0, -1, gg.xl, gg.yl);
      }
      if (o.why) return 2;
      if (mode != 0) return 0;
      if (need) return 1;
      finish_path(o.gx, hr, q, len);  // (a dead genome problem: never looked at)
      if (alias) {
        // the deferred target: the genome's result in transcript coordinates; earlier in yield order wins ties (:249)
        TPath<S> a = o.gx;
        a.ystart = (S)alias_tr - (S)o.gx.l.yend;
        a.yend = (S)alias_tr + (S)len + (S)o.gx.r.yend;
        if (!o.have_best || a.score > o.best.score || (a.score == o.best.score && alias_pos < best_pos)) {
          o.have_best = true;
          o.best_tx = alias_tx;
          o.best_ent = alias_ent;
          o.best = a;
        }
      }
      return 0;
    };
    if (mine) {
      cand0 = rec.cand_off;
      if (cand0 + rec.n_hits > p.cand_cap) {
        bail = true;  // the wave-per-read kernel raises the pool fault
        why = 7;
      } else {
        if (tp.round > 0) memo = tp.memos[idx];  // (round 0: no rounds yet)
        // thresholds, src/aligner.rs:130-138 (binary32 product, truncation toward zero)
        const float prod = p.opts.min_aln_score_percent * (float)L;
        const int ms_pct = (prod != prod) ? 0 : (prod >= 2147483648.0f ? 2147483647 : (prod <= -2147483648.0f ? (-2147483647 - 1) : (int)prod));
        const int min_aln_score = max(ms_pct, p.opts.min_aln_score);
        int max_aln_score = min_aln_score;
        int band_width = (min_aln_score < 0) ? 0 : max(L - min_aln_score, 0);
        int x_drop = band_width;
        const int range = (int)p.opts.multimap_score_range;
        const bool intron_mode = p.opts.intron_mode != 0;
        if (band_width > (int)p.max_bw) {  // the wave-per-read kernel reports the inconsistency
          bail = true;
          why = 1;
        }
        uint32_t n_acc = 0, hno = 0;
        const uint32_t n_sm = rec.smem_cnt;
        for (uint32_t si = 0; !bail && !sleep && si < n_sm; si++) {
          const SmemT<C> sm = load_sm(si);
          for (C rr = sm.hi; !bail && !sleep && rr > sm.lo; rr--, hno++) {
            // ================= align_seed_hit (src/aligner.rs:198-314) =================
            const int bw = band_width, xd = x_drop;
            // a round of the memo that was requested from this hit on: its records are for the state in force now
            while (k_round + 1 < (int)memo.n_rounds && memo.first_hit[k_round + 1] == (uint16_t)hno) {
              k_round++;
              ord_r = 0;
              round_ok = true;
            }
            HitOut<S, C> h;
            const int st = walk_hit(true, sm, rr, si == 0 && rr == sm.hi, bw, xd, h);
            if (st == 2) {
              bail = true;
              why = h.why;
              break;
            }
            if (st == 1) {
              // results are missing: this hit is the round's frontier; its problems are in pend[]
              if (memo.n_rounds >= TPR_MAX_ROUNDS || tp.last_round != 0 || pend_over || n_pend == 0) {
                bail = true;
                why = 6;
                break;
              }
              // While no candidate has been accepted the state is the wide initial one and the first acceptance
              // will narrow it: ask for few hits (1, 2, 4, ...).  After that it rarely moves: all the remaining hits,
              // as far as pend[] holds their problems.  A hit behind the frontier that this path cannot take ends the
              // batch; if the replay gets there it asks again, or leaves the read to the wave-per-read kernel.
              uint32_t left = n_acc > 0 ? 0xFFFFFFFFu : (1u << min((int)memo.n_rounds, 5));
              uint32_t si2 = si;
              SmemT<C> sm2 = sm;
              C rr2 = rr;
              while (--left) {
                rr2--;
                while (rr2 <= sm2.lo) {  // next SMEM with occurrences
                  si2++;
                  if (si2 >= n_sm) break;
                  sm2 = load_sm(si2);
                  rr2 = sm2.hi;
                }
                if (si2 >= n_sm) break;
                const int mark = n_pend;
                HitOut<S, C> h2;
                const int st2 = walk_hit(false, sm2, rr2, false, bw, xd, h2);
                if (st2 == 2 || pend_over) {
                  n_pend = mark;
                  pend_over = false;
                  break;
                }
              }
              sleep = true;
              f_bw = bw;
              f_xd = xd;
              f_hno = hno;
              break;
            }
            calls += h.calls;
            winbytes += h.win;
            dp_cells += h.cells;
            dp_cols += h.cols;
            const ExonEntryT<C>* ent = ix.exon_grid + h.e0;
            // ---- exonic vs unspliced (:263-313) ----
            const bool exonic = h.have_best && (h.gx_dead || h.best.score >= h.gx.score);
            const TPath<S>& sel = exonic ? h.best : h.gx;
            const int sc = sel.score;
            // ================= back in align_read's loop (:146-174) =================
            bool accept = intron_mode || exonic;
            if (sc < p.opts.min_aln_score || sc < min_aln_score || sc < max_aln_score - range) accept = false;
            if (!accept) continue;
            if (n_acc >= (uint32_t)TPR_KEEP) {
              bail = true;
              why = 6;
              break;
            }
            TCand k;
            k.n_y = 0;
            k.tx_ystart = k.tx_yend = 0;
            k.tx_ylen = 0;
            for (int m = 0; m < TPR_MAX_MK; m++) k.mk_k[m] = k.ycl[m] = 0;
            k.nops = sel.nops;
            k.nl = sel.nl;
            k.len = sel.len;
            k.l_sp = sel.l.sp;
            k.r_sp = sel.r.sp;
            k.l_rec = sel.l.rec;
            k.r_rec = sel.r.rec;
            S cy0, cy1;
            int aln_type;
            uint32_t type_idx = THM_NO_IDX;
            if (exonic) {
              const TPath<S>& best = h.best;
              aln_type = THM_ALN_EXONIC;
              type_idx = h.best_tx;
              // lift_tx_to_gx (src/txome.rs:110-160)
              const ExonEntryT<C> ge = ent[h.best_ent];
              const int ys_ = (int)best.ystart, ye_ = (int)best.yend;
              k.tx_ystart = ys_;
              k.tx_yend = ye_;
              k.tx_ylen = ge.seq_len;
              const int e_lo = (int)ge.txoff, e_hi = e_lo + (int)(ge.end - ge.start);
              const bool inside = ys_ >= e_lo && ys_ < e_hi &&
                                  (ye_ < e_hi || (ye_ == e_hi && (ge.exon_idx + 1 >= ge.n_exons || !(best.xend < L))));
              if (inside) {
                cy0 = (S)ge.start + (S)(ys_ - e_lo);
                cy1 = (S)ge.start + (S)(ye_ - e_lo);
              } else {
                PathView pv;
                pv.nl = best.nl;
                pv.len = best.len;
                pv.nops = best.nops;
                pv.l_sp = best.l.sp;
                pv.r_sp = best.r.sp;
                pv.l_ops = best.l.rec >= 0 ? tp.dp_ops + tp.recs[best.l.rec].ops_off : nullptr;
                pv.r_ops = best.r.rec >= 0 ? tp.dp_ops + tp.recs[best.r.rec].ops_off : nullptr;
                const thm_tx tx = ix.txs[h.best_tx];
                const thm_exon* ex = ix.exons + tx.exon_begin;
                const uint64_t* toff = ix.exon_txoff + tx.exon_begin;
                const int ne = (int)tx.n_exons;
                int lo = 0, hi = ne;
                while (lo < hi) {  // exon where the alignment starts (:123-126)
                  const int mid = (lo + hi) >> 1;
                  if ((int)(toff[mid] + (ex[mid].end - ex[mid].start)) <= ys_)
                    lo = mid + 1;
                  else
                    hi = mid;
                }
                int e = lo;
                // transcript positions advance on Match / Subst / Del; their number must be yend - ystart (:154)
                int n_adv = 0;
                const bool all_adv = !pv.l_ops && !pv.r_ops;
                if (all_adv) {
                  n_adv = best.nops;
                } else {
                  for (int kk = 0; kk < best.nops; kk++) n_adv += pv.op(kk) != OPK_INS;
                }
                if (e >= ne || n_adv != ye_ - ys_) {  // panics in the reference: the wave-per-read kernel reports them
                  bail = true;
                  why = 7;
                  break;
                }
                thm_exon cur = ex[e];
                int xsum = (int)toff[e];
                cy0 = (S)cur.start + (S)(ys_ - xsum);
                const bool trailing_clip = best.xend < L;
                int n_y = 0;
                int scan_k = 0, scan_adv = 0;  // ops [0, scan_k) hold scan_adv advancing ones
                for (;;) {
                  const int bnd = xsum + (int)(cur.end - cur.start);
                  if (e + 1 >= ne || bnd > ye_) break;
                  const int need_adv = bnd - ys_;  // 1-based rank, among the advancing ops, of the op that reaches bnd
                  int kstar;
                  if (all_adv) {
                    kstar = need_adv;
                  } else {
                    while (scan_adv < need_adv && scan_k < best.nops) {
                      scan_adv += pv.op(scan_k) != OPK_INS;
                      scan_k++;
                    }
                    kstar = scan_adv == need_adv ? scan_k : -1;
                  }
                  if (kstar < 0) break;
                  if (kstar >= best.nops && !trailing_clip) break;
                  if (n_y >= TPR_MAX_MK) {
                    bail = true;
                    why = 6;
                    break;
                  }
                  const thm_exon nxt = ex[e + 1];
                  k.mk_k[n_y] = (uint32_t)kstar;
                  k.ycl[n_y] = (uint32_t)(nxt.start - cur.end);
                  n_y++;
                  xsum = bnd;
                  cur = nxt;
                  e++;
                }
                if (bail) break;
                k.n_y = (uint8_t)n_y;
                cy1 = (S)cur.start + (S)(ye_ - xsum);
              }
            } else {
              cy0 = h.gx.ystart;
              cy1 = h.gx.yend;
              aln_type = THM_ALN_INTERGENIC;
              // first interval gene_intervals.find yields (:283-288, :306); only reached in intron mode
              const C gs = (C)cy0, ge_ = (C)cy1;
              const uint32_t g0 = (uint32_t)(gs >> GRID_SHIFT), g1 = (uint32_t)((ge_ > gs ? ge_ - 1 : gs) >> GRID_SHIFT);
              const uint32_t f0 = ix.gene_grid_off[g0], f1 = ix.gene_grid_off[g1 + 1];
              if (f1 - f0 > 8u * (uint32_t)TPR_MAX_ENT) {
                bail = true;
                why = 2;
                break;
              }
              int best_rank = 0x0fffffff;
              for (uint32_t t = f0; t < f1; t++) {
                const GridEntryT<C> g = ix.gene_grid[t];
                const bool overlap = gs < g.end && g.start < ge_;
                const uint32_t home = max(g0, (uint32_t)(g.start >> GRID_SHIFT));
                const bool primary = (g.rank & 0xffu) == (home & 0xffu);
                const int r2 = (overlap && primary) ? (int)(g.rank >> 8) : -1;
                if (r2 >= 0 && r2 < best_rank) {
                  best_rank = r2;
                  type_idx = g.value;
                  aln_type = THM_ALN_INTRONIC;
                }
              }
            }
            // concat_to_chr_aln (:429-449)
            RefRecT<C> cref = h.ref;
            if (!((C)cy0 >= h.ref.start && (C)cy0 < h.ref.end)) {
              uint32_t dummy;
              idx_to_ref_thread<C>(ix, (C)cy0, cref, dummy);
            }
            if (cref.strand != 0) {
              k.ch0 = (uint64_t)((C)cy0 - cref.start);
              k.ch1 = (uint64_t)((C)cy1 - cref.start);
              k.rev = 0;
            } else {
              k.ch0 = (uint64_t)(C)(cref.len - ((C)cy1 - cref.start));
              k.ch1 = (uint64_t)(C)(cref.len - ((C)cy0 - cref.start));
              k.rev = 1;
            }
            k.ylen = cref.len;
            k.score = sc;
            k.xstart = sel.xstart;
            k.xend = sel.xend;
            k.ref_id = h.ref_id;
            k.name_rank = h.ref.name_rank;
            k.type_idx = type_idx;
            k.strand = h.ref.strand != 0 ? 1 : 0;
            k.aln_type = (uint8_t)aln_type;
            cd[n_acc] = k;
            n_acc++;
            // narrow the band (:162-172)
            const int lim = max(L + range - sc, 0);
            band_width = min(band_width, lim);
            x_drop = min(x_drop, lim);
            max_aln_score = max(max_aln_score, sc);
            // requests behind this hit were made for the band and X-drop in force when it was reached
            if (band_width != bw || x_drop != xd) round_ok = false;
          }
        }
        if (!bail && !sleep) {
          done = true;
          // ============ retain / filter_overlapping / sort / primary (:177-187) ============
          // retain(score >= max - range) keeps the order; stable sort by (ref_name, strand, ystart) (:322-327)
          int srt[TPR_KEEP];
          int m = 0;
          for (int t = 0; t < TPR_KEEP; t++) srt[t] = res[t] = 0;
          for (int t = 0; t < (int)n_acc; t++) {
            if (cd[t].score < max_aln_score - range) continue;
            const TCand& a = cd[t];
            int pos = m;  // insertion behind every kept candidate that is not greater: stable
            while (pos > 0) {
              const TCand& b = cd[srt[pos - 1]];
              bool a_less;
              if (a.name_rank != b.name_rank)
                a_less = a.name_rank < b.name_rank;
              else if (a.strand != b.strand)
                a_less = a.strand < b.strand;
              else
                a_less = a.ch0 < b.ch0;
              if (!a_less) break;
              srt[pos] = srt[pos - 1];
              pos--;
            }
            srt[pos] = t;
            m++;
          }
          // sweep (:329-346)
          {
            uint64_t max_end = 0, l_yend = 0;
            uint32_t l_rank = 0, l_strand = 0;
            int l_score = 0;
            for (int s = 0; s < m; s++) {
              const TCand& a = cd[srt[s]];
              if (nres == 0 || a.ch0 >= max_end || a.name_rank != l_rank || a.strand != l_strand) {
                max_end = a.ch1;
                res[nres] = srt[s];
                nres++;
                l_rank = a.name_rank;
                l_strand = a.strand;
                l_score = a.score;
                l_yend = a.ch1;
              } else {
                if (a.score > l_score) {
                  res[nres - 1] = srt[s];
                  l_score = a.score;
                  l_yend = a.ch1;
                }
                max_end = max(max_end, l_yend);
              }
            }
          }
          // stable sort by -score (:183)
          for (int s = 1; s < (int)nres; s++) {
            const int v = res[s];
            int pos = s;
            while (pos > 0 && cd[res[pos - 1]].score < cd[v].score) {
              res[pos] = res[pos - 1];
              pos--;
            }
            res[pos] = v;
          }
        }
      }
    }
    if (bail) sleep = false;

    // ---- workgroup allocations: op bytes of finished reads; records, DP op room, queue and list slots of sleeping /
    // bailing reads.  One exclusive scan per quantity over the workgroup, one atomic per quantity and workgroup. ----
    // a read left to the wave-per-read kernels goes to the workgroup-per-read (team) kernel if it has the hits for it
    const bool bail_team = bail && tp.team != nullptr && rec.n_hits >= TEAM_MIN_HITS && rec.n_hits <= TEAM_MAX_HITS;
    unsigned want[NALLOC];
    for (int a2 = 0; a2 < NALLOC; a2++) want[a2] = 0;
    if (done) {
      for (uint32_t t = 0; t < nres; t++) {
        const TCand& a = cd[res[t]];
        const int lead5 = a.xstart > 0 ? 5 : 0, trail5 = (L - a.xend) > 0 ? 5 : 0;
        const unsigned nb = (unsigned)(lead5 + a.nops + 5 * (int)a.n_y + trail5);
        const unsigned tnb = a.aln_type == THM_ALN_EXONIC ? (unsigned)(lead5 + a.nops + trail5) : 0u;
        want[A_OPS] += ((nb + 3u) & ~3u) + ((tnb + 3u) & ~3u);
      }
    }
    if (sleep) {
      want[A_REC] = (unsigned)n_pend;
      want[A_ACT] = 1;
      for (int t = 0; t < n_pend; t++) {
        want[A_DPO] += ((unsigned)pend[t].xlen + (unsigned)pend[t].ylen + 3u) & ~3u;
        want[A_Q0 + pend[t].cls - 1]++;
      }
    }
    if (bail) want[bail_team ? A_BAILT : A_BAIL] = 1;
    unsigned incl[NALLOC];
    for (int a2 = 0; a2 < NALLOC; a2++) {
      unsigned v = want[a2];
      for (int o2 = 1; o2 < 64; o2 <<= 1) {
        const unsigned t0 = (unsigned)__shfl_up((int)v, o2);
        if (lane >= o2) v += t0;
      }
      incl[a2] = v;
      if (lane == 63) s_part[wave][a2] = v;
    }
    __syncthreads();
    if (threadIdx.x < NALLOC) {
      const int a2 = (int)threadIdx.x;
      const unsigned tot = s_part[0][a2] + s_part[1][a2] + s_part[2][a2] + s_part[3][a2];
      unsigned long long* cur = a2 == A_OPS ? p.ops_cursor
                                : a2 == A_REC ? tp.rec_cursor
                                : a2 == A_DPO ? tp.dp_ops_cursor
                                : a2 == A_ACT ? tp.n_act_out
                                : a2 == A_BAIL ? tp.bail_count
                                : a2 == A_BAILT ? tp.team_count
                                                : &tp.q_cur[a2 - A_Q0];
      s_base[a2] = (tot && cur) ? atomicAdd(cur, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long mine_at[NALLOC], blk_end[NALLOC];
    for (int a2 = 0; a2 < NALLOC; a2++) {
      unsigned long long b2 = s_base[a2];
      unsigned tot = 0;
      for (int w = 0; w < 4; w++) {
        if (w < wave) b2 += s_part[w][a2];
        tot += s_part[w][a2];
      }
      mine_at[a2] = b2 + incl[a2] - want[a2];
      blk_end[a2] = s_base[a2] + tot;
    }
    const bool pool_ok = blk_end[A_OPS] <= p.cand_ops_cap;
    if (!pool_ok && threadIdx.x == 0 && blk_end[A_OPS] > s_base[A_OPS]) atomicOr(p.fault, FAULT_OPS_POOL);  // the host grows the pool and replays the batch
    if (!pool_ok) done = false;
    // request pools exhausted: the read goes to the wave-per-read kernel instead (no replay needed)
    const bool req_ok = blk_end[A_REC] <= tp.rec_cap && blk_end[A_REC] < 0x7FFFFF00ull && blk_end[A_DPO] <= tp.dp_ops_cap &&
                        blk_end[A_Q0] <= tp.q_stride && blk_end[A_Q0 + 1] <= tp.q_stride && blk_end[A_Q0 + 2] <= tp.q_stride &&
                        blk_end[A_Q0 + 3] <= tp.q_stride;
    if (sleep && !req_ok) {
      sleep = false;
      const unsigned long long slot = atomicAdd(tp.bail_count, 1ull);  // (a late bail has no slot of this pass's allocation)
      tp.bail[slot] = idx;
      why = 6;
      tp.act_out[mine_at[A_ACT]] = 0xFFFFFFFFu;  // its slot in the next round's list stays: marked empty
      {  // ... and so do its slots in the DP queues (as far as they lie inside the lists)
        unsigned long long qs4[4] = {mine_at[A_Q0], mine_at[A_Q0 + 1], mine_at[A_Q0 + 2], mine_at[A_Q0 + 3]};
        for (int t = 0; t < n_pend; t++) {
          const int c2 = pend[t].cls - 1;
          if (qs4[c2] < tp.q_stride) tp.q_list[(size_t)c2 * tp.q_stride + qs4[c2]] = 0xFFFFFFFFu;
          qs4[c2]++;
        }
      }
    }
    if (bail) {
      if (bail_team)
        tp.team[mine_at[A_BAILT]] = idx;
      else
        tp.bail[mine_at[A_BAIL]] = idx;
    }
    if (sleep) {
      tp.act_out[mine_at[A_ACT]] = (uint32_t)idx;
      // ---- the collected problems become records, queued by band class ----
      unsigned long long dpo = mine_at[A_DPO], qs4[4] = {mine_at[A_Q0], mine_at[A_Q0 + 1], mine_at[A_Q0 + 2], mine_at[A_Q0 + 3]};
      const unsigned long long w_rec = mine_at[A_REC];
      const unsigned w_n = (unsigned)n_pend;
      for (int t = 0; t < n_pend; t++) {
        const Pend& e = pend[t];
        DpRec d;
        d.x0 = e.x0;
        d.y0 = e.y0;
        d.ops_off = dpo;
        d.xlen = e.xlen;
        d.ylen = e.ylen;
        d.bw = (uint16_t)f_bw;
        d.xd = (uint16_t)min(f_xd, 65535);
        d.dir = e.dir;
        d.cls = e.cls;
        d.pad_ = 0;
        d.read = (uint32_t)idx;
        d.score = 0;
        d.xend = d.yend = d.nops = 0;
        d.done = 0;
        d.cells = d.cols = 0;
        d.pad2_ = 0;
        const uint32_t ri = (uint32_t)(w_rec + (unsigned)t);
        tp.recs[ri] = d;
        tp.q_list[(size_t)(e.cls - 1) * tp.q_stride + qs4[e.cls - 1]] = ri;
        qs4[e.cls - 1]++;
        dpo += ((unsigned)e.xlen + (unsigned)e.ylen + 3u) & ~3u;
      }
      ReadMemo m2 = memo;
      m2.base[memo.n_rounds] = (uint32_t)w_rec;
      m2.cnt[memo.n_rounds] = (uint16_t)w_n;
      m2.first_hit[memo.n_rounds] = (uint16_t)f_hno;
      m2.n_rounds = (uint8_t)(memo.n_rounds + 1);
      for (int k = 0; k < 7; k++) m2.pad_[k] = 0;
      tp.memos[idx] = m2;
    }

    unsigned long long opb = 0;
    unsigned ty[3] = {0, 0, 0};
    if (done) {
      unsigned long long my_off = mine_at[A_OPS];
      Cand* cands = p.cands + cand0;
      uint32_t* order = p.order + 2 * cand0;
      for (uint32_t t = 0; t < nres; t++) {
        const TCand& a = cd[res[t]];
        const int lead = a.xstart, trail = L - a.xend;
        const int lead5 = lead > 0 ? 5 : 0, trail5 = trail > 0 ? 5 : 0;
        const int nb = lead5 + a.nops + 5 * (int)a.n_y + trail5;
        const bool exonic = a.aln_type == THM_ALN_EXONIC;
        const int tnb = exonic ? lead5 + a.nops + trail5 : 0;
        PathView pv;
        pv.nl = a.nl;
        pv.len = a.len;
        pv.nops = a.nops;
        pv.l_sp = a.l_sp;
        pv.r_sp = a.r_sp;
        pv.l_ops = a.l_rec >= 0 ? tp.dp_ops + tp.recs[a.l_rec].ops_off : nullptr;
        pv.r_ops = a.r_rec >= 0 ? tp.dp_ops + tp.recs[a.r_rec].ops_off : nullptr;
        emit_stream(p.cand_ops + my_off, nb, pv, lead, trail, a.rev != 0, (int)a.n_y, a.mk_k, a.ycl);
        const unsigned long long goff = my_off;
        my_off += ((unsigned)nb + 3u) & ~3u;
        unsigned long long toff = 0;
        if (exonic) {
          emit_stream(p.cand_ops + my_off, tnb, pv, lead, trail, false, 0, a.mk_k, a.ycl);
          toff = my_off;
          my_off += ((unsigned)tnb + 3u) & ~3u;
        }
        Cand c;
        c.ystart = a.ch0;
        c.yend = a.ch1;
        c.ylen = a.ylen;
        c.ops_off = goff;
        c.ops_len = (uint32_t)nb;
        c.score = a.score;
        c.ref_id = a.ref_id;
        c.xstart = (uint32_t)a.xstart;
        c.xend = (uint32_t)a.xend;
        c.tx_or_gene_idx = a.type_idx;
        c.name_rank = a.name_rank;
        c.strand = a.strand;
        c.aln_type = a.aln_type;
        c.primary = 0;
        c.pad_ = 0;
        c.tx_ystart = c.tx_yend = c.tx_ylen = 0;
        c.tx_ops_off = 0;
        c.tx_ops_len = 0;
        c.tx_score = 0;
        c.tx_xstart = c.tx_xend = 0;
        if (exonic) {
          c.tx_ystart = (uint64_t)a.tx_ystart;
          c.tx_yend = (uint64_t)a.tx_yend;
          c.tx_ylen = a.tx_ylen;
          c.tx_ops_off = toff;
          c.tx_ops_len = (uint32_t)tnb;
          c.tx_score = a.score;
          c.tx_xstart = (uint32_t)a.xstart;
          c.tx_xend = (uint32_t)a.xend;
        }
        cands[t] = c;
        order[t] = t;
        opb += (unsigned long long)(nb + tnb);
        ty[0] += a.aln_type == THM_ALN_EXONIC;
        ty[1] += a.aln_type == THM_ALN_INTRONIC;
        ty[2] += a.aln_type == THM_ALN_INTERGENIC;
      }
      p.read_n_alns[idx] = nres;
      p.read_op_bytes[idx] = opb;
      tp.recs_rw[idx].len = 0xFFFFFFFFu;  // finished: the kernels behind this one skip the read
    }

    // ---- counters of the reads finished in this pass ----
    {
      auto add = [&](int slot, unsigned long long v) {
        for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2);
        if (lane == 0 && v) atomicAdd(&s_cnt[slot], v);
      };
      add(THM_CNT_READS, done ? 1ull : 0ull);
      add(THM_CNT_ALIGNED, (done && nres) ? 1ull : 0ull);
      add(THM_CNT_UNMAPPED, (done && !nres) ? 1ull : 0ull);
      add(THM_CNT_ALNS, done ? (unsigned long long)nres : 0ull);
      add(THM_CNT_EXONIC, done ? (unsigned long long)ty[0] : 0ull);
      add(THM_CNT_INTRONIC, done ? (unsigned long long)ty[1] : 0ull);
      add(THM_CNT_INTERGENIC, done ? (unsigned long long)ty[2] : 0ull);
      add(THM_CNT_SWG_CALLS, done ? (unsigned long long)calls : 0ull);
      add(THM_CNT_DP_CELLS, done ? dp_cells : 0ull);
      add(THM_CNT_DP_COLS, done ? dp_cols : 0ull);
      add(THM_CNT_OP_BYTES, done ? opb : 0ull);
      add(THM_CNT_WINDOW_BYTES, done ? (unsigned long long)winbytes : 0ull);
      if (tp.stats) {
        const unsigned long long mk = __ballot(mine && !done && !sleep);
        if (lane == 0 && mk) atomicAdd(&s_stats[0], (unsigned)__popcll(mk));
        for (int w = 1; w < 8; w++) {
          const unsigned long long m2 = __ballot(mine && !done && !sleep && why == w);
          if (lane == 0 && m2) atomicAdd(&s_stats[w], (unsigned)__popcll(m2));
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < THM_N_COUNTERS) {
    const unsigned long long v = s_cnt[threadIdx.x];
    if (v) p.wave_counters[(size_t)blockIdx.x * THM_N_COUNTERS + threadIdx.x] += v;  // the row is this workgroup's in every round
  }
  if (tp.stats && threadIdx.x < 8 && s_stats[threadIdx.x]) atomicAdd(&tp.stats[threadIdx.x], (unsigned long long)s_stats[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------------------
// The reads the control kernel takes (fast class, fewer than max_hits hits), by descending hit count: the threads of a
// wavefront walk the hits of 64 reads in lockstep, so a wavefront takes as long as its read with the most hits --
// reads of like hit counts belong together, and the long ones go first.  Two launches: phase 0 counts the reads per
// hit count, phase 1 places them (order within a hit count: as the workgroups come).
// bins: [0, 64) counts, [64, 128) cursors.
// ---------------------------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256) void tpr_order_kernel(const ReadRecT<C>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins,
                                                        uint32_t* out, unsigned long long* n_out, const int* fault_seed, int phase) {
  __shared__ unsigned h[64], base[64];
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  int b = -1;
  if (r < n && *fault_seed == 0) {
    const uint32_t len = recs[r].len, nh = recs[r].n_hits;
    if (len <= max_len && nh < max_hits) b = (int)min(nh, 63u);
  }
  unsigned pos = 0;
  if (b >= 0) pos = atomicAdd(&h[b], 1u);
  __syncthreads();
  if (phase == 0) {
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], (unsigned long long)h[threadIdx.x]);
    return;
  }
  if (threadIdx.x < 64) {
    // reads with more hits come first: the bin's start is the number of reads with more hits
    unsigned long long before = 0, total = 0;
    for (int k = 0; k < 64; k++) {
      const unsigned long long c = bins[k];
      if (k > (int)threadIdx.x) before += c;
      total += c;
    }
    base[threadIdx.x] = (unsigned)(before + (h[threadIdx.x] ? atomicAdd(&bins[64 + threadIdx.x], (unsigned long long)h[threadIdx.x]) : 0ull));
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = total;
  }
  __syncthreads();
  if (b >= 0) out[base[b] + pos] = (uint32_t)r;
}

// ---------------------------------------------------------------------------------------------------------------------
// DP kernel: one request per wavefront.  The requests queued since the last round, band class by band class:
// q_list[c][q_done[c] .. q_cur[c]).  A wave's first chunk of a class is its own by position (no atomic: a launch
// that finds nothing to do costs nothing), further chunks come from the class's work counter.
// ---------------------------------------------------------------------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void dp_class(const DpParams& p, uint8_t* xs, uint8_t* ys, unsigned long long* trace_lds, uint8_t* opsb, uint32_t ops_cap,
                                         unsigned wave_global, unsigned n_waves, int& fault) {
  // the trace of a one-cell-per-lane problem (16 bytes per column) is in LDS; the rare wider ones keep theirs in a
  // wave-private slice of global memory, so that the LDS footprint -- hence the waves per CU -- is set by the common case
  unsigned long long* trace = CPL == 1 ? trace_lds : p.trace_scratch + (size_t)wave_global * p.trace_per_wave;
  const int lane = lane_id();
  // (the cursor may have run past the list when the request pools were exhausted: those reads went to the wave-per-read
  // kernel and their slots, as far as they exist, hold 0xFFFFFFFF)
  const unsigned long long q0 = min(p.q_done[CPL - 1], (unsigned long long)p.q_stride), q1 = min(p.q_cur[CPL - 1], (unsigned long long)p.q_stride);
  if (q0 >= q1) return;
  const uint32_t* list = p.q_list + (size_t)(CPL - 1) * p.q_stride;
  // Requests are handed out by position, wave w takes w, w + n_waves, ...: a request is one extension of a few dozen
  // columns, a wave gets dozens of them, so the shares even out -- and there is no work counter (one hot word serves
  // about 88 M returning atomics per second: 175 000 chunks of two took 2 ms, whatever the occupancy).
  (void)fault;
  for (unsigned long long q_next = q0 + wave_global; q_next < q1; q_next += n_waves) {
    const uint32_t ri = list[q_next];
    if (ri == 0xFFFFFFFFu) continue;
    const DpRec rq = p.recs[ri];
    const int xlen = (int)bcast_first((int)rq.xlen), ylen = (int)bcast_first((int)rq.ylen);
    const int bw = (int)bcast_first((int)rq.bw), xd = (int)bcast_first((int)rq.xd);
    const int dir = (int)bcast_first((int)rq.dir);
    if ((uint32_t)xlen + 64u > p.x_cap || (uint32_t)ylen + 64u > p.y_cap || min(2 * bw + 1, xlen + 1) > 64 * CPL) {
      fault |= 1;
      continue;
    }
#pragma unroll 1
    for (int t = lane; t < xlen; t += 64) xs[t] = rq.x0[t * dir];
#pragma unroll 1
    for (int t = lane; t < ylen; t += 64) ys[t] = rq.y0[t * dir];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    SwgResult r = swg_extend_wave<CPL>(xs, 1, xlen, ys, 1, ylen, bw, xd, trace);
    if (CPL == 1)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    else
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // lane 0's trace stores must be visible to the loads of all lanes
    int nops = swg_traceback_wave<CPL>(trace, r.xend, r.yend, bw, opsb, 1, (int)ops_cap);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (nops < 0 || nops > xlen + ylen) {
      fault |= nops < 0 ? 2 : 4;
      nops = 0;
    }
    uint8_t* out = p.dp_ops + rq.ops_off;
#pragma unroll 1
    for (int t = lane; t < nops; t += 64) out[t] = opsb[t];
    if (lane == 0) {
      DpRec* d = p.recs + ri;
      d->score = r.score;
      d->xend = (uint16_t)r.xend;
      d->yend = (uint16_t)r.yend;
      d->nops = (uint16_t)nops;
      d->cells = r.cells;  // counted by the control kernel when (and if) the read is finished with this result
      d->cols = r.cols;
      d->done = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

// CPLMAX: the widest band class of the run (the LDS trace is sized for it)
template <int CPLMAX>
__global__ __launch_bounds__(256) void extend_dp_kernel(DpParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = bcast_first((int)(threadIdx.x >> 6));
  const unsigned wave_global = blockIdx.x * 4u + (unsigned)wave, n_waves = gridDim.x * 4u;
  const uint32_t tr_bytes = (p.y_cap + 1) * 16;
  const uint32_t ops_cap = p.x_cap + p.y_cap + 16;
  const uint32_t per_wave = p.x_cap + p.y_cap + tr_bytes + ops_cap;
  uint8_t* xs = smem + (size_t)wave * per_wave;
  uint8_t* ys = xs + p.x_cap;
  unsigned long long* trace = (unsigned long long*)(ys + p.y_cap);
  uint8_t* opsb = (uint8_t*)trace + tr_bytes;
  int fault = 0;
  dp_class<1>(p, xs, ys, trace, opsb, ops_cap, wave_global, n_waves, fault);
  if constexpr (CPLMAX >= 2) dp_class<2>(p, xs, ys, trace, opsb, ops_cap, wave_global, n_waves, fault);
  if constexpr (CPLMAX >= 3) dp_class<3>(p, xs, ys, trace, opsb, ops_cap, wave_global, n_waves, fault);
  if constexpr (CPLMAX >= 4) dp_class<4>(p, xs, ys, trace, opsb, ops_cap, wave_global, n_waves, fault);
  if (lane_id() == 0 && fault) atomicOr(p.fault, 2 | (fault << 4));  // FAULT_INTERNAL (kernels_extend.hip) + which check (diagnosis)
}

}  // namespace dev

size_t extend_dp_lds_bytes(uint32_t x_cap, uint32_t y_cap) {
  const size_t tr = (size_t)(y_cap + 1) * 16;
  return 4 * ((size_t)x_cap + y_cap + tr + x_cap + y_cap + 16);
}
// global trace scratch of one wave (problems of more than 64 band slots), in bytes
size_t extend_dp_trace_bytes(uint32_t y_cap, int cpl_max) { return cpl_max > 1 ? (size_t)(y_cap + 2) * (size_t)cpl_max * 16 : 0; }

template <class C>
static hipError_t launch_extend_ctl_t(const ExtendParamsT<C>& p, const TprParamsT<C>& tp, int n_blocks, hipStream_t s) {
  if (n_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(dev::extend_ctl_kernel<C>, dim3(n_blocks), dim3(256), 0, s, p, tp);
  return hipGetLastError();
}
hipError_t launch_extend_ctl(const ExtendParamsT<uint32_t>& p, const TprParamsT<uint32_t>& tp, int n_blocks, hipStream_t s) {
  return launch_extend_ctl_t(p, tp, n_blocks, s);
}
hipError_t launch_extend_ctl(const ExtendParamsT<uint64_t>& p, const TprParamsT<uint64_t>& tp, int n_blocks, hipStream_t s) {
  return launch_extend_ctl_t(p, tp, n_blocks, s);
}

template <class C>
static hipError_t launch_tpr_order_t(const ReadRecT<C>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins, uint32_t* out,
                                     unsigned long long* n_out, const int* fault_seed, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(dev::tpr_order_kernel<C>, dim3(blocks), dim3(256), 0, s, recs, n, max_len, max_hits, bins, out, n_out, fault_seed, 0);
  hipLaunchKernelGGL(dev::tpr_order_kernel<C>, dim3(blocks), dim3(256), 0, s, recs, n, max_len, max_hits, bins, out, n_out, fault_seed, 1);
  return hipGetLastError();
}
hipError_t launch_tpr_order(const ReadRecT<uint32_t>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins, uint32_t* out,
                            unsigned long long* n_out, const int* fault_seed, hipStream_t s) {
  return launch_tpr_order_t(recs, n, max_len, max_hits, bins, out, n_out, fault_seed, s);
}
hipError_t launch_tpr_order(const ReadRecT<uint64_t>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins, uint32_t* out,
                            unsigned long long* n_out, const int* fault_seed, hipStream_t s) {
  return launch_tpr_order_t(recs, n, max_len, max_hits, bins, out, n_out, fault_seed, s);
}

hipError_t launch_extend_dp(const DpParams& p, int cpl_max, int n_blocks, hipStream_t s) {
  const size_t lds = extend_dp_lds_bytes(p.x_cap, p.y_cap);
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(256), lds, s, p);
    return hipGetLastError();
  };
  switch (cpl_max) {
    case 1: return go(dev::extend_dp_kernel<1>);
    case 2: return go(dev::extend_dp_kernel<2>);
    case 3: return go(dev::extend_dp_kernel<3>);
    case 4: return go(dev::extend_dp_kernel<4>);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace thm
