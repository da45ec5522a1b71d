// kernels_extend.hip -- the per-read aligner: one read per wavefront.
//
// Restates aligner::align_read (reference src/aligner.rs:123-190) and everything
// below it on the device:
//     align_seed_hit        src/aligner.rs:198-314
//     extend_left_right     src/aligner.rs:352-407   (two SwgExtend::extend calls, swg_device.h)
//     extend_seed_match     src/aligner.rs:410-426
//     concat_to_chr_aln     src/aligner.rs:429-449
//     filter_overlapping    src/aligner.rs:317-349
//     lift_mem_to_tx        src/txome.rs:82-103
//     lift_tx_to_gx         src/txome.rs:110-160
//     Index::idx_to_ref     src/index.rs:287-290
//     Index::seq_slice      src/index.rs:304-323     (a plain slice of the text: both strands are stored)
//     IntervalTree::find    bio 0.37.1, answered from the interval grids in the same yield order
//
// The hits of one read are processed strictly in the reference's order because
// band_width / x_drop / max_aln_score are loop-carried (src/aligner.rs:143-175).
// Reads are independent, so the parallelism is: reads over wavefronts, band
// cells over lanes.  Control flow is wave-uniform and wave-uniform values are kept
// on the scalar unit (readfirstlane) so that tree nodes, exons and transcript
// records come through the scalar cache and do not occupy vector registers.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.h"
#include "swg_device.h"

namespace thm {
namespace dev {

// FAULT_OPS_POOL / FAULT_INTERNAL go to the batch's fault word; FAULT_CONTRACT (a condition that panics in the
// reference), FAULT_RETRY (more introns in one alignment than the fast kernel's marker list holds) and FAULT_BAND are per read
// FAULT_BAND (per read): the read's band does not fit this launch's class (cannot happen while the band is monotone in the
// read length, which is how the classes are cut; the reference's own check, assert!(band_width <= max_band_width) at
// src/swg.rs:32, is per call): the read gets the status THM_ERR_INTERNAL and no alignments, the batch goes on
enum : int { FAULT_OPS_POOL = 1, FAULT_INTERNAL = 2, FAULT_CONTRACT = 4, FAULT_RETRY = 8, FAULT_BAND = 16 };

// signed type that holds a text coordinate of width C and small negative offsets from it
template <class C>
struct CoordTraits {
  typedef int S;
};
template <>
struct CoordTraits<uint64_t> {
  typedef long long S;
};
__device__ __forceinline__ uint32_t readlane_c(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t readlane_c(uint64_t v, int l) {
  return ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffu), l);
}

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v) {
  return ((unsigned long long)(unsigned)bcast_first((int)(v >> 32)) << 32) | (unsigned)bcast_first((int)(v & 0xffffffffu));
}
__device__ __forceinline__ void wfence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

template <class C>
struct RefInfoT {
  C start, end, len;  // C = uint32_t (text below 2^31 symbols) keeps this arithmetic on the 32-bit scalar unit
  uint32_t id;
  uint32_t name_rank;
  bool strand;
};
// Index::idx_to_ref: refs.partition_point(|x| x.end_idx <= idx) -- answered in two loads: the contig copy that
// holds the first symbol of idx's bin, then forward over the (rare) boundaries inside the bin
template <class C, class IX>
__device__ RefInfoT<C> idx_to_ref(const IX& ix, C idx) {
  uint32_t lo = uload(&ix.ref_bin[idx >> GRID_SHIFT]);
  RefRecT<C> r = uload(&ix.ref_recs[lo]);
  while (r.end <= idx && lo + 1 < ix.n_refs) {
    lo++;
    r = uload(&ix.ref_recs[lo]);
  }
  RefInfoT<C> o;
  o.start = r.start;
  o.end = r.end;
  o.len = r.len;
  o.id = lo;
  o.name_rank = r.name_rank;
  o.strand = r.strand != 0;
  return o;
}

// What retain / filter_overlapping / the final sort (src/aligner.rs:177-187) look at, per accepted candidate: kept in
// LDS for the first KEYCAP candidates of a read, so that the usual multi-candidate read (a handful) is finished
// from registers without going back to the candidate array in global memory.
struct CandKey {
  uint64_t ystart, yend;
  int32_t score;
  uint32_t name_rank;
  uint32_t bytes;        // serialised op bytes (genome + transcript streams)
  uint32_t strand_type;  // strand | aln_type << 8
};
static_assert(sizeof(CandKey) == 32, "CandKey layout");
constexpr int KEYCAP = 16;

// wave-private buffers: LDS in the register-resident kernels, a slice of global memory in the any-width
// kernel (GS).  Lanes exchange data through them, so every exchange is followed by wsync(): a wavefront-scope
// fence for LDS, a workgroup-scope one (waits for the stores; same-CU L1 is coherent) for global memory.
template <bool GS_>
struct WctxT {
  static constexpr bool GS = GS_;
  uint8_t* rd;   // sanitised read, zero padded
  uint8_t* win;   // window the extension reads (genome or transcript; 16-byte aligned copy)
  uint8_t* wing;  // the hit's genome window, kept while its transcripts are tried
  unsigned long long* trace;    // LDS: trace of a one-cell-per-lane extension (16 bytes per column)
  unsigned long long* trace_g;  // global memory: trace of a wider extension (rare for 91 bp reads; keeps the LDS footprint small)
  uint8_t* pa;  // three path buffers (op kinds 0..3), rotated by pointer swap
  uint8_t* pb;
  uint8_t* pc;
  int* mk_k;      // intron markers of the alignment being emitted: op index they precede ...
  uint32_t* ycl;  // ... and their lengths
  int mk_cap;
  CandKey* ck;    // keys of the first KEYCAP accepted candidates (register-resident kernels only)
  int* dp;        // any-width kernel: column state of swg_extend_tiled (4 arrays of dp_stride ints)
  int dp_stride;
  int L, opcap, wcap;
  unsigned cells, cols, calls, winbytes;
  int fault;
  unsigned long long pool_off;  // wave-private slice of the global op pool (bump allocated)
  unsigned pool_left;
#ifdef THM_PROF
  unsigned long long prof_last;
  unsigned long long prof_acc[10];
  unsigned long long prof_cols[6];
  int prof_hit, prof_tx;  // index of the hit within its read, 1 while a transcript target is extended
#endif
};
template <class W>
__device__ __forceinline__ void wsync(const W&) {
  if (W::GS)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  else
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Section timing for tuning builds (-DTHM_PROF, libthermite_amd_prof.so): every
// mark charges the shader clocks since the previous mark to one slot.
enum { PS_SETUP = 0, PS_STAGE = 1, PS_DP = 2, PS_TRACEBACK = 3, PS_TREE = 4, PS_TXPREP = 5, PS_LIFT = 6, PS_EMIT = 7,
       PS_FINAL = 8, PS_OTHER = 9 };
#ifdef THM_PROF
#define PROF_MARK(c, slot)                                       \
  do {                                                           \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    (c).prof_acc[slot] += t_ - (c).prof_last;                    \
    (c).prof_last = t_;                                          \
  } while (0)
#else
#define PROF_MARK(c, slot) \
  do {                     \
  } while (0)
#endif

// One query of an interval grid (thermite_internal.h): the intervals overlapping
// [qs, qe) come out in IntervalTree::find order, i.e. by ascending pre-order rank.
// Up to 64 candidate entries sit one per lane in registers; larger candidate sets
// (dense loci) are re-read from memory on every step.
template <class C, class E>
struct GridQueryT {
  const E* ent;  // candidates [0, cnt): GridEntryT<C> (genes) or ExonEntryT<C> (exons)
  uint32_t cnt;
  C qs, qe;
  uint32_t b0;
  int last;     // rank of the interval yielded last (-1 before the first)
  int my_rank;  // this lane's candidate: its rank if it overlaps and is the primary copy, else -1
  uint32_t my_val;
};
template <class C, class E>
__device__ __forceinline__ int grid_entry_rank(const E& e, C qs, C qe, uint32_t b0) {
  const bool overlap = qs < e.end && e.start < qe;
  const uint32_t home = max(b0, (uint32_t)(e.start >> GRID_SHIFT));  // first queried bin this interval is listed in
  const bool primary = (e.rank & 0xffu) == (home & 0xffu);
  return (overlap && primary) ? (int)(e.rank >> 8) : -1;
}
template <class C, class E>
__device__ void grid_begin(GridQueryT<C, E>& g, const uint32_t* off, const E* entries, C qs, C qe) {
  const uint32_t b0 = (uint32_t)(qs >> GRID_SHIFT), b1 = (uint32_t)((qe > qs ? qe - 1 : qs) >> GRID_SHIFT);
  const uint32_t e0 = uload(&off[b0]), e1 = uload(&off[b1 + 1]);
  g.ent = entries + e0;
  g.cnt = e1 - e0;
  g.qs = qs;
  g.qe = qe;
  g.b0 = b0;
  g.last = -1;
  g.my_rank = -1;
  g.my_val = 0;
  if (g.cnt <= 64 && (uint32_t)lane_id() < g.cnt) {
    const E* e = &g.ent[lane_id()];
    // the four leading fields only (an exon entry carries more: read by the caller for the one entry that wins)
    struct Head {
      C start, end;
      uint32_t value, rank;
    } h;
    h.start = e->start;
    h.end = e->end;
    h.value = e->value;
    h.rank = e->rank;
    g.my_rank = grid_entry_rank<C>(h, qs, qe, b0);
    g.my_val = h.value;
  }
}
// next overlapping interval in yield order (its value and its index among the candidates); false when exhausted
template <class C, class E>
__device__ bool grid_next(GridQueryT<C, E>& g, uint32_t& value, uint32_t& ent_idx) {
  const int BIG = 0x0fffffff;
  if (g.cnt == 0) return false;
  if (g.cnt <= 64) {
    const int cand = (g.my_rank > g.last) ? g.my_rank : BIG;
    const int best = wave_min(cand);
    if (best == BIG) return false;
    const unsigned long long m = __ballot(g.my_rank == best);
    ent_idx = (uint32_t)__builtin_ctzll(m);
    value = (uint32_t)__builtin_amdgcn_readlane((int)g.my_val, (int)ent_idx);
    g.last = best;
    return true;
  }
  int best = BIG;
  uint32_t bval = 0, bidx = 0;
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < g.cnt; c0 += 64) {
    const uint32_t i = c0 + (uint32_t)lane_id();
    int r = -1;
    uint32_t v = 0;
    if (i < g.cnt) {
      const E* e = &g.ent[i];
      struct Head {
        C start, end;
        uint32_t value, rank;
      } h;
      h.start = e->start;
      h.end = e->end;
      h.value = e->value;
      h.rank = e->rank;
      r = grid_entry_rank<C>(h, g.qs, g.qe, g.b0);
      v = h.value;
    }
    const int cand = (r > g.last) ? r : BIG;
    const int cb = wave_min(cand);
    if (cb < best) {
      best = cb;
      const unsigned long long m = __ballot(cand == cb);
      const int wl = (int)__builtin_ctzll(m);
      bval = (uint32_t)__builtin_amdgcn_readlane((int)v, wl);
      bidx = c0 + (uint32_t)wl;
    }
  }
  if (best == BIG) return false;
  value = bval;
  ent_idx = bidx;
  g.last = best;
  return true;
}

template <class S>
struct PathT {
  int score, xstart, xend, nops;
  S ystart, yend;  // in the coordinates r / lo_abs were given in
};

// what extend_lr did for the hit's genome window, for reuse by its transcripts
struct LrMemo {
  SwgResult R, Lt;
  int nr, nl;
  int yr, yl;    // columns that were available to the right / left extension
  int yoff_r;    // offset in the window buffer of y[0] of the right extension
  int yoff_l;    // offset of y[0] of the left extension (which walks backwards)
};

// One SwgExtend::extend + trace.  The band slots that can ever hold a cell number
// min(2*bw+1, |x|+1): when that fits 64 the one-cell-per-lane code is exact even
// inside a kernel compiled for a wider band (slots >= |x|+1 are never valid), and
// it issues half the instructions per column.  CPL == 0: the any-width kernel (band in
// tiles, everything in global memory).
template <int CPL, class W>
__device__ __forceinline__ int swg_and_trace(W& c, const uint8_t* xs, int dx, int xlen, const uint8_t* ys, int dy, int ylen,
                                             int bw, int xd, uint8_t* ops, int stride, int max_ops, SwgResult& r) {
  int n;
#ifndef THM_NO_SHORTCUT
  if (swg_one_mismatch_shortcut(xs, dx, xlen, ys, dy, ylen, xd, ops, stride, max_ops, r, n)) {
    wsync(c);
    PROF_MARK(c, PS_DP);
    return n;
  }
#endif
  if constexpr (CPL == 0) {
    r = swg_extend_tiled(xs, dx, xlen, ys, dy, ylen, bw, xd, c.trace_g, c.dp, c.dp_stride);
    tiled_sync();
    PROF_MARK(c, PS_DP);
    n = swg_traceback_tiled(c.trace_g, r.xend, r.yend, bw, ops, stride, max_ops);
  } else if (CPL > 1 && min(2 * bw + 1, xlen + 1) <= 64) {
    r = swg_extend_wave<1>(xs, dx, xlen, ys, dy, ylen, bw, xd, c.trace);
    wsync(c);
    PROF_MARK(c, PS_DP);
    n = swg_traceback_wave<1>(c.trace, r.xend, r.yend, bw, ops, stride, max_ops);
  } else if (CPL > 2 && min(2 * bw + 1, xlen + 1) <= 128) {
    // fits two cells per lane: a third fewer instructions per column than the kernel's own width, and the trace stays in
    // LDS (ext_caps: these kernels' LDS trace holds two cells per lane)
    unsigned long long* tr = c.trace;
    r = swg_extend_wave<2>(xs, dx, xlen, ys, dy, ylen, bw, xd, tr);
    wsync(c);
    PROF_MARK(c, PS_DP);
    n = swg_traceback_wave<2>(tr, r.xend, r.yend, bw, ops, stride, max_ops);
  } else {
    // more than 64 band slots: the trace (CPL * 16 bytes per column) goes to the wave's scratch in global
    // memory; the stores of lane 0 must be visible to the loads of all lanes, hence the agent-scope fence
    constexpr int K = CPL > 0 ? CPL : 1;
    unsigned long long* tr = (K > 1) ? c.trace_g : c.trace;
    r = swg_extend_wave<K>(xs, dx, xlen, ys, dy, ylen, bw, xd, tr);
    if (K > 1)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    else
      wsync(c);
    PROF_MARK(c, PS_DP);
    n = swg_traceback_wave<K>(tr, r.xend, r.yend, bw, ops, stride, max_ops);
  }
  wsync(c);
  PROF_MARK(c, PS_TRACEBACK);
  return n;
}

// extend_left_right, reference src/aligner.rs:352-407.  `win` holds ref_seq bytes
// from absolute coordinate win0; ref_seq itself spans [lo_abs, hi_abs).
template <int CPL, class S, class W>
__device__ PathT<S> extend_lr(W& c, const uint8_t* win, S win0, S lo_abs, S hi_abs, S r, int q, int len, int bw, int xd,
                              uint8_t* buf, LrMemo& memo) {
  const int L = c.L;
  PathT<S> p;
  PROF_MARK(c, PS_OTHER);
  // right: x = read[q+len..], y = ref_seq[r+len..]   (:360-362)
  const int xr = L - (q + len);
  const S yr_avail = hi_abs - (r + len);
  const int yr = (int)min(yr_avail, (S)(xr + bw + 1));
  SwgResult R, Lt;
  int nr = swg_and_trace<CPL>(c, c.rd + q + len, 1, xr, win + (int)(r + len - win0), 1, yr, bw, xd, buf + c.opcap - 1, -1,
                              c.opcap, R);
  // left: both reversed, y = ref_seq[r.saturating_sub(L+bw)..r]   (:364-375)
  const int xl = q;
  const S rel = r - lo_abs;
  const S y0 = lo_abs + (rel > L + bw ? rel - (L + bw) : (S)0);
  const int yl = (int)min(r - y0, (S)(xl + bw + 1));
  int nl = swg_and_trace<CPL>(c, c.rd + q - 1, -1, xl, win + (int)(r - 1 - win0), -1, yl, bw, xd, buf, 1,
                              c.opcap - max(nr, 0), Lt);
  c.cells += R.cells + Lt.cells;
  c.cols += R.cols + Lt.cols;
  c.calls += 2;
#ifdef THM_PROF
  {  // how much of the DP runs on at most 32 band slots (two such problems could share a wave)
    const bool nr_ = min(2 * bw + 1, xr + 1) <= 32, nl_ = min(2 * bw + 1, xl + 1) <= 32;
    c.prof_cols[0] += R.cols + Lt.cols;
    c.prof_cols[1] += (nr_ ? R.cols : 0) + (nl_ ? Lt.cols : 0);
    c.prof_cols[2] += (nr_ && nl_) ? min(R.cols, Lt.cols) : 0;
    c.prof_cols[3] += (c.prof_hit == 0) ? R.cols + Lt.cols : 0;
    c.prof_cols[4] += c.prof_tx ? R.cols + Lt.cols : 0;
    // extensions of the shape "one mismatch, then exact to the end of x" (their result is known without DP)
    {
      const uint8_t* yR = win + (int)(r + len - win0);
      const uint8_t* yL = win + (int)(r - 1 - win0);
      bool dr = false, dl = false;
      for (int t0 = 1; t0 < xr; t0 += 64) {
        const int t = t0 + lane_id();
        dr = dr || (t < xr && (t >= yr || c.rd[q + len + t] != yR[t]));
      }
      for (int t0 = 1; t0 < xl; t0 += 64) {
        const int t = t0 + lane_id();
        dl = dl || (t < xl && (t >= yl || c.rd[q - 1 - t] != yL[-t]));
      }
      const bool m1r = xr >= 3 && yr >= xr && __ballot(dr) == 0ull, m1l = xl >= 3 && yl >= xl && __ballot(dl) == 0ull;
      c.prof_cols[5] += (m1r ? R.cols : 0) + (m1l ? Lt.cols : 0);
    }
  }
#endif
  if (nr < 0 || nl < 0 || nl + len + nr > c.opcap) {
    c.fault |= FAULT_INTERNAL;
    nl = 0;
    nr = 0;
  }
  const int lane = lane_id();
  // rev(left.ops) ++ Match x len ++ right.ops   (:388-394); the clips are implied by xstart / xend
  #pragma unroll 1
  for (int t = lane; t < len; t += 64) buf[nl + t] = OPK_MATCH;
  #pragma unroll 1
  for (int t0 = 0; t0 < nr; t0 += 64) {
    const int t = t0 + lane;
    uint8_t v = 0;
    if (t < nr) v = buf[c.opcap - nr + t];
    wsync(c);
    if (t < nr) buf[nl + len + t] = v;
    wsync(c);
  }
  PROF_MARK(c, PS_TRACEBACK);
  memo.R = R;
  memo.Lt = Lt;
  memo.nr = nr;
  memo.nl = nl;
  memo.yr = yr;
  memo.yl = yl;
  memo.yoff_r = (int)(r + len - win0);
  memo.yoff_l = (int)(r - 1 - win0);
  p.nops = nl + len + nr;
  p.score = Lt.score + len * MATCH_SCORE + R.score;
  p.ystart = r - Lt.yend;
  p.yend = r + len + R.yend;
  p.xstart = q - Lt.xend;
  p.xend = q + len + R.xend;
  return p;
}

// Stage [a, b) of a global byte array into c.win with 16-byte loads.  Returns the
// coordinate that c.win[0] corresponds to (a rounded down to the 16-byte grid of
// the source address; the arrays carry 16 bytes of padding at both ends of use).
template <class W>
__device__ int stage_window(W& c, uint8_t* dst, const uint8_t* src, int a, int b) {
  const unsigned mis = (unsigned)((uintptr_t)(src + a) & 15u);
  const int n = (b - a) + (int)mis;
  if (n > c.wcap) {
    c.fault |= FAULT_INTERNAL;
    return a;
  }
  const uint4* g = (const uint4*)(src + a - mis);
  uint4* w = (uint4*)dst;
  for (int t = lane_id(); t * 16 < n; t += 64) w[t] = g[t];
  c.winbytes += (unsigned)(b - a);
  wsync(c);
  PROF_MARK(c, PS_STAGE);
  return a - (int)mis;
}

// lift_tx_to_gx, reference src/txome.rs:110-160, without materialising the lifted
// list: returns the introns as markers (c.mk_k[m] = index of the path op the
// Yclip precedes, nops = "after the last op"; c.ycl[m] = its length) and the
// lifted start / end.  A Yclip is pushed before the first op at which the
// transcript position has reached an exon end (:133-141); the reference's op list
// includes a trailing Xclip when the read is clipped on the right, which gets its
// own loop iteration -- so an alignment that ends exactly on an exon boundary
// still receives the intron (the edge case noted at src/txome.rs:132).
template <class S, class W, class IX>
__device__ int lift_markers(W& c, const IX& ix, const thm_tx& tx, const uint8_t* path, int n, bool trailing_clip,
                            int ystart, int yend, S& gx_ystart, S& gx_yend) {
  const thm_exon* ex = ix.exons + tx.exon_begin;
  const uint64_t* toff = ix.exon_txoff + tx.exon_begin;
  const int ne = (int)tx.n_exons;
  const int lane = lane_id();
  // exon where the alignment starts: while exon_sum + len <= i  (:123-126)
  int lo = 0, hi = ne;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const thm_exon xm = uload(&ex[mid]);
    if ((int)(uload(&toff[mid]) + (xm.end - xm.start)) <= ystart)
      lo = mid + 1;
    else
      hi = mid;
  }
  int e = lo;
  if (e >= ne) {  // index panic in the reference
    c.fault |= FAULT_CONTRACT;
    gx_ystart = gx_yend = 0;
    return 0;
  }
  thm_exon cur = uload(&ex[e]);
  int exon_sum = (int)uload(&toff[e]);
  gx_ystart = (S)cur.start + (S)(ystart - exon_sum);
  // transcript positions advance on Match / Subst / Del; total must equal yend - ystart (:154)
  int n_adv = 0;
  #pragma unroll 1
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    const uint8_t op = (k < n) ? path[k] : (uint8_t)OPK_INS;
    n_adv += __popcll(__ballot(op != OPK_INS));
  }
  if (n_adv != yend - ystart) c.fault |= FAULT_CONTRACT;
  int n_y = 0;
  for (;;) {
    const int bnd = exon_sum + (int)(cur.end - cur.start);  // transcript offset of this exon's end
    if (e + 1 >= ne || bnd > yend) break;
    // index of the op that follows the advancing op which brings the position to bnd
    const int need = bnd - ystart;  // 1-based rank among advancing ops
    int kstar = -1, seen = 0;
    for (int k0 = 0; k0 < n && kstar < 0; k0 += 64) {
      const int k = k0 + lane;
      const uint8_t op = (k < n) ? path[k] : (uint8_t)OPK_INS;
      const unsigned long long m = __ballot(op != OPK_INS);
      const int cnt = __popcll(m);
      if (seen + cnt >= need) {
        const int rank = need - seen;  // rank within this chunk, >= 1
        const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const bool mine = ((m >> lane) & 1ull) && ((int)__popcll(m & le) == rank);
        const unsigned long long sel = __ballot(mine);
        kstar = k0 + __builtin_ctzll(sel) + 1;
      }
      seen += cnt;
    }
    if (kstar < 0) break;                        // cannot happen when n_adv is consistent
    if (kstar >= n && !trailing_clip) break;     // boundary reached by the very last op: no further iteration
    if (n_y >= c.mk_cap) {
      // more introns than the marker list holds: the any-width kernel (list sized by the longest transcript) takes the read
      c.fault |= W::GS ? FAULT_INTERNAL : FAULT_RETRY;
      break;
    }
    const thm_exon nxt = uload(&ex[e + 1]);
    if (lane == 0) {
      c.mk_k[n_y] = kstar;
      c.ycl[n_y] = (uint32_t)(nxt.start - cur.end);
    }
    n_y++;
    exon_sum = bnd;
    cur = nxt;
    e++;
  }
  gx_yend = (S)cur.start + (S)(yend - exon_sum);
  wsync(c);
  PROF_MARK(c, PS_LIFT);
  return n_y;
}

// Serialise [Xclip(xstart)] path-with-introns [Xclip(L-xend)] straight into the
// global op pool, every lane writing its own bytes; `reverse` mirrors the whole
// list (concat_to_chr_aln on a reverse-strand Ref, src/aligner.rs:440-447).
// Returns the pool offset, byte count in n_bytes.
template <class W, class PP>
__device__ unsigned long long emit_alignment(W& c, const PP& p, const uint8_t* path, int n, int xstart, int xend,
                                             bool reverse, int n_y, int& n_bytes) {
  const int lane = lane_id();
  const int lead = xstart, trail = c.L - xend;
  const int lead5 = lead > 0 ? 5 : 0, trail5 = trail > 0 ? 5 : 0;
  const int total = lead5 + n + 5 * n_y + trail5;
  n_bytes = total;
  // the op pool is handed out to waves in 4 KiB slices (one atomic per slice, not per alignment)
  if ((unsigned)total > c.pool_left) {
    const unsigned grab = max(4096u, (unsigned)total);
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.ops_cursor, (unsigned long long)grab);
    g = bcast64(g);
    if (g + grab > p.cand_ops_cap) {
      // pool exhausted: the host grows it and replays the batch; this attempt must not leave byte
      // counts behind that exceed what was allocated (the scans and compact_kernel still run)
      c.fault |= FAULT_OPS_POOL;
      c.pool_left = 0;
      n_bytes = 0;
      return 0;
    }
    c.pool_off = g;
    c.pool_left = grab;
  }
  const unsigned long long off = c.pool_off;
  c.pool_off += (unsigned)total;
  c.pool_left -= (unsigned)total;
  uint8_t* o = p.cand_ops + off;
  auto put5 = [&](int pos, uint8_t kind, uint32_t v) {
    o[pos] = kind;
    o[pos + 1] = (uint8_t)v;
    o[pos + 2] = (uint8_t)(v >> 8);
    o[pos + 3] = (uint8_t)(v >> 16);
    o[pos + 4] = (uint8_t)(v >> 24);
  };
  // forward byte position of an element; the reversed position is total - (pos + size)
  #pragma unroll 1
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    if (k < n) {
      int before = 0;
      for (int m = 0; m < n_y; m++) before += (c.mk_k[m] <= k) ? 1 : 0;
      const int pos = lead5 + k + 5 * before;
      o[reverse ? total - (pos + 1) : pos] = path[k];
    }
  }
  #pragma unroll 1
  for (int m = lane; m < n_y; m += 64) {
    const int pos = lead5 + c.mk_k[m] + 5 * m;
    put5(reverse ? total - (pos + 5) : pos, THM_OP_YCLIP, c.ycl[m]);
  }
  if (lane == 0 && lead > 0) put5(reverse ? total - 5 : 0, THM_OP_XCLIP, (uint32_t)lead);
  if (lane == 1 && trail > 0) put5(reverse ? 0 : total - 5, THM_OP_XCLIP, (uint32_t)trail);
  return off;
}

#ifndef THM_READ_RUN
#define THM_READ_RUN 4
#endif
constexpr int READ_RUN = THM_READ_RUN;  // consecutive reads per queue atomic of the wave-per-read kernels
constexpr int TEAM_MAX_CHUNKS = 16384;  // chunks a team keeps book of (reads beyond that stay with the sequential path)
constexpr int TEAM_CHUNK = 4;           // hits per chunk of a team

// per-wave buffer sizes, shared by the kernel's carve-up and the host's sizing functions
struct ExtCaps {
  uint32_t lcap, wcap, ycols, trb, opcap;
};
// cpl: cells per lane of the kernel's widest band (0: the any-width kernel)
__host__ __device__ inline ExtCaps ext_caps(uint32_t max_read_len, uint32_t max_bw, int cpl = 1) {
  ExtCaps k;
  k.lcap = (max_read_len + 31u) & ~15u;
  k.wcap = (2u * (max_read_len + max_bw) + max_read_len + 48u) & ~15u;
  k.ycols = max_read_len + max_bw + 2u;
  // LDS trace: one cell per lane (16 bytes per column).  The kernels for bands beyond 128 slots (three and four cells per
  // lane: 150 bp reads with a band of +-64) keep room for two cells per lane: most of their extensions have 65..128 slots
  // (min(2 bw + 1, |x| + 1)), and a trace in global memory -- a store per column, an agent-scope fence, a traceback of
  // dependent global loads -- is what such a launch was spending its time on.  Wider extensions still use trace_g.
  k.trb = (k.ycols + 1u) * (cpl >= 3 ? 32u : 16u);
  k.opcap = (2u * max_read_len + 2u * max_bw + 31u) & ~15u;
  return k;
}
// any-width kernel: layout of one wave's slice of global memory
struct SlowLayout {
  uint64_t rd, win, wing, pa, pb, pc, mk_k, ycl, dp, trace, total;
  uint32_t dp_stride;
};
__host__ __device__ inline SlowLayout slow_layout(uint32_t max_read_len, uint32_t max_bw, uint32_t mk_cap) {
  const ExtCaps k = ext_caps(max_read_len, max_bw);
  SlowLayout s;
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) {
    const uint64_t at = o;
    o += (bytes + 63u) & ~63ull;
    return at;
  };
  s.rd = take(k.lcap);
  s.win = take(k.wcap);
  s.wing = take(k.wcap);
  s.pa = take(k.opcap);
  s.pb = take(k.opcap);
  s.pc = take(k.opcap);
  s.mk_k = take((uint64_t)mk_cap * 4);
  s.ycl = take((uint64_t)mk_cap * 4);
  const uint32_t tiles = (2u * max_bw + 1u + 63u) / 64u;
  s.dp_stride = tiles * 64u + 64u;
  s.dp = take((uint64_t)s.dp_stride * 4u * 4u);
  s.trace = take((uint64_t)(k.ycols + 1u) * tiles * 16u);
  s.total = o;
  return s;
}

#ifndef THM_KARG_QUAL
#define THM_KARG_QUAL volatile
#endif
// C: coordinate width.  CPL: band slots per lane of the register-resident DP (1..4), or 0 for the any-width
// kernel (band in tiles, wave-private buffers in global memory): the slow path for reads whose band or length
// exceeds what LDS and registers hold.  MINW: waves per SIMD the register budget is set for.
//
// TEAM > 0: the kernel for reads with very many seed hits (plan_kernel lists them).  A workgroup of
// TEAM wavefronts works on ONE read at a time.  The hits of a read must be taken in order because band, X-drop
// and best score are carried from hit to hit (src/aligner.rs:143-175) -- but they change only when a hit beats the
// best score so far, which happens a handful of times per read, early.  So the hits are cut into the chunks of
// up to 64 the loop below walks anyway, and in every round wave w takes chunk (base + w), starting from the state
// that is known to hold at chunk `base` -- speculating that the chunks before its own leave the state alone.
// After the round the chunks up to and including the first one that changed the state are valid (each of them did
// start from the true state); their candidates and counters are kept, the state moves on, and the chunks behind
// them are redone in the next round.  Exact: every hit that is kept was extended under exactly the band, X-drop
// and threshold the sequential loop would have used.  Speed-up on a read whose state has settled: TEAM-fold.
template <class C, int CPL, int MINW, int TEAM = 0>
__global__ __launch_bounds__(TEAM > 0 ? 64 * TEAM : 256, TEAM > 0 ? 1 : MINW) void extend_kernel(ExtendParamsT<C> p_by_value) {
  typedef typename CoordTraits<C>::S S;
  constexpr bool GS = (CPL == 0);
  constexpr bool TM = TEAM > 0;
  constexpr int TW = TM ? TEAM : 1;
  // team state (LDS, unused when TEAM == 0)
  __shared__ int t_res[TW][8];        // per wave and round: accepted candidates, state changed?, band, X-drop, best score, per-read fault bits
  __shared__ int t_state[4];          // the state in force at chunk t_ctl[1]: band, X-drop, best score
  __shared__ unsigned t_ctl[4];       // [0] list slot of the read, [1] first chunk of the round, [2] accepted so far, [3] per-read fault bits
  __shared__ unsigned char t_nacc[TM ? TEAM_MAX_CHUNKS : 1];  // accepted candidates per finished chunk (<= TEAM_CHUNK)
  static_assert(!TM || sizeof(t_res) + sizeof(t_state) + sizeof(t_ctl) + sizeof(t_nacc) <= TEAM_STATIC_LDS, "launch.h: TEAM_STATIC_LDS");
  typedef WctxT<GS> Wctx;
  // The parameter block (about 80 dwords) is read from the kernel-argument segment where it is
  // needed (scalar loads, THM_KARG_QUAL = volatile keeps them at their use sites) instead of being
  // loaded at entry: the kernel is far over the scalar register budget, and every parameter that
  // lives in a register across the hit loop is spilled to VGPR lanes and read back with v_readlane.
#if __HIP_DEVICE_COMPILE__
  typedef THM_KARG_QUAL const __attribute__((address_space(4))) ExtendParamsT<C> KArgs;
  KArgs& p = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  const ExtendParamsT<C>& p = p_by_value;
#endif
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = bcast_first((int)(threadIdx.x >> 6));  // wave-uniform: LDS bases stay on the scalar unit
  const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + (unsigned)wave;
  // ---- wave-private buffers (LDS carve must match extend_lds_bytes; global layout = slow_layout) ----
  const ExtCaps caps = ext_caps(p.max_read_len, p.max_bw, CPL);
  const uint32_t lcap = caps.lcap, wcap = caps.wcap, opcap = caps.opcap;
  Wctx c;
  if constexpr (GS) {
    const SlowLayout sl = slow_layout(p.max_read_len, p.max_bw, p.mk_cap);
    uint8_t* base = p.slow_scratch + (size_t)wave_global * p.slow_scratch_per_wave;
    c.rd = base + sl.rd;
    c.win = base + sl.win;
    c.wing = base + sl.wing;
    c.trace = nullptr;
    c.trace_g = (unsigned long long*)(base + sl.trace);
    c.pa = base + sl.pa;
    c.pb = base + sl.pb;
    c.pc = base + sl.pc;
    c.mk_k = (int*)(base + sl.mk_k);
    c.ycl = (uint32_t*)(base + sl.ycl);
    c.dp = (int*)(base + sl.dp);
    c.dp_stride = (int)sl.dp_stride;
    c.mk_cap = (int)p.mk_cap;
    c.ck = nullptr;
  } else {
    const uint32_t per_wave = lcap + 2u * wcap + caps.trb + 3u * opcap + 8u * FAST_MAX_YCLIPS + (uint32_t)(KEYCAP * sizeof(CandKey));
    uint8_t* base = smem + (size_t)wave * per_wave;
    c.rd = base;
    c.win = c.rd + lcap;
    c.wing = c.win + wcap;
    c.trace = (unsigned long long*)(c.wing + wcap);
    c.trace_g = p.trace_scratch + (size_t)wave_global * ((size_t)(caps.ycols + 1u) * (CPL > 0 ? CPL : 1) * 2u);
    c.pa = (uint8_t*)c.trace + caps.trb;
    c.pb = c.pa + opcap;
    c.pc = c.pb + opcap;
    c.mk_k = (int*)(c.pc + opcap);
    c.ycl = (uint32_t*)(c.mk_k + FAST_MAX_YCLIPS);
    c.ck = (CandKey*)(c.ycl + FAST_MAX_YCLIPS);
    c.dp = nullptr;
    c.dp_stride = 0;
    c.mk_cap = FAST_MAX_YCLIPS;
  }
  c.opcap = (int)opcap;
  c.wcap = (int)wcap;
  c.cells = c.cols = c.calls = c.winbytes = 0;
  c.fault = 0;
  c.pool_off = 0;
  c.pool_left = 0;
#ifdef THM_PROF
  for (int t = 0; t < 10; t++) c.prof_acc[t] = 0;
  for (int t = 0; t < 6; t++) c.prof_cols[t] = 0;
  c.prof_hit = c.prof_tx = 0;
  c.prof_last = __builtin_amdgcn_s_memtime();
#endif

#ifdef THM_TIMELINE
  // diagnosis build (tools/timeline.py): when do the waves of the wave-per-read kernel run out of work, and how long do
  // reads take?
  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  unsigned long long tl_last = tl_start;
  unsigned tl_last_hits = 0, tl_reads = 0, tl_tx = 0;
  unsigned long long tl_longest = 0;
  auto tl_close = [&](unsigned long long now) {  // the read that started at tl_last ends now
    if (!tl_reads) return;
    const unsigned long long d = now - tl_last;  // 10 ns units
    const unsigned long long key = (d << 32) | ((unsigned long long)min(tl_tx, 65535u) << 16) | tl_last_hits;
    if (key > tl_longest) tl_longest = key;
  };
#endif
  // SMEM pool overflow in the seed stage: the per-read SMEM runs are incomplete (the host grows the pool and
  // replays the batch); nothing of them may be read
  if (uload(p.fault_seed) != 0) return;
  auto& ix = p.ix;
  unsigned long long k_aligned = 0, k_unmapped = 0, k_alns = 0, k_reads = 0, k_opb = 0;
  unsigned k_type[3] = {0, 0, 0};  // wave-uniform (counted with ballots)
  unsigned long long k_cells = 0, k_cols = 0, k_calls = 0, k_win = 0;
  int batch_fault = 0;

  // Reads are handed out by atomic counters, READ_RUN consecutive reads at a time (the second to fourth read of a run
  // start without the atomic's round trip, and their records and bases share cache lines with the first: 3.95 ->
  // 3.89 ms).  Small chunks balance the waves
  // (the cost per read varies by orders of magnitude: repeats), but one hot word serves only ~88 M
  // returning atomics per second; so there are EXT_NQ counters on separate cache lines, each over
  // its own contiguous share of the batch, and a wave that finds its counter exhausted moves on to
  // the next one.
  const bool list_only = GS || p.list_only != 0;
  const unsigned n_total = (list_only || p.skip_scan != 0) ? 0u : (unsigned)p.reads.n_reads;
  const unsigned q_share = (n_total + EXT_NQ - 1) / EXT_NQ;
  unsigned my_q = wave_global % EXT_NQ, q_tried = 0;
  // Longest jobs first: reads with many seed hits (repeats; up to a few hundred hits, i.e.
  // milliseconds, against ~50 us for a typical read) are listed by plan_kernel and handed
  // out before everything else, one per wave; left in input order the last ones would start near
  // the end of the batch and the whole grid would wait for them (a quarter of the kernel's time
  // on the benchmark workload).  The any-width kernel works from its list alone.
  bool heavy_phase = true;
  // (see ExtendParamsT::team) the team list is the team kernel's while it is short, else the wave-per-read kernel's
  const unsigned n_team_all = (GS || p.team_count == nullptr) ? 0u : (unsigned)uload(p.team_count);
  const bool team_active = n_team_all <= p.team_limit;
  const unsigned n_heavy_own = TM ? 0u : (unsigned)uload(p.heavy_count);
  const unsigned n_heavy = TM ? (team_active ? n_team_all : 0u) : n_heavy_own + ((!GS && !team_active && !list_only) ? n_team_all : 0u);
  const unsigned list_q = (GS ? EXT_NQ + 1 : EXT_NQ) * EXT_QSTRIDE;
  // (Issuing the atomic for the next read while the current one is worked on was tried: the pending return value
  // stays live across the whole hit loop and costs 400 bytes per lane of spills -- three times slower.)
  unsigned pend_idx = 0, pend_n = 0;  // rest of the run of consecutive reads one queue atomic handed out
  // run length: at most READ_RUN, and short enough that every wave of the grid gets about eight runs (small batches)
  const unsigned read_run = max(1u, min((unsigned)READ_RUN, n_total / (gridDim.x * 4u * 8u)));
  for (;;) {
    bool from_heavy = false, got = false;
    unsigned idx = 0;
    if (!TM && pend_n) {
      idx = pend_idx++;
      pend_n--;
      got = true;
    }
    if constexpr (TM) {
      // one read for the whole workgroup
      if (threadIdx.x == 0) t_ctl[0] = atomicAdd(p.queue + (EXT_NQ + 2) * EXT_QSTRIDE, 1u);
      __syncthreads();
      const unsigned g = t_ctl[0];
      __syncthreads();
      if (g >= n_heavy) break;
      idx = (unsigned)uload(&p.team[g]);
      got = true;
      from_heavy = true;
      heavy_phase = false;
    }
    if (!got && heavy_phase) {
      unsigned g = 0;
      if (lane == 0) g = atomicAdd(p.queue + list_q, 1u);
      g = (unsigned)bcast_first((int)g);
      if (g < n_heavy) {
        idx = (g < n_heavy_own) ? (unsigned)uload(&p.heavy[g]) : (unsigned)uload(&p.team[g - n_heavy_own]);
        got = true;
        from_heavy = true;
      } else {
        heavy_phase = false;
      }
    }
    while (!got && q_tried < EXT_NQ && n_total) {
      unsigned g = 0;
      if (lane == 0) g = atomicAdd(p.queue + my_q * EXT_QSTRIDE, read_run);
      g = (unsigned)bcast_first((int)g);
      const unsigned lo = my_q * q_share, hi = min(lo + q_share, n_total);
      if (lo < hi && g < hi - lo) {
        idx = lo + g;
        pend_idx = idx + 1;
        pend_n = min(read_run - 1u, hi - lo - g - 1u);
        got = true;
        break;
      }
      my_q = (my_q + 1) % EXT_NQ;
      q_tried++;
    }
    if (!got) break;
    // everything needed to start on the read, in one scalar load (launch.h, ReadRecT)
    const ReadRecT<C> rec = uload(&p.read_recs[idx]);
    // not this launch's read: the slow class (and reads beyond every class) are listed by plan_kernel
    if (rec.len > p.max_read_len) continue;
    const uint64_t r0 = rec.base_off;
    const int L = (int)rec.len;
    c.L = L;
    #pragma unroll 1
    for (int t = lane; t < (int)lcap; t += 64) c.rd[t] = (t < L) ? p.reads.bases[r0 + t] : (uint8_t)0;  // already upper-cased and sanitised
    wsync(c);

    // thresholds, reference src/aligner.rs:130-138 (binary32 product, truncation toward zero)
    const float prod = p.opts.min_aln_score_percent * (float)L;
    int ms_pct = (prod != prod) ? 0 : (prod >= 2147483648.0f ? 2147483647 : (prod <= -2147483648.0f ? (-2147483647 - 1) : (int)prod));
    const int min_aln_score = max(ms_pct, p.opts.min_aln_score);
    int max_aln_score = min_aln_score;
    int band_width = (min_aln_score < 0) ? 0 : max(L - min_aln_score, 0);
    int x_drop = band_width;
    const int range = (int)p.opts.multimap_score_range;
    const bool intron_mode = p.opts.intron_mode != 0;
    bool band_bad = false;
    if (band_width > (int)p.max_bw || (CPL > 0 && 2 * band_width + 1 > 64 * CPL)) {
      c.fault |= FAULT_BAND;
      band_bad = true;
      band_width = x_drop = 0;
    }

    const uint64_t cand0 = rec.cand_off;
    Cand* cands = p.cands + cand0;
    uint32_t* order = p.order + 2 * cand0;  // two scratch lists of the read's hit count each
    const uint64_t n_hits_cap = rec.n_hits;
    if (!GS && !from_heavy && n_hits_cap >= HEAVY_HITS) continue;  // went out with the heavy reads
#ifdef THM_TIMELINE
    {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      tl_close(now);
      tl_last = now;
      tl_last_hits = (unsigned)min(n_hits_cap, (uint64_t)65535);
      tl_tx = 0;
      tl_reads++;
    }
#endif
    uint32_t n_acc = 0;
    unsigned acc_bytes = 0;  // op bytes and type of the most recent accepted candidate
    int acc_type = 0;

    PROF_MARK(c, PS_SETUP);
    const uint64_t s0 = rec.smem_off;
    uint32_t n_sm = rec.smem_cnt;
    if (cand0 + n_hits_cap > p.cand_cap) {  // candidate pool too small: the host grows it and reruns
      c.fault |= FAULT_OPS_POOL;
      n_sm = 0;
    }
    if (band_bad) n_sm = 0;  // no hit is extended
#ifdef THM_PROF
    c.prof_hit = 0;
#endif
    // ---- team bookkeeping (TEAM > 0; see the comment at the kernel's head) ----
    unsigned t_base = 0, t_total = 0;
    if constexpr (TM) {
      for (uint32_t si = 0; si < n_sm; si++) {  // chunks of the read: every SMEM's occurrences in chunks of TEAM_CHUNK
        const uint64_t cnt = (si == 0) ? (uint64_t)(rec.hi0 - rec.lo0) : (uint64_t)(uload(&p.smems[s0 + si].hi) - uload(&p.smems[s0 + si].lo));
        t_total += (unsigned)((cnt + TEAM_CHUNK - 1) / TEAM_CHUNK);
      }
      if (threadIdx.x == 0) {
        t_state[0] = band_width;
        t_state[1] = x_drop;
        t_state[2] = max_aln_score;
        t_ctl[1] = 0;
        t_ctl[2] = 0;
        t_ctl[3] = 0;
      }
      __syncthreads();
    }
    for (;;) {  // rounds of the team; a single pass otherwise
    unsigned cc = 0;            // chunk counter of the walk
    uint64_t h_start = 0;       // hit ordinal of the chunk's first hit = its first candidate slot
    const unsigned my_chunk = t_base + (unsigned)wave;
    int st_bw = 0, st_xd = 0, st_max = 0;
    unsigned sv_cells = 0, sv_cols = 0, sv_calls = 0, sv_win = 0;
    if constexpr (TM) {
      band_width = t_state[0];
      x_drop = t_state[1];
      max_aln_score = t_state[2];
      st_bw = band_width;
      st_xd = x_drop;
      st_max = max_aln_score;
      n_acc = 0;
      sv_cells = c.cells;
      sv_cols = c.cols;
      sv_calls = c.calls;
      sv_win = c.winbytes;
    }
    for (uint32_t si = 0; si < n_sm; si++) {
      SmemT<C> sm;
      if (si == 0) {  // the first SMEM travels in the read's record
        sm.lo = rec.lo0;
        sm.hi = rec.hi0;
        sm.qpos = rec.qpos0;
        sm.len = rec.len0;
      } else {
        sm = uload(&p.smems[s0 + si]);
      }
      const int q = sm.qpos, len = sm.len;
      C rr = sm.hi;
      if constexpr (TM) {
        // a team works in chunks of TEAM_CHUNK hits (the finer the chunks, the less is redone when a hit moves the
        // state); a wave goes straight to its own chunk of the round
        const uint64_t cnt = (uint64_t)(sm.hi - sm.lo);
        const unsigned n_ch = (unsigned)((cnt + TEAM_CHUNK - 1) / TEAM_CHUNK);
        if (my_chunk < cc || my_chunk >= cc + n_ch) {
          cc += n_ch;
          h_start += cnt;
          continue;
        }
        const unsigned k = my_chunk - cc;
        rr = sm.hi - (C)k * (C)TEAM_CHUNK;
        h_start += (uint64_t)k * TEAM_CHUNK;
        cc = my_chunk;
      }
      while (rr > sm.lo) {
        const uint32_t chunk = (uint32_t)min((C)(TM ? TEAM_CHUNK : 64), (C)(rr - sm.lo));
        if (TM && cc != my_chunk) break;  // the wave's one chunk of this round is done
        Cand* const cslot = TM ? cands + h_start : cands;  // a team's chunk writes its candidates from its first hit's slot
        const uint64_t slot_cap = TM ? (uint64_t)chunk : n_hits_cap;
        C my_sa = 0;
        if (si == 0 && rr == sm.hi && chunk == 1)
          my_sa = rec.sa0;  // ... and so does its first (here: only) occurrence
        else if ((uint32_t)lane < chunk)
          my_sa = ix.sa[rr - 1 - lane];
        for (uint32_t t = 0; t < chunk; t++) {
          const S hr = (S)readlane_c(my_sa, bcast_first((int)t));
          // ================= align_seed_hit (src/aligner.rs:198-314) =================
          const int bw = band_width, xd = x_drop;
          // The genome window is requested before the contig lookup so that the two memory
          // latencies overlap; it is loaded unclamped (the text is padded) and clamped to the
          // contig logically below (:212-215).
          const S gw_a = max(hr - (S)(L + bw), (S)0);
          const S gw_b = min(hr + (S)(len + L + bw), (S)ix.n);
          const unsigned gw_mis = (unsigned)((uintptr_t)(ix.text + gw_a) & 15u);
          const int gw_n = (int)(gw_b - gw_a) + (int)gw_mis;
          const uint4* gw_src = (const uint4*)(ix.text + gw_a - gw_mis);
          uint4 gw_v = make_uint4(0, 0, 0, 0);
          if (gw_n <= c.wcap && lane * 16 < gw_n) gw_v = gw_src[lane];
          const RefInfoT<C> ref = idx_to_ref<C>(ix, (C)hr);
          const C qs = (C)hr, qe = (C)(hr + len);  // the seed on the concatenated text

          // One loop runs the genome extension (target 0) and then one extension per
          // transcript yielded by exon_to_tx.find (:231-258), so that the extension
          // code is instantiated once.
          PathT<S> gx, best;
          gx.score = gx.nops = gx.xstart = gx.xend = 0;
          gx.ystart = gx.yend = 0;
          best = gx;
          bool have_best = false;
          uint32_t best_tx = 0;
          uint8_t* cur_buf = c.pb;
          uint8_t* best_buf = c.pc;
          bool genome_done = false;
          LrMemo gmemo, tmemo;
          GridQueryT<C, ExonEntryT<C>> eg;
          uint32_t best_ent = 0;  // the exon-grid entry the best transcript was reached through
          for (;;) {
            S win0, lo_abs, hi_abs, t_r;
            int t_q, t_len;
            uint8_t* buf;
            uint32_t tx_idx = 0, cur_ent = 0;
            bool by_coords = false;
            if (!genome_done) {
              // genome window (:212-215)
              const S rs = (S)ref.start;
              const S seq_start = max((hr > L + bw) ? hr - (S)(L + bw) : (S)0, rs);
              const S seq_end = min(hr + (S)(len + L + bw), (S)ref.end - 1);
              if (gw_n > c.wcap) {
                c.fault |= FAULT_INTERNAL;
              } else {
                uint4* wdst = (uint4*)c.wing;
                if (lane * 16 < gw_n) wdst[lane] = gw_v;
#pragma unroll 1
                for (int t2 = lane + 64; t2 * 16 < gw_n; t2 += 64) wdst[t2] = gw_src[t2];
              }
              c.winbytes += (unsigned)(seq_end - seq_start);
              wsync(c);
              PROF_MARK(c, PS_STAGE);
              win0 = gw_a - (S)gw_mis;
              lo_abs = seq_start;
              hi_abs = seq_end;
              t_r = hr;
              t_q = q;
              t_len = len;
              buf = c.pa;
            } else {
              // next interval exon_to_tx.find would yield (interval grid, same order)
              uint32_t ent_idx = 0;
              const bool found = grid_next(eg, tx_idx, ent_idx);
              PROF_MARK(c, PS_TREE);
              if (!found) break;
#ifdef THM_TIMELINE
              tl_tx++;
#endif
              cur_ent = ent_idx;
              // the entry carries what is needed of its transcript and exon (thermite_internal.h, ExonEntryT)
              const ExonEntryT<C> ge = uload(&eg.ent[ent_idx]);
              S xs, xe;
              int exon_sum;
              thm_tx tx;
              tx.seq_off = ge.seq_off;
              tx.seq_len = ge.seq_len;
              if (ge.prev_end <= qs) {
                // lift_mem_to_tx (src/txome.rs:82-103): this exon is the first of its transcript that intersects the seed
                xs = (S)ge.start;
                xe = (S)ge.end;
                exon_sum = (int)ge.txoff;
              } else {
                // the previous exon reaches into the seed (a seed across a short intron), or the exons of this transcript
                // do not ascend: first exon in transcript order that intersects, by looking at them all
                tx = uload(&ix.txs[tx_idx]);
                int fe = -1;
                for (uint32_t e0 = 0; e0 < tx.n_exons && fe < 0; e0 += 64) {
                  const uint32_t e = e0 + (uint32_t)lane;
                  bool hit = false;
                  if (e < tx.n_exons) {
                    const thm_exon x = ix.exons[tx.exon_begin + e];
                    const C x0 = (C)x.start, x1 = (C)x.end;
                    hit = (qs >= x0 && qs < x1) || (x0 >= qs && x0 < qe);
                  }
                  const unsigned long long m = __ballot(hit);
                  if (m) fe = (int)e0 + __builtin_ctzll(m);
                }
                if (fe < 0) {  // unreachable!() in the reference
                  c.fault |= FAULT_CONTRACT;
                  continue;
                }
                const thm_exon x = uload(&ix.exons[tx.exon_begin + fe]);
                exon_sum = (int)uload(&ix.exon_txoff[tx.exon_begin + fe]);
                xs = (S)x.start;
                xe = (S)x.end;
              }
              int tr_ = (int)((hr > xs) ? hr - xs : (S)0) + exon_sum;
              const int start_offset = (int)((xs > hr) ? xs - hr : (S)0);
              const int t_end = (int)(min(hr + (S)len, xe) - xs) + exon_sum;
              t_q = q + start_offset;
              t_len = t_end - tr_;
              const int tlen = (int)tx.seq_len;
              // The seed lies inside this exon, and so does every base the genome extensions looked at (R.jmax /
              // Lt.jmax columns beyond the seed, results that do not depend on anything further: `broke`): the
              // transcript reads the same bases there (its sequence is assembled from the exons), extend_seed_match
              // finds the same mismatch next to the seed, and the two SwgExtend::extend calls return what they
              // returned for the genome window.  Known from the coordinates alone: no transcript window is staged.
              by_coords = ge.prev_end <= qs && t_q == q && t_len == len && gmemo.R.broke && gmemo.Lt.broke &&
                          hr + (S)(len + gmemo.R.jmax) <= xe && hr - (S)gmemo.Lt.jmax >= xs;
              int w0 = 0;
              // window of the transcript around the lifted seed
              const int ws = (tr_ > L + bw) ? tr_ - (L + bw) : 0;
              const int we = min(tlen, tr_ + t_len + L + bw + 1);
              if (by_coords) c.winbytes += (unsigned)(we - ws);  // THM_CNT_WINDOW_BYTES counts the target's window, staged or not
              if (!by_coords) {
              w0 = stage_window(c, c.win, ix.tx_seq + tx.seq_off, ws, we);
              // extend_seed_match (src/aligner.rs:410-426): ballots of the first mismatch
              {
                int ext = 0;
                for (bool done = false; !done;) {
                  const int tt = ext + lane;
                  const int rp = tr_ + t_len + tt;
                  const int qp = t_q + t_len + tt;
                  const bool ok = (rp < tlen) && (qp < L) && (c.win[rp - w0] == c.rd[qp]);
                  const unsigned long long bad = __ballot(!ok);
                  if (bad) {
                    ext += __builtin_ctzll(bad);
                    done = true;
                  } else {
                    ext += 64;
                  }
                }
                t_len += ext;
                ext = 0;
                for (bool done = false; !done;) {
                  const int tt = ext + lane + 1;
                  const int rp = tr_ - tt;
                  const int qp = t_q - tt;
                  const bool ok = (rp >= 0) && (qp >= 0) && (c.win[rp - w0] == c.rd[qp]);
                  const unsigned long long bad = __ballot(!ok);
                  if (bad) {
                    ext += __builtin_ctzll(bad);
                    done = true;
                  } else {
                    ext += 64;
                  }
                }
                tr_ -= ext;
                t_q -= ext;
                t_len += ext;
              }
              }
              win0 = (S)w0;
              t_r = (S)tr_;
              lo_abs = 0;
              hi_abs = (S)tlen;
              buf = cur_buf;
              PROF_MARK(c, PS_TXPREP);
            }
            PathT<S> pth;
            bool reused = false;
            if (genome_done && by_coords) {
              reused = true;
              pth.score = gx.score;
              pth.nops = gx.nops;
              pth.xstart = gx.xstart;
              pth.xend = gx.xend;
              pth.ystart = t_r - gmemo.Lt.yend;
              pth.yend = t_r + t_len + gmemo.R.yend;
#pragma unroll 1
              for (int t2 = lane; t2 < gx.nops; t2 += 64) buf[t2] = c.pa[t2];
              wsync(c);
              c.calls += 2;  // two extend() calls in the reference's terms
              PROF_MARK(c, PS_TXPREP);
            } else if (genome_done && t_q == q && t_len == len) {
              // Same seed on the read: if the transcript window agrees with the genome
              // window on every column the genome extensions looked at (the hit sits
              // inside one exon and the alignment does not reach its ends), the two
              // SwgExtend::extend calls would return what they returned for the genome.
              const int xr = L - (t_q + t_len);
              const int yr = (int)min(hi_abs - (t_r + t_len), (S)(xr + bw + 1));
              const S rel = t_r - lo_abs;
              const S y0 = lo_abs + (rel > L + bw ? rel - (L + bw) : (S)0);
              const int yl = (int)min(t_r - y0, (S)(t_q + bw + 1));
              const SwgResult& R = gmemo.R;
              const SwgResult& Lt = gmemo.Lt;
              bool ok = (R.broke ? yr >= R.jmax : yr == gmemo.yr) && (Lt.broke ? yl >= Lt.jmax : yl == gmemo.yl);
              if (ok) {
                const uint8_t* tr2 = c.win + (int)(t_r + t_len - win0);
                const uint8_t* gr_ = c.wing + gmemo.yoff_r;
                const uint8_t* tl_ = c.win + (int)(t_r - 1 - win0);
                const uint8_t* gl_ = c.wing + gmemo.yoff_l;
                bool differ = false;
#pragma unroll 1
                for (int j0 = 0; j0 < R.jmax; j0 += 64) {
                  const int j = j0 + lane;
                  differ = differ || (j < R.jmax && tr2[j] != gr_[j]);
                }
#pragma unroll 1
                for (int j0 = 0; j0 < Lt.jmax; j0 += 64) {
                  const int j = j0 + lane;
                  differ = differ || (j < Lt.jmax && tl_[-j] != gl_[-j]);
                }
                ok = __ballot(differ) == 0ull;
              }
              if (ok) {
                reused = true;
                pth.score = gx.score;
                pth.nops = gx.nops;
                pth.xstart = gx.xstart;
                pth.xend = gx.xend;
                pth.ystart = t_r - Lt.yend;
                pth.yend = t_r + t_len + R.yend;
#pragma unroll 1
                for (int t2 = lane; t2 < gx.nops; t2 += 64) buf[t2] = c.pa[t2];
                wsync(c);
                c.calls += 2;  // two extend() calls in the reference's terms
                PROF_MARK(c, PS_TXPREP);
              }
            }
#ifdef THM_PROF
            c.prof_tx = genome_done ? 1 : 0;
#endif
            if (!reused)
              pth = extend_lr<CPL, S>(c, genome_done ? c.win : c.wing, win0, lo_abs, hi_abs, t_r, t_q, t_len, bw, xd, buf,
                                      genome_done ? tmemo : gmemo);
            if (!genome_done) {
              gx = pth;
              genome_done = true;
              // (asking before the genome extension, so that the two round trips run under it, was tried: the query's
              // registers stay live across the DP loops -- 7 % slower)
              grid_begin<C>(eg, ix.exon_grid_off, ix.exon_grid, qs, qe);
            } else {
              if (!have_best || pth.score > best.score) {  // strictly better (:249)
                have_best = true;
                best_tx = tx_idx;
                best_ent = cur_ent;
                best = pth;
                uint8_t* tmp = cur_buf;
                cur_buf = best_buf;
                best_buf = tmp;
              }
              if (pth.score >= L * MATCH_SCORE) break;  // cannot beat an exact match (:253-257)
            }
          }

          // ---- exonic vs unspliced (:263-313) ----
          int aln_type;
          uint32_t type_idx = THM_NO_IDX;
          S cy0, cy1;  // concatenated coordinates of the genome alignment
          const uint8_t* g_path;
          int g_n, g_ny = 0;
          int sc, xs_, xe_;
          const bool exonic = have_best && best.score >= gx.score;
          if (exonic) {
            g_path = best_buf;
            g_n = best.nops;
            aln_type = THM_ALN_EXONIC;
            type_idx = best_tx;
            sc = best.score;
            xs_ = best.xstart;
            xe_ = best.xend;
            cy0 = cy1 = 0;  // set by the lift below, only if the alignment is kept
          } else {
            cy0 = gx.ystart;
            cy1 = gx.yend;
            g_path = c.pa;
            g_n = gx.nops;
            sc = gx.score;
            xs_ = gx.xstart;
            xe_ = gx.xend;
            aln_type = THM_ALN_INTERGENIC;
          }
          // ================= back in align_read's loop (:146-174) =================
          bool accept = intron_mode || exonic;
          if (sc < p.opts.min_aln_score || sc < min_aln_score || sc < max_aln_score - range) accept = false;
          if (accept) {
            uint32_t best_tlen = 0;
            if (exonic) {
              // lift_tx_to_gx (src/txome.rs:110-160).  An alignment that stays inside the exon its seed was lifted through
              // has no intron to insert: its ends move by the exon's offset.  (Ending exactly on the exon's last base
              // still receives the intron when a clip follows, :132-141 -- that case takes the general path.)
              const ExonEntryT<C> ge = uload(&eg.ent[best_ent]);
              best_tlen = ge.seq_len;
              const int ys_ = (int)best.ystart, ye_ = (int)best.yend;
              const int e_lo = (int)ge.txoff, e_hi = e_lo + (int)(ge.end - ge.start);
              const bool inside = ge.prev_end <= qs && ys_ >= e_lo && ys_ < e_hi &&
                                  (ye_ < e_hi || (ye_ == e_hi && (ge.exon_idx + 1 >= ge.n_exons || !(best.xend < L))));
              if (inside) {
                cy0 = (S)ge.start + (S)(ys_ - e_lo);
                cy1 = (S)ge.start + (S)(ye_ - e_lo);
                g_ny = 0;
                PROF_MARK(c, PS_LIFT);
              } else {
                g_ny = lift_markers<S>(c, ix, uload(&ix.txs[best_tx]), best_buf, best.nops, best.xend < L, ys_, ye_, cy0, cy1);
              }
            } else {
              // first interval gene_intervals.find yields (:283-288, :306); only reached in intron mode
              GridQueryT<C, GridEntryT<C>> gg;
              grid_begin<C>(gg, ix.gene_grid_off, ix.gene_grid, (C)cy0, (C)cy1);
              uint32_t gene = 0, gene_ent = 0;
              if (grid_next(gg, gene, gene_ent)) {
                aln_type = THM_ALN_INTRONIC;
                type_idx = gene;
              }
              PROF_MARK(c, PS_TREE);
            }
            // concat_to_chr_aln (:429-449).  Index::idx_to_ref(ystart): the alignment starts inside the contig copy of
            // its hit (the genome window is clamped to it, a transcript's exons lie on one copy), so the lookup is the
            // hit's own; anything else goes through the search.
            const RefInfoT<C> cref = ((C)cy0 >= ref.start && (C)cy0 < ref.end) ? ref : idx_to_ref<C>(ix, (C)cy0);
            uint64_t ch0, ch1;
            bool rev;
            if (cref.strand) {
              ch0 = (uint64_t)((C)cy0 - cref.start);
              ch1 = (uint64_t)((C)cy1 - cref.start);
              rev = false;
            } else {
              ch0 = (uint64_t)(C)(cref.len - ((C)cy1 - cref.start));
              ch1 = (uint64_t)(C)(cref.len - ((C)cy0 - cref.start));
              rev = true;
            }
            int nb = 0, tnb = 0;
            const unsigned long long off = emit_alignment(c, p, g_path, g_n, xs_, xe_, rev, g_ny, nb);
            unsigned long long toff2 = 0;
            if (exonic) toff2 = emit_alignment(c, p, best_buf, best.nops, best.xstart, best.xend, false, 0, tnb);
            if (n_acc >= slot_cap) {
              c.fault |= FAULT_INTERNAL;
            } else if (lane == 0) {
              Cand cd;
              cd.ystart = ch0;
              cd.yend = ch1;
              cd.ylen = cref.len;
              cd.ops_off = off;
              cd.ops_len = (uint32_t)nb;
              cd.score = sc;
              cd.ref_id = ref.id;
              cd.xstart = (uint32_t)xs_;
              cd.xend = (uint32_t)xe_;
              cd.tx_or_gene_idx = type_idx;
              cd.name_rank = ref.name_rank;
              cd.strand = ref.strand ? 1 : 0;
              cd.aln_type = (uint8_t)aln_type;
              cd.primary = 0;
              cd.pad_ = 0;
              cd.tx_ystart = cd.tx_yend = cd.tx_ylen = 0;
              cd.tx_ops_off = 0;
              cd.tx_ops_len = 0;
              cd.tx_score = 0;
              cd.tx_xstart = cd.tx_xend = 0;
              if (exonic) {
                cd.tx_ystart = (uint64_t)best.ystart;
                cd.tx_yend = (uint64_t)best.yend;
                cd.tx_ylen = best_tlen;
                cd.tx_ops_off = toff2;
                cd.tx_ops_len = (uint32_t)tnb;
                cd.tx_score = best.score;
                cd.tx_xstart = (uint32_t)best.xstart;
                cd.tx_xend = (uint32_t)best.xend;
              }
              cslot[n_acc] = cd;
              if (!GS && !TM && n_acc < (uint32_t)KEYCAP) {
                CandKey k;
                k.ystart = ch0;
                k.yend = ch1;
                k.score = sc;
                k.name_rank = ref.name_rank;
                k.bytes = (uint32_t)(nb + tnb);
                k.strand_type = (ref.strand ? 1u : 0u) | ((uint32_t)aln_type << 8);
                c.ck[n_acc] = k;
              }
            }
            n_acc++;
            acc_bytes = (unsigned)(nb + tnb);
            acc_type = aln_type;
            PROF_MARK(c, PS_EMIT);
            // narrow the band (:162-172)
            const int lim = max(L + range - sc, 0);
            band_width = min(band_width, lim);
            x_drop = min(x_drop, lim);
            max_aln_score = max(max_aln_score, sc);
          }
#ifdef THM_PROF
          c.prof_hit++;
#endif
        }
        rr -= chunk;
        h_start += chunk;
        cc++;
      }
    }
    if constexpr (!TM) {
      break;
    } else {
      // ---- end of a round: which chunks are valid? ----
      if (t_total > (unsigned)TEAM_MAX_CHUNKS) c.fault |= FAULT_INTERNAL;  // plan_kernel keeps such reads away from the team
      const bool active = my_chunk < t_total;
      const bool changed = active && (band_width != st_bw || x_drop != st_xd || max_aln_score != st_max);
      if (lane == 0) {
        t_res[wave][0] = active ? (int)n_acc : 0;
        t_res[wave][1] = changed ? 1 : 0;
        t_res[wave][2] = band_width;
        t_res[wave][3] = x_drop;
        t_res[wave][4] = max_aln_score;
        t_res[wave][5] = active ? (c.fault & (FAULT_RETRY | FAULT_CONTRACT)) : 0;  // (FAULT_BAND is set before the rounds, by every wave alike)
      }
      __syncthreads();
      int valid = (int)min((unsigned)TEAM, t_total > t_base ? t_total - t_base : 0u);
      int wstar = -1;
      for (int w2 = 0; w2 < valid; w2++)
        if (t_res[w2][1]) {
          wstar = w2;
          break;
        }
      if (wstar >= 0) valid = wstar + 1;
      c.fault &= ~(FAULT_RETRY | FAULT_CONTRACT);  // per-read outcomes travel through t_ctl[3] (valid chunks only)
      if (wave >= valid) {  // speculation failed (or no chunk): this round's work of the wave does not count
        c.cells = sv_cells;
        c.cols = sv_cols;
        c.calls = sv_calls;
        c.winbytes = sv_win;
      }
      if (threadIdx.x == 0) {
        unsigned acc = 0, fl = 0;
        for (int w2 = 0; w2 < valid; w2++) {
          if (t_base + (unsigned)w2 < (unsigned)TEAM_MAX_CHUNKS) t_nacc[t_base + w2] = (unsigned char)t_res[w2][0];
          acc += (unsigned)t_res[w2][0];
          fl |= (unsigned)t_res[w2][5];
        }
        if (wstar >= 0) {
          t_state[0] = t_res[wstar][2];
          t_state[1] = t_res[wstar][3];
          t_state[2] = t_res[wstar][4];
        }
        t_ctl[1] = t_base + (unsigned)valid;
        t_ctl[2] += acc;
        t_ctl[3] |= fl;
      }
      __syncthreads();
      t_base = t_ctl[1];
      if (t_base >= t_total) break;
    }
    }  // rounds
    // team: the whole workgroup goes through the final section (the two rank sorts are shared out over the waves);
    // everything that has a side effect or keeps the read's books is wave 0's
    const bool lead = !TM || wave == 0;
    if constexpr (TM) {
      n_acc = t_ctl[2];
      max_aln_score = t_state[2];
      c.fault |= (int)t_ctl[3];
    }
    PROF_MARK(c, PS_OTHER);
#ifdef THM_PROF_FINAL
    const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
#endif
    // per-read outcomes that are not alignments
    if (c.fault & FAULT_RETRY) {
      // more introns in one alignment than this kernel's marker list holds: the any-width kernel redoes the read
      c.fault &= ~(FAULT_RETRY | FAULT_CONTRACT);
      if (lead && lane == 0) {
        const unsigned long long slot = atomicAdd(p.retry_count, 1ull);
        p.retry[slot] = idx;
      }
      c.cells = c.cols = c.calls = c.winbytes = 0;
      continue;
    }
    if (c.fault & FAULT_BAND) {
      c.fault &= ~FAULT_BAND;
      n_acc = 0;
      if (lead && lane == 0) {
        p.read_status[idx] = THM_ERR_INTERNAL;
        atomicAdd(p.n_contract, 1ull);  // (tells the host to fetch the statuses)
      }
    }
    if (c.fault & FAULT_CONTRACT) {
      // a condition that panics in the reference (lift_mem_to_tx / lift_tx_to_gx): no alignments, per-read status
      c.fault &= ~FAULT_CONTRACT;
      n_acc = 0;
      if (lead && lane == 0) {
        p.read_status[idx] = THM_ERR_OUT_OF_CONTRACT;
        atomicAdd(p.n_contract, 1ull);
      }
    }
    batch_fault |= c.fault;
#ifdef THM_PROF_FINAL
    const unsigned long long tf1 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t nres = 0;
    unsigned long long opb = 0;
    if (!TM && n_acc == 1) {
      // the common case: one candidate, which by construction passes retain() (its score is the maximum)
      nres = 1;
      if (lane == 0) order[0] = 0;
      opb = acc_bytes;
      k_type[0] += (acc_type == THM_ALN_EXONIC);
      k_type[1] += (acc_type == THM_ALN_INTRONIC);
      k_type[2] += (acc_type == THM_ALN_INTERGENIC);
    } else if (!GS && !TM && n_acc > 1 && n_acc <= (uint32_t)KEYCAP) {
      // ============ retain / filter_overlapping / sort / primary (:177-187), a handful of candidates: lane t holds
      // candidate t's keys; ranks by comparing against every other candidate through readlane, no memory traffic ======
      wsync(c);
      CandKey k;
      k.ystart = k.yend = 0;
      k.score = 0;
      k.name_rank = k.bytes = k.strand_type = 0;
      if ((uint32_t)lane < n_acc) k = c.ck[lane];
      const uint32_t my_strand = k.strand_type & 0xffu;
      // retain(score >= max - range): the survivors keep their order (= lane order)
      const bool keep = (uint32_t)lane < n_acc && k.score >= max_aln_score - range;
      const unsigned long long mk = __ballot(keep);
      const uint32_t m = (uint32_t)__popcll(mk);
      auto rl64 = [](uint64_t v, int l) {
        return ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffu), l);
      };
      int res_ci = 0;  // lane s: index of the s-th surviving candidate of the sweep
      if (m == 1) {
        nres = 1;
        res_ci = (int)__builtin_ctzll(mk);
      } else if (m > 1) {
        // stable sort by (ref_name, strand, ystart) (:322-327): rank among the survivors
        uint32_t r1 = 0;
        for (unsigned long long w = mk; w; w &= w - 1ull) {
          const int u = (int)__builtin_ctzll(w);
          const uint32_t ur = (uint32_t)__builtin_amdgcn_readlane((int)k.name_rank, u);
          const uint32_t us = (uint32_t)__builtin_amdgcn_readlane((int)my_strand, u);
          const uint64_t uy = rl64(k.ystart, u);
          bool less;
          if (ur != k.name_rank)
            less = ur < k.name_rank;
          else if (us != my_strand)
            less = us < my_strand;
          else if (uy != k.ystart)
            less = uy < k.ystart;
          else
            less = u < lane;
          r1 += less ? 1u : 0u;
        }
        // sweep (:329-346)
        uint64_t max_end = 0, l_yend = 0;
        uint32_t l_rank = 0, l_strand = 0;
        int l_score = 0;
        for (uint32_t sidx = 0; sidx < m; sidx++) {
          const int ci = (int)__builtin_ctzll(__ballot(keep && r1 == sidx));
          const uint64_t a_ystart = rl64(k.ystart, ci), a_yend = rl64(k.yend, ci);
          const uint32_t a_rank = (uint32_t)__builtin_amdgcn_readlane((int)k.name_rank, ci);
          const uint32_t a_strand = (uint32_t)__builtin_amdgcn_readlane((int)my_strand, ci);
          const int a_score = __builtin_amdgcn_readlane(k.score, ci);
          if (nres == 0 || a_ystart >= max_end || a_rank != l_rank || a_strand != l_strand) {
            max_end = a_yend;
            if ((uint32_t)lane == nres) res_ci = ci;
            nres++;
            l_rank = a_rank;
            l_strand = a_strand;
            l_score = a_score;
            l_yend = a_yend;
          } else {
            if (a_score > l_score) {
              if ((uint32_t)lane == nres - 1) res_ci = ci;
              l_score = a_score;
              l_yend = a_yend;
            }
            max_end = max(max_end, l_yend);
          }
        }
      }
      // stable sort by -score (:183); order[] is what compact_kernel follows
      const bool mine = (uint32_t)lane < nres;
      const int my_sc = __shfl(k.score, res_ci);
      uint32_t r2 = 0;
      for (uint32_t u = 0; u < nres; u++) {
        const int su = __builtin_amdgcn_readlane(my_sc, (int)u);
        r2 += (su > my_sc || (su == my_sc && (int)u < lane)) ? 1u : 0u;
      }
      if (mine) order[r2] = (uint32_t)res_ci;
      const uint32_t my_bytes = (uint32_t)__shfl((int)k.bytes, res_ci);
      // (the shuffles run with every lane active: ds_bpermute returns 0 for a source lane that is masked off)
      const uint32_t my_st = (uint32_t)__shfl((int)k.strand_type, res_ci);
      const int ty = mine ? (int)(my_st >> 8) : -1;
      unsigned long long ob = mine ? (unsigned long long)my_bytes : 0ull;
      for (int o = 32; o > 0; o >>= 1) ob += __shfl_xor(ob, o);
      opb = ob;
      k_type[0] += (unsigned)__popcll(__ballot(ty == THM_ALN_EXONIC));
      k_type[1] += (unsigned)__popcll(__ballot(ty == THM_ALN_INTRONIC));
      k_type[2] += (unsigned)__popcll(__ballot(ty == THM_ALN_INTERGENIC));
    } else if (TM ? n_acc >= 1 : n_acc > 1) {
      // between the steps: every store visible to every wave that reads it (a workgroup barrier for a team)
      auto step = [&] {
        if constexpr (TM)
          __syncthreads();
        else
          __threadfence_block();
      };
      const uint32_t t0_first = TM ? (uint32_t)wave * 64u : 0u, t0_step = TM ? 64u * (uint32_t)TW : 64u;
      step();
    // ============ retain / filter_overlapping / sort / primary (:177-187) ============
    uint32_t* la = order;               // list A
    uint32_t* lb = order + n_hits_cap;  // list B
    uint32_t m = 0;
    // retain(score >= max - range), keeps order
    if constexpr (TM) {
      // the candidates of chunk cc sit in the slots from its first hit on, t_nacc[cc] of them
      if (lead) {
      unsigned cc2 = 0;
      uint64_t hs = 0;
      for (uint32_t si = 0; si < n_sm; si++) {
        C lo2, hi2;
        if (si == 0) {
          lo2 = rec.lo0;
          hi2 = rec.hi0;
        } else {
          const SmemT<C> sm2 = uload(&p.smems[s0 + si]);
          lo2 = sm2.lo;
          hi2 = sm2.hi;
        }
        C left = hi2 - lo2;
        while (left > 0) {
          const uint32_t chunk = (uint32_t)min((C)TEAM_CHUNK, left);
          const uint32_t na = cc2 < (unsigned)TEAM_MAX_CHUNKS ? (uint32_t)t_nacc[cc2] : 0u;
          const uint32_t slot = (uint32_t)hs + (uint32_t)lane;
          const bool keep = (uint32_t)lane < na && (cands[slot].score >= max_aln_score - range);
          const unsigned long long mk = __ballot(keep);
          if (keep) la[m + __popcll(mk & ((1ull << lane) - 1ull))] = slot;
          m += (uint32_t)__popcll(mk);
          left -= chunk;
          hs += chunk;
          cc2++;
        }
      }
      if (lane == 0) t_ctl[1] = m;
      }
      __syncthreads();
      m = t_ctl[1];
    } else {
    #pragma unroll 1
    for (uint32_t t0 = 0; t0 < n_acc; t0 += 64) {
      const uint32_t t = t0 + lane;
      const bool keep = (t < n_acc) && (cands[t].score >= max_aln_score - range);
      const unsigned long long mk = __ballot(keep);
      if (keep) la[m + __popcll(mk & ((1ull << lane) - 1ull))] = t;
      m += (uint32_t)__popcll(mk);
    }
    }
    step();
    if (m == 1) {
      nres = 1;
    } else if (m > 1) {
      // stable sort by (ref_name, strand, ystart) (:322-327): rank sort la -> lb
      #pragma unroll 1
      for (uint32_t t0 = t0_first; t0 < m; t0 += t0_step) {
        const uint32_t t = t0 + lane;
        if (t < m) {
          const Cand a = cands[la[t]];
          uint32_t rank = 0;
          for (uint32_t u = 0; u < m; u++) {
            const Cand b = cands[la[u]];
            bool less;
            if (b.name_rank != a.name_rank)
              less = b.name_rank < a.name_rank;
            else if (b.strand != a.strand)
              less = b.strand < a.strand;
            else if (b.ystart != a.ystart)
              less = b.ystart < a.ystart;
            else
              less = u < t;
            rank += less ? 1u : 0u;
          }
          lb[rank] = la[t];
        }
      }
      step();
      // sweep (:329-346): result into la
      if (lead) {
      uint64_t max_end = 0;
      uint32_t l_rank = 0, l_strand = 0;
      int l_score = 0;
      uint64_t l_yend = 0;
      for (uint32_t s = 0; s < m; s++) {
        const uint32_t ci = (uint32_t)bcast_first((int)lb[s]);
        const Cand a = cands[ci];
        if (nres == 0 || a.ystart >= max_end || a.name_rank != l_rank || a.strand != l_strand) {
          max_end = a.yend;
          if (lane == 0) la[nres] = ci;
          nres++;
          l_rank = a.name_rank;
          l_strand = a.strand;
          l_score = a.score;
          l_yend = a.yend;
        } else {
          if (a.score > l_score) {
            if (lane == 0) la[nres - 1] = ci;
            l_score = a.score;
            l_yend = a.yend;
          }
          max_end = max(max_end, l_yend);
        }
      }
      if (TM && lane == 0) t_ctl[1] = nres;
      }
      step();
      if constexpr (TM) nres = t_ctl[1];
      // stable sort by -score (:183): rank sort la -> lb, then copy back
      #pragma unroll 1
      for (uint32_t t0 = t0_first; t0 < nres; t0 += t0_step) {
        const uint32_t t = t0 + lane;
        if (t < nres) {
          const int sa_ = cands[la[t]].score;
          uint32_t rank = 0;
          for (uint32_t u = 0; u < nres; u++) {
            const int sb = cands[la[u]].score;
            rank += (sb > sa_ || (sb == sa_ && u < t)) ? 1u : 0u;
          }
          lb[rank] = la[t];
        }
      }
      step();
      #pragma unroll 1
      for (uint32_t t = t0_first + lane; t < nres; t += t0_step) la[t] = lb[t];
      step();
    }
      // per-read totals (a team: wave 0's)
      #pragma unroll 1
      for (uint32_t t0 = 0; lead && t0 < nres; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        int ty = -1;
        if (t < nres) {
          const Cand a = cands[la[t]];
          opb += a.ops_len + a.tx_ops_len;
          ty = a.aln_type;
        }
        k_type[0] += (unsigned)__popcll(__ballot(ty == THM_ALN_EXONIC));
        k_type[1] += (unsigned)__popcll(__ballot(ty == THM_ALN_INTRONIC));
        k_type[2] += (unsigned)__popcll(__ballot(ty == THM_ALN_INTERGENIC));
      }
      for (int o = 32; o > 0; o >>= 1) opb += __shfl_xor(opb, o);
    }
#ifdef THM_PROF_FINAL
    const unsigned long long tf2 = __builtin_amdgcn_s_memtime();
#endif
    if (lead && lane == 0) {
      p.read_n_alns[idx] = nres;
      p.read_op_bytes[idx] = opb;
    }
#ifdef THM_PROF_FINAL
    {
      const unsigned long long tf3 = __builtin_amdgcn_s_memtime();
      c.prof_cols[3] += tf1 - tf0;
      c.prof_cols[4] += tf2 - tf1;
      c.prof_cols[5] += tf3 - tf2;
      c.prof_cols[1] += (n_acc > 1) ? 1 : 0;
      c.prof_cols[2] += 1;
    }
#endif
    PROF_MARK(c, PS_FINAL);
    if (lead) {
      k_reads++;
      if (nres)
        k_aligned++;
      else
        k_unmapped++;
      k_alns += nres;
      k_opb += opb;
    }
    k_cells += c.cells;
    k_cols += c.cols;
    k_calls += c.calls;
    k_win += c.winbytes;
    c.cells = c.cols = c.calls = c.winbytes = 0;
    wsync(c);
  }
#ifdef THM_TIMELINE
  if (!TM && !GS && lane == 0 && p.prof && tl_reads) {
    const unsigned long long tl_end = __builtin_amdgcn_s_memrealtime();
    atomicMax(&p.prof[0], ~tl_start);  // earliest start
    atomicMax(&p.prof[1], ~tl_end);    // first wave to leave (the work counters ran dry)
    atomicMax(&p.prof[2], tl_end);     // last wave to leave
    atomicMax(&p.prof[3], (tl_end << 20) | min((tl_end - tl_last) / 100ull, 1048575ull));  // ... and the duration (us) of its last read
    tl_close(tl_end);
    atomicMax(&p.prof[4], tl_longest);  // the longest read: duration << 32 | transcript targets << 16 | hits
    atomicAdd(&p.prof[5], 1ull);
    atomicAdd(&p.prof[6], tl_end - tl_start);
    atomicAdd(&p.prof[7], (unsigned long long)tl_reads);
    if (p.list_only == 0) {
      // waves leaving per 100 us, on the launch's clock (every wave has started by now: prof[0] is final)
      const unsigned long long t0 = ~atomicMax(&p.prof[0], 0ull);
      const unsigned long long rel = (tl_end - t0) / 10000ull;  // 100 us units
      const int b = rel < 30 ? 0 : (int)min(rel - 29ull, 7ull);  // < 3.0 ms, 3.0-3.1, ..., 3.5-3.6, >= 3.6 ms
      atomicAdd(&p.prof[8 + b], 1ull);
    }
  }
#endif
#ifdef THM_PROF
  if (lane == 0 && p.prof)
    {
      for (int t = 0; t < 10; t++) atomicAdd(&p.prof[t], c.prof_acc[t]);
      for (int t = 0; t < 6; t++) atomicAdd(&p.prof[10 + t], c.prof_cols[t]);
    }
#endif
  batch_fault |= c.fault & (FAULT_OPS_POOL | FAULT_INTERNAL);
  if (lane == 0) {
    if (batch_fault) atomicOr(p.fault, batch_fault & (FAULT_OPS_POOL | FAULT_INTERNAL));
    if (k_reads | k_calls | k_cells | k_win) {
      unsigned long long* row = p.wave_counters + (size_t)wave_global * THM_N_COUNTERS;
      row[THM_CNT_READS] = k_reads;
      row[THM_CNT_ALIGNED] = k_aligned;
      row[THM_CNT_UNMAPPED] = k_unmapped;
      row[THM_CNT_ALNS] = k_alns;
      row[THM_CNT_EXONIC] = (unsigned long long)k_type[0];
      row[THM_CNT_INTRONIC] = (unsigned long long)k_type[1];
      row[THM_CNT_INTERGENIC] = (unsigned long long)k_type[2];
      row[THM_CNT_SWG_CALLS] = k_calls;
      row[THM_CNT_DP_CELLS] = k_cells;
      row[THM_CNT_DP_COLS] = k_cols;
      row[THM_CNT_OP_BYTES] = k_opb;
      row[THM_CNT_WINDOW_BYTES] = k_win;
    }
  }
}

// counters[k] += sum of wave_counters[row][k] (the rows the extend kernels' waves left; zeroed before the launches)
__global__ __launch_bounds__(1024) void counters_reduce_kernel(const unsigned long long* rows, uint32_t n_rows, unsigned long long* counters) {
  __shared__ unsigned long long part[64][THM_N_COUNTERS];
  static_assert(THM_N_COUNTERS == 16, "one thread per counter and row group");
  const int k = (int)(threadIdx.x & 15u), g = (int)(threadIdx.x >> 4);
  unsigned long long sum = 0;
  for (uint32_t r = (uint32_t)g; r < n_rows; r += 64) sum += rows[(size_t)r * THM_N_COUNTERS + k];
  part[g][k] = sum;
  __syncthreads();
  if (g == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 64; i++) t += part[i][k];
    if (t) counters[k] += t;
  }
}

// ---- final layout: alignments of read r at alns[read_aln_off[r]..], op streams back to back in the same order (gx
// ops, then tx ops of an exonic alignment) ----
// A read's alignments follow one another in its op stream, so a read is a serial chain: a few dependent loads and
// ~200 bytes of copying per alignment, about 4 us each.  With one 16-lane group per read the launch lasted as long
// as its most multi-mapped read (58 alignments on the chr21-sized text: 0.25 ms of a 0.29 ms launch; 1 300 on the
// 2.2 G-symbol text, whose repeat families are larger: 1.9 ms).  Reads beyond COMPACT_HEAVY_N alignments therefore go
// through two more kernels: a scan of their op lengths (one workgroup per read), then the copies, four alignments per
// group, all groups of the grid at once.
constexpr uint32_t COMPACT_HEAVY_N = 8, COMPACT_CHUNK = 4;

// A candidate record as the 16 lanes of a group hold it: lane j its dwords j and 16 + j (Cand and thm_aln are 28 dwords
// each; two coalesced loads, two coalesced stores, and the record never occupies 28 registers of every lane).
struct CandRegs {
  uint32_t lo, hi;
};
static_assert(sizeof(Cand) == 112 && sizeof(thm_aln) == 112, "compact_emit permutes the dwords of these layouts");
static_assert(offsetof(Cand, ops_off) == 48 && offsetof(Cand, tx_ops_off) == 56 && offsetof(Cand, score) == 64 &&
                  offsetof(Cand, ops_len) == 80 && offsetof(Cand, tx_ops_len) == 84 && offsetof(Cand, strand) == 108 &&
                  offsetof(thm_aln, ops_off) == 24 && offsetof(thm_aln, tx_ops_off) == 56 && offsetof(thm_aln, xlen) == 80 &&
                  offsetof(thm_aln, ops_len) == 84 && offsetof(thm_aln, tx_ops_len) == 104 && offsetof(thm_aln, strand) == 108,
              "compact_emit permutes the dwords of these layouts");

__device__ __forceinline__ CandRegs cand_load(const Cand* c, int sub) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(c);
  CandRegs r;
  r.lo = w[sub];
  r.hi = sub < 12 ? w[16 + sub] : 0u;
  return r;
}

// alignment `ai` of the batch: op streams to ops[o..], record to alns[ai]; returns the op bytes written
__device__ __forceinline__ uint32_t compact_emit(const CompactParams& p, const CandRegs c, uint64_t ai, uint64_t o, uint32_t xlen, bool primary, int sub) {
  const uint64_t ops_off = (uint64_t)__shfl(c.lo, 12, 16) | ((uint64_t)__shfl(c.lo, 13, 16) << 32);
  const uint64_t tx_ops_off = (uint64_t)__shfl(c.lo, 14, 16) | ((uint64_t)__shfl(c.lo, 15, 16) << 32);
  const uint32_t ops_len = __shfl(c.hi, 4, 16), tx_ops_len = __shfl(c.hi, 5, 16), flags = __shfl(c.hi, 11, 16);
  if (ai >= p.alns_cap || o + ops_len + tx_ops_len > p.ops_cap) return ops_len + tx_ops_len;  // never without a fault; keeps every store in bounds
  // 64 bytes per step of the 16 lanes, the four loads of a lane in flight together (a byte at a time would be one
  // memory round trip per 16 bytes)
  auto copy_ops = [&](const uint8_t* src, uint8_t* dst, uint32_t len) {
    #pragma unroll 1
    for (uint32_t b0 = 0; b0 < len; b0 += 64) {
      uint8_t v[4];
      #pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t b = b0 + (uint32_t)sub + 16u * (uint32_t)k;
        v[k] = (b < len) ? src[b] : (uint8_t)0;
      }
      #pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t b = b0 + (uint32_t)sub + 16u * (uint32_t)k;
        if (b < len) dst[b] = v[k];
      }
    }
  };
  copy_ops(p.cand_ops + ops_off, p.ops + o, ops_len);
  const uint64_t go = o, to = o + ops_len;
  copy_ops(p.cand_ops + tx_ops_off, p.ops + to, tx_ops_len);
  // the record: thm_aln dword i comes from Cand dword src(i), or is one of ops_off / tx_ops_off / xlen / the flag bytes
  //   i        0  1  2  3  4  5  6  7  8  9 10 11 12 13 14 15 | 16 17 18 19 20 21 22 23 24 25 26 27
  //   src(i)   0  1  2  3  4  5  go go 6  7  8  9 10 11 to to | 16 17 18 19 xl 20 22 23 24 25 21 fl
  const int src_lo = (int)((0x00BA987600543210ull >> (4 * sub)) & 15u);
  const int src_hi = (int)((0x0000B59876403210ull >> (4 * sub)) & 15u);  // (lane of c.hi = Cand dword - 16)
  uint32_t w0 = __shfl(c.lo, src_lo, 16), w1 = __shfl(c.hi, src_hi, 16);
  const uint64_t txo = ((flags >> 8) & 0xFFu) == THM_ALN_EXONIC ? to : 0ull;
  if (sub == 6) w0 = (uint32_t)go;
  if (sub == 7) w0 = (uint32_t)(go >> 32);
  if (sub == 14) w0 = (uint32_t)txo;
  if (sub == 15) w0 = (uint32_t)(txo >> 32);
  if (sub == 4) w1 = xlen;
  // Cand: strand, aln_type, primary, pad -> thm_aln: strand, primary (src/aligner.rs:185-187), aln_type, pad
  if (sub == 11) w1 = (flags & 0xFFu) | ((primary ? 1u : 0u) << 8) | (((flags >> 8) & 0xFFu) << 16);
  uint32_t* out = reinterpret_cast<uint32_t*>(p.alns + ai);
  out[sub] = w0;
  if (sub < 12) out[16 + sub] = w1;
  return ops_len + tx_ops_len;
}

// a faulted attempt (pool overflow) is replayed by the host with larger pools: its counts and offsets are not to be trusted
__device__ __forceinline__ bool compact_faulted(const CompactParams& p) { return p.fault[0] != 0 || (p.fault[1] & FAULT_OPS_POOL) != 0; }

template <int K>
__global__ __launch_bounds__(256) void compact_kernel(CompactParams p) {
  // 16 lanes per read (four reads per wavefront), the groups striding over the reads: one workgroup per 16 reads
  // (31 250 of them for a batch of 500 000, their waves alive for 2.5 us each) kept a seventh of the wave slots
  // occupied (SQ_WAVE_CYCLES / duration, profiles/r03/sq_counters_bench.json).  A read is a chain of dependent
  // loads (count -> offsets -> order -> candidate -> op bytes), so a group works on K reads at once, each link of
  // the K chains in flight together.
  const int sub = (int)(threadIdx.x & 15u);
  if (compact_faulted(p)) return;
  const uint64_t n_groups = (uint64_t)gridDim.x * 16, g = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  for (uint64_t rb = g; rb < p.n_reads; rb += (uint64_t)K * n_groups) {
    uint32_t n[K], xlen[K], l0[K];
    uint64_t cand0[K], a0[K], o[K];
    #pragma unroll
    for (int k = 0; k < K; k++) {
      const uint64_t r = rb + (uint64_t)k * n_groups;
      n[k] = r < p.n_reads ? p.read_n_alns[r] : 0u;
    }
    #pragma unroll
    for (int k = 0; k < K; k++) {
      const uint64_t r = rb + (uint64_t)k * n_groups;
      if (n[k] > COMPACT_HEAVY_N) {
        if (sub == 0) {
          const unsigned long long h = atomicAdd(&p.heavy_cnt[0], 1ull);
          if (h < p.heavy_cap) p.heavy_list[h] = r;
        }
        n[k] = 0;
      }
      if (n[k]) {
        cand0[k] = p.read_cand_off[r];
        a0[k] = p.read_aln_off[r];
        o[k] = p.read_ops_off[r];
        xlen[k] = (uint32_t)(p.read_offsets[r + 1] - p.read_offsets[r]);
      }
    }
    #pragma unroll
    for (int k = 0; k < K; k++)
      if (n[k]) l0[k] = p.order[2 * cand0[k]];
    CandRegs cd[K];
    #pragma unroll
    for (int k = 0; k < K; k++)
      if (n[k]) cd[k] = cand_load(p.cands + cand0[k] + l0[k], sub);
    #pragma unroll
    for (int k = 0; k < K; k++) {
      if (!n[k]) continue;
      uint64_t ok = o[k] + compact_emit(p, cd[k], a0[k], o[k], xlen[k], true, sub);
      const uint32_t* la = p.order + 2 * cand0[k];
      for (uint32_t t = 1; t < n[k]; t++)  // (one read in forty has a second alignment)
        ok += compact_emit(p, cand_load(p.cands + cand0[k] + la[t], sub), a0[k] + t, ok, xlen[k], false, sub);
    }
  }
}

// heavy reads, step 1: rel[a0 + t] = op bytes of the read's alignments before t; one descriptor per COMPACT_CHUNK alignments
__global__ __launch_bounds__(256) void compact_heavy_scan_kernel(CompactParams p) {
  __shared__ uint32_t wave_sum[4];
  __shared__ unsigned long long desc_base;
  if (compact_faulted(p)) return;
  const uint64_t H = min((unsigned long long)p.heavy_cap, p.heavy_cnt[0]);
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (uint64_t h = blockIdx.x; h < H; h += gridDim.x) {
    const uint64_t r = p.heavy_list[h];
    const uint32_t n = p.read_n_alns[r];
    const uint64_t cand0 = p.read_cand_off[r], a0 = p.read_aln_off[r];
    const Cand* cands = p.cands + cand0;
    const uint32_t* la = p.order + 2 * cand0;
    uint32_t carry = 0;
    for (uint32_t tile = 0; tile < n; tile += 256) {
      const uint32_t t = tile + (uint32_t)tid;
      uint32_t len = 0;
      if (t < n) {
        const Cand* cd = &cands[la[t]];
        len = cd->ops_len + cd->tx_ops_len;
      }
      uint32_t inc = len;  // inclusive scan within the wave
      #pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
      }
      if (lane == 63) wave_sum[wv] = inc;
      __syncthreads();
      uint32_t before = carry, total = 0;
      #pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k < wv) before += wave_sum[k];
        total += wave_sum[k];
      }
      if (t < n && a0 + t < p.alns_cap) p.rel[a0 + t] = before + inc - len;
      carry += total;
      __syncthreads();
    }
    const uint32_t nd = (n + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
    if (tid == 0) desc_base = atomicAdd(&p.heavy_cnt[1], (unsigned long long)nd);
    __syncthreads();
    const unsigned long long base = desc_base;
    for (uint32_t k = (uint32_t)tid; k < nd; k += 256)
      if (base + k < p.heavy_cap) p.heavy_desc[base + k] = (r << 24) | (uint64_t)k;
    __syncthreads();
  }
}

// heavy reads, step 2: a 16-lane group per descriptor
__global__ __launch_bounds__(256) void compact_heavy_copy_kernel(CompactParams p) {
  const int sub = (int)(threadIdx.x & 15u);
  if (compact_faulted(p)) return;
  const uint64_t D = min((unsigned long long)p.heavy_cap, p.heavy_cnt[1]);
  const uint64_t n_groups = (uint64_t)gridDim.x * 16;
  for (uint64_t d = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4; d < D; d += n_groups) {
    const uint64_t desc = p.heavy_desc[d], r = desc >> 24;
    const uint32_t t0 = (uint32_t)(desc & 0xFFFFFFu) * COMPACT_CHUNK, n = p.read_n_alns[r];
    const uint64_t cand0 = p.read_cand_off[r], a0 = p.read_aln_off[r], o0 = p.read_ops_off[r];
    const Cand* cands = p.cands + cand0;
    const uint32_t* la = p.order + 2 * cand0;
    const uint32_t xlen = (uint32_t)(p.read_offsets[r + 1] - p.read_offsets[r]);
    for (uint32_t t = t0; t < n && t < t0 + COMPACT_CHUNK; t++) {
      if (a0 + t >= p.alns_cap) break;
      (void)compact_emit(p, cand_load(&cands[la[t]], sub), a0 + t, o0 + p.rel[a0 + t], xlen, t == 0, sub);
    }
  }
}

}  // namespace dev

size_t extend_lds_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl) {
  const dev::ExtCaps k = dev::ext_caps(max_read_len, max_bw, cpl);
  const uint32_t per_wave = k.lcap + 2u * k.wcap + k.trb + 3u * k.opcap + 8u * FAST_MAX_YCLIPS + (uint32_t)(dev::KEYCAP * sizeof(dev::CandKey));
  return 4 * (size_t)per_wave;
}

// global trace scratch of one wave (extensions over more than 64 band slots), in bytes
size_t extend_trace_scratch_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl) {
  return cpl > 1 ? (size_t)(max_read_len + max_bw + 3u) * (size_t)cpl * 16u : 0;
}

// everything one wave of the any-width kernel keeps in global memory, in bytes
size_t extend_slow_scratch_bytes(uint32_t max_read_len, uint32_t max_bw, uint32_t mk_cap) {
  return (size_t)dev::slow_layout(max_read_len, max_bw, mk_cap).total;
}

constexpr int EXT_MINW_CPL34 = 4;  // default register budget of the three- and four-cell kernels (waves per SIMD)
// waves per SIMD the wave-per-read kernel chosen for (cpl, coordinate width) is compiled for = workgroups of 4 waves
// that fit a CU; the host launches no more than that (a workgroup beyond it starts when the first ones leave, finds
// the work counters dry and only delays the end of the launch)
int extend_waves_per_simd(int cpl, bool wide) {
  static const int minw_env = [] {
    const char* e = getenv("THM_EXT_MINW");
    const int v = e ? atoi(e) : 0;
    return (v >= 4 && v <= 8) ? v : 0;
  }();
  if (cpl == 0) return 2;
  if (cpl > 2) {
    // three and four cells per lane (bands beyond +-63: BASELINE config 5's +-64): tuning knob THM_EXT_MINW_CPL3 = 3 | 4 | 5 | 6
    static const int v34 = [] {
      const char* e = getenv("THM_EXT_MINW_CPL3");
      const int v = e ? atoi(e) : 0;
      return (v >= 3 && v <= 6) ? v : 0;
    }();
    return (v34 && !wide) ? v34 : EXT_MINW_CPL34;
  }
  if (wide) {
    static const int wide_env = [] {
      const char* e = getenv("THM_EXT_MINW_WIDE");
      const int v = e ? atoi(e) : 0;
      return (v == 4 || v == 5) ? v : 0;
    }();
    return wide_env ? wide_env : 5;
  }
  if (minw_env) {
    if (cpl == 1) return (minw_env == 4 || minw_env == 5 || minw_env == 6) ? minw_env : 8;
    return (minw_env == 4 || minw_env == 5 || minw_env == 8) ? minw_env : 6;
  }
  return cpl == 1 ? 6 : 5;
}

template <class C>
static hipError_t launch_extend_t(const ExtendParamsT<C>& p, int cpl, int n_blocks, hipStream_t s, bool team) {
  const size_t lds4 = cpl == 0 ? 0 : extend_lds_bytes(p.max_read_len, p.max_bw, cpl);
  const size_t lds = team ? lds4 / 4 * TEAM_WAVES : lds4;
  const unsigned threads = team ? 64u * TEAM_WAVES : 256u;
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(threads), lds, s, p);
    return hipGetLastError();
  };
  if (team) {
    if (cpl == 1) return go(dev::extend_kernel<C, 1, 4, TEAM_WAVES>);
    if (cpl == 2) return go(dev::extend_kernel<C, 2, 4, TEAM_WAVES>);
    return hipErrorInvalidValue;
  }
  // register budget: MINW waves per SIMD.  Measured on the 32-bit-coordinate kernels (round 2, benchmark workload,
  // same box).  One cell per lane: 3.67 ms at 6 waves against 3.78 at 5, 3.90 at 8 (560 bytes per lane of spills), 3.96
  // at 4 -> 6.  Two cells per lane: 3.89 ms at 6 (80 VGPRs, 136 bytes per lane of spills), 3.96 at 5 (96 VGPRs, 76 bytes),
  // 4.30 at 7, 4.23 at 8, 4.64 at 4; but at 6 waves the spills move 3.2 GB per 500 k-read launch through the memory
  // system (rocprofv3 FETCH_SIZE / WRITE_SIZE) against 1.2 GB at 5 and 0.53 GB of algorithmic bytes: 1 % of throughput
  // for 2.6x less traffic -> 5.  Wider bands run at 4.  Tuning knob THM_EXT_MINW = 4 | 5 | 6 | 8 for the one- and
  // two-cell kernels.  The 64-bit-coordinate kernels (more live state per hit) were at 4 waves per SIMD until the end of
  // round 2: 5 is 8 % faster (4.29 -> 3.95 ms on the benchmark workload with THM_FORCE_WIDE; knob THM_EXT_MINW_WIDE = 4 | 5).
  if (cpl == 0) return go(dev::extend_kernel<C, 0, 2>);
  if constexpr (sizeof(C) == 8) {
    const int minw = extend_waves_per_simd(cpl, true);
    switch (cpl) {
      case 1: return minw == 5 ? go(dev::extend_kernel<C, 1, 5>) : go(dev::extend_kernel<C, 1, 4>);
      case 2: return minw == 5 ? go(dev::extend_kernel<C, 2, 5>) : go(dev::extend_kernel<C, 2, 4>);
      case 3: return go(dev::extend_kernel<C, 3, 4>);
      case 4: return go(dev::extend_kernel<C, 4, 4>);
      default: return hipErrorInvalidValue;
    }
  } else {
    const int minw = extend_waves_per_simd(cpl, false);
    switch (cpl) {
      case 1:
        if (minw == 4) return go(dev::extend_kernel<C, 1, 4>);
        if (minw == 5) return go(dev::extend_kernel<C, 1, 5>);
        if (minw == 6) return go(dev::extend_kernel<C, 1, 6>);
        return go(dev::extend_kernel<C, 1, 8>);
      case 2:
        if (minw == 4) return go(dev::extend_kernel<C, 2, 4>);
        if (minw == 5) return go(dev::extend_kernel<C, 2, 5>);
        if (minw == 8) return go(dev::extend_kernel<C, 2, 8>);
        return go(dev::extend_kernel<C, 2, 6>);
      case 3:
      case 4: {
        const int minw = extend_waves_per_simd(cpl, false);
        if (cpl == 3) {
          if (minw == 3) return go(dev::extend_kernel<C, 3, 3>);
          if (minw == 5) return go(dev::extend_kernel<C, 3, 5>);
          if (minw == 6) return go(dev::extend_kernel<C, 3, 6>);
          return go(dev::extend_kernel<C, 3, 4>);
        }
        if (minw == 3) return go(dev::extend_kernel<C, 4, 3>);
        if (minw == 5) return go(dev::extend_kernel<C, 4, 5>);
        if (minw == 6) return go(dev::extend_kernel<C, 4, 6>);
        return go(dev::extend_kernel<C, 4, 4>);
      }
      default: return hipErrorInvalidValue;
    }
  }
}
hipError_t launch_extend(const ExtendParamsT<uint32_t>& p, int cpl, int n_blocks, hipStream_t s, bool team) { return launch_extend_t(p, cpl, n_blocks, s, team); }
hipError_t launch_extend(const ExtendParamsT<uint64_t>& p, int cpl, int n_blocks, hipStream_t s, bool team) { return launch_extend_t(p, cpl, n_blocks, s, team); }

hipError_t launch_counters_reduce(const unsigned long long* wave_counters, uint32_t n_rows, unsigned long long* counters, hipStream_t s) {
  if (n_rows == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::counters_reduce_kernel, dim3(1), dim3(1024), 0, s, wave_counters, n_rows, counters);
  return hipGetLastError();
}

hipError_t launch_compact(const CompactParams& p, hipStream_t s) {
  static const unsigned n_cu = [] {
    int dev = 0, cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    return (unsigned)cu;
  }();
  // 8 workgroups of 256 threads fill a CU's wave slots
  const unsigned blocks = (unsigned)std::min<uint64_t>((p.n_reads + 15) / 16, (uint64_t)n_cu * 8);
  if (blocks == 0) return hipSuccess;
  static const int k_reads = [] {
    const char* e = getenv("THM_COMPACT_K");
    const int v = e ? atoi(e) : 2;
    return v == 1 || v == 2 || v == 4 ? v : 2;
  }();
  if (k_reads == 1)
    hipLaunchKernelGGL(dev::compact_kernel<1>, dim3(blocks), dim3(256), 0, s, p);
  else if (k_reads == 2)
    hipLaunchKernelGGL(dev::compact_kernel<2>, dim3(blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(dev::compact_kernel<4>, dim3(blocks), dim3(256), 0, s, p);
  // the reads left on the heavy list (counts on the device: both launches find them empty for most batches)
  hipLaunchKernelGGL(dev::compact_heavy_scan_kernel, dim3(n_cu), dim3(256), 0, s, p);
  hipLaunchKernelGGL(dev::compact_heavy_copy_kernel, dim3(n_cu * 4), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace thm
