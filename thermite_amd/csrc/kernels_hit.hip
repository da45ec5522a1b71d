// kernels_hit.hip -- hit summaries: what align_seed_hit (reference src/aligner.rs:198-314) does with a seed hit that does
// not depend on the state align_read carries from hit to hit.
//
// Band, X-drop and best score change while a read's hits are taken in order (src/aligner.rs:143-175), so the hits of a
// read cannot simply be aligned side by side.  But most of the work on a hit never looks at that state: the contig of
// the hit (Index::idx_to_ref, src/index.rs:287-290), the exon intervals over the seed in exon_to_tx.find's yield order
// (:231-236), lift_mem_to_tx (src/txome.rs:82-103), extend_seed_match (src/aligner.rs:410-426) -- and what each
// SwgExtend::extend call is going to meet: nothing (empty x or y), one mismatch next to the seed and an exact match
// behind it (the result is known, swg_device.h::swg_one_mismatch_shortcut), or a real DP.  The band enters only through
// the number of y symbols an extension is given, min(A, |x| + bw + 1) with A the symbols the target has left: every
// classification below is stated in terms of A alone (|y| >= |x| iff A >= |x|, and so on).
//
// So one kernel walks every hit of the batch once, all hits side by side: a small group of lanes per hit, sixteen
// bytes per lane -- a comparison of up to 128 read bases with the text is one load per operand for the whole group,
// without a loop whose trip count differs from hit to hit (the lanes of a wavefront step in lockstep: a thread per hit
// with its own byte loops runs every wavefront as long as its slowest lane, and every one of its loads costs a cache
// line look-up per lane).  The result is a HitSum per hit (launch.h); the control kernel (kernels_tpr.hip), one
// thread per read, replays a read's hits from these records and turns the extensions that need a DP into requests.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.h"
#include "swg_device.h"

namespace thm {
namespace dev {

namespace {

#ifndef THM_HIT_GL  // the default build: four lanes per hit, and the launchers that pick a variant
#define THM_HIT_GL 4
#define THM_HIT_MAIN
#endif
constexpr int GL = THM_HIT_GL;  // lanes per hit (a power of two up to 8)
constexpr int ENT_PER_LANE = 24 / GL;  // exon-grid entries a lane holds: 24 per query in all

struct V16 {
  uint64_t lo, hi;
};
__device__ __forceinline__ V16 ldg16(const uint8_t* p) {
  V16 v;
  __builtin_memcpy(&v, p, 16);  // one global_load_dwordx4 (unaligned access is enabled for global memory)
  return v;
}
// keep the first vb bytes (0..16) of a 16-byte difference mask
__device__ __forceinline__ void keep_first(uint64_t& lo, uint64_t& hi, int vb) {
  if (vb <= 0) {
    lo = hi = 0;
  } else if (vb < 8) {
    lo &= (1ull << (8 * vb)) - 1ull;
    hi = 0;
  } else if (vb == 8) {
    hi = 0;
  } else if (vb < 16) {
    hi &= (1ull << (8 * (vb - 8))) - 1ull;
  }
}
__device__ __forceinline__ int first_byte(uint64_t lo, uint64_t hi) { return lo ? (int)(__builtin_ctzll(lo) >> 3) : 8 + (int)(__builtin_ctzll(hi) >> 3); }
__device__ __forceinline__ int last_byte(uint64_t lo, uint64_t hi) { return hi ? 8 + (int)((63 - __builtin_clzll(hi)) >> 3) : (int)((63 - __builtin_clzll(lo)) >> 3); }

struct Grp {
  int gl;       // lane within the group
  int gbase;    // first lane of the group within the wavefront
  // the group's eight bits of a wavefront ballot (every lane of a group is active together: control flow is
  // uniform within a group)
  __device__ __forceinline__ unsigned ballot(bool p) const { return (unsigned)((__ballot(p) >> gbase) & ((1ull << GL) - 1ull)); }
  __device__ __forceinline__ int bcast(int v, int l) const { return __shfl(v, gbase + l); }
  __device__ __forceinline__ int gmin(int v) const {
    if (GL > 1) v = min(v, __shfl_xor(v, 1));
    if (GL > 2) v = min(v, __shfl_xor(v, 2));
    if (GL > 4) v = min(v, __shfl_xor(v, 4));
    return v;
  }
};

// leading positions t < n with a[t] == b[t] (may read up to 15 bytes behind a + n / b + n: every array these are used on
// carries that much padding)
__device__ int g_match_fwd(const Grp& g, const uint8_t* a, const uint8_t* b, int n) {
  for (int base = 0; base < n; base += 16 * GL) {
    const int off = base + g.gl * 16, vb = min(16, n - off);
    uint64_t d0 = 0, d1 = 0;
    if (vb > 0) {
      const V16 x = ldg16(a + off), y = ldg16(b + off);
      d0 = x.lo ^ y.lo;
      d1 = x.hi ^ y.hi;
      keep_first(d0, d1, vb);
    }
    const bool has = (d0 | d1) != 0;
    const unsigned m = g.ballot(has);
    if (m) {
      const int l = __builtin_ctz(m);
      const int idx = g.bcast(has ? first_byte(d0, d1) : 0, l);
      return base + l * 16 + idx;
    }
  }
  return n;
}
// positions t < n with a[-1 - t] == b[-1 - t], walking backwards from a and b (exclusive); reads nothing below a - n / b - n
__device__ int g_match_bwd(const Grp& g, const uint8_t* a, const uint8_t* b, int n) {
  for (int done = 0; done < n; done += 16 * GL) {
    const int t0 = done + g.gl * 16;  // this lane: t in [t0, t0 + 16), the 16 bytes that end at a - t0
    const int vb = min(16, n - t0);
    uint64_t d0 = 0, d1 = 0;
    int tmis = 0;
    if (vb > 0) {
      // a short last chunk is loaded from the range's first byte instead (its wanted bytes come first then)
      const int back = vb == 16 ? t0 + 16 : n;
      const V16 x = ldg16(a - back), y = ldg16(b - back);
      d0 = x.lo ^ y.lo;
      d1 = x.hi ^ y.hi;
      keep_first(d0, d1, vb);
      if (d0 | d1) tmis = t0 + (vb - 1 - last_byte(d0, d1));
    }
    const bool has = (d0 | d1) != 0;
    const unsigned m = g.ballot(has);
    if (m) return g.bcast(tmis, __builtin_ctz(m));
  }
  return n;
}
// a[0 .. n) == b[0 .. n); `uniform`: every a[t] equals c
__device__ bool g_equal_uniform(const Grp& g, const uint8_t* a, const uint8_t* b, int n, uint8_t c, bool& uniform) {
  const uint64_t splat = 0x0101010101010101ull * (uint64_t)c;
  bool diff = false, nonu = false;
  for (int base = 0; base < n; base += 16 * GL) {
    const int off = base + g.gl * 16, vb = min(16, n - off);
    if (vb > 0) {
      const V16 x = ldg16(a + off), y = ldg16(b + off);
      uint64_t d0 = x.lo ^ y.lo, d1 = x.hi ^ y.hi, u0 = x.lo ^ splat, u1 = x.hi ^ splat;
      keep_first(d0, d1, vb);
      keep_first(u0, u1, vb);
      diff = diff || (d0 | d1) != 0;
      nonu = nonu || (u0 | u1) != 0;
    }
  }
  uniform = g.ballot(nonu) == 0;
  return g.ballot(diff) == 0;
}

// One extension before any DP.  x0 / y0: the first symbols as the extension walks them, dir = +1 (right) or -1 (left):
// symbol t is x0[t * dir].  A: symbols the target has left in that direction.
__device__ HitSide g_classify(const Grp& g, const uint8_t* x0, const uint8_t* y0, int dir, int xlen, long long A) {
  HitSide s;
  s.eq = 0;
  s.A = (uint32_t)(A < 0 ? 0 : (A > 0xFFFFFFFFll ? 0xFFFFFFFFll : A));
  s.kind = SK_EMPTY;
  if (xlen == 0 || A <= 0) return s;
  const uint8_t bx = x0[0], by = y0[0];
  if (bx == by) {
    s.kind = SK_UNK_EQ;
    return s;
  }
  if (xlen == 1) {
    s.kind = SK_SINGLE;
    return s;
  }
  // the rest of x and y as forward ranges: a left extension walks both backwards, so its symbols 1 .. n - 1 are the n - 1
  // bytes BEFORE the first ones
  const int n1 = xlen - 1;
  if (xlen >= 3 && A >= (long long)xlen) {
    const uint8_t* xa = dir > 0 ? x0 + 1 : x0 - n1;
    const uint8_t* ya = dir > 0 ? y0 + 1 : y0 - n1;
    bool uniform;
    if (g_equal_uniform(g, xa, ya, n1, bx, uniform) && !uniform) {
      s.kind = SK_SHORTCUT;
      return s;
    }
  }
  s.kind = SK_UNK;
  if (A >= (long long)xlen + 1) {  // x == y[1 .. |x| + 1): one leading deletion, which costs no gap-open
    const uint8_t* xf = dir > 0 ? x0 : x0 - n1;
    const uint8_t* yf = dir > 0 ? y0 + 1 : y0 - 1 - n1;
    bool dummy;
    if (g_equal_uniform(g, xf, yf, xlen, 0, dummy)) s.kind = SK_UNK_DEL;
  }
  return s;
}

__device__ __forceinline__ int side_score(const HitSide& s, int xlen) { return s.kind == SK_SHORTCUT ? xlen - 2 : 0; }  // kinds <= SK_SHORTCUT

}  // namespace

#ifdef THM_HIT_MAIN
// (read, q, len, hr) of every hit of the reads this path takes: thread per read (such a read has few hits)
template <class C>
__global__ __launch_bounds__(256) void hit_expand_kernel(HitParamsT<C> p) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= p.reads.n_reads || *p.fault_seed != 0) return;
  const ReadRecT<C> rec = p.read_recs[r];
  if (rec.len > p.max_read_len || rec.n_hits >= p.max_hits || rec.cand_off + rec.n_hits > p.slot_cap) return;
  uint64_t slot = rec.cand_off;
  for (uint32_t si = 0; si < rec.smem_cnt; si++) {
    SmemT<C> sm;
    if (si == 0) {
      sm.lo = rec.lo0;
      sm.hi = rec.hi0;
      sm.qpos = rec.qpos0;
      sm.len = rec.len0;
    } else {
      sm = p.smems[rec.smem_off + si];
    }
    for (C rr = sm.hi; rr > sm.lo; rr--) {  // occurrences by descending suffix-array rank (src/index.rs:236-248 + the reverse at :253)
      HitHdr h;
      h.hr = (uint64_t)((si == 0 && rr == sm.hi) ? rec.sa0 : p.ix.sa[rr - 1]);
      h.read = (uint32_t)r;
      h.q = sm.qpos;
      h.len = sm.len;
      p.hdr[slot++] = h;
    }
  }
}

#endif
template <class C>
struct HCoord {
  typedef int S;
};
template <>
struct HCoord<uint64_t> {
  typedef long long S;
};

#define HIT_CAT2(a, b) a##b
#define HIT_CAT(a, b) HIT_CAT2(a, b)
#define hit_summary_kernel_gl HIT_CAT(hit_summary_kernel_gl, THM_HIT_GL)
template <class C>
__global__ __launch_bounds__(256, 4) void hit_summary_kernel_gl(HitParamsT<C> p) {
  typedef typename HCoord<C>::S S;  // signed coordinates
  const auto& ix = p.ix;
  Grp g;
  g.gl = (int)(threadIdx.x & (GL - 1));
  g.gbase = (int)(threadIdx.x & 63u & ~(unsigned)(GL - 1));
  const uint64_t n_slots = min(*p.total_hits, p.slot_cap);
  const uint64_t groups = (uint64_t)gridDim.x * (256 / GL);
  for (uint64_t slot = (uint64_t)blockIdx.x * (256 / GL) + (threadIdx.x / GL); slot < n_slots; slot += groups) {
    const HitHdr hd = p.hdr[slot];
    if (hd.read == 0xFFFFFFFFu) continue;
    const ReadRecT<C>* rp = p.read_recs + hd.read;
    const uint64_t base_off = rp->base_off;
    const int L = (int)rp->len;
    const uint8_t* rd = p.reads.bases + base_off;
    const int q = hd.q, len = hd.len;
    const S hr = (S)hd.hr;
    // the widest band the read can meet: the one it starts with (src/aligner.rs:130-138)
    const float prod = p.opts.min_aln_score_percent * (float)L;
    const int ms_pct = (prod != prod) ? 0 : (prod >= 2147483648.0f ? 2147483647 : (prod <= -2147483648.0f ? (-2147483647 - 1) : (int)prod));
    const int min_aln_score = max(ms_pct, p.opts.min_aln_score);
    const int bw0 = (min_aln_score < 0) ? 0 : max(L - min_aln_score, 0);

    HitSum* const out = p.sums + slot;
    struct {
      uint32_t ref_id, win_fixed;
      HitSide gr, gl;
      uint8_t win_nl, win_nr, win_nvl, win_nvr, n_tgt, n_open, flags, why;
      uint16_t win_var[HIT_MAX_VAR];
      HitTgt known;
    } o;
    o.win_fixed = 0;
    o.win_nl = o.win_nr = o.win_nvl = o.win_nvr = 0;
    for (int k = 0; k < HIT_MAX_VAR; k++) o.win_var[k] = 0;
    o.n_tgt = o.n_open = 0;
    o.flags = 0;
    o.why = 0;
    // Index::idx_to_ref: the contig copy that holds the bin's first symbol, then forward over the boundaries inside the bin
    uint32_t rid = ix.ref_bin[(C)hr >> GRID_SHIFT];
    RefRecT<C> ref = ix.ref_recs[rid];
    while (ref.end <= (C)hr && rid + 1 < ix.n_refs) {
      rid++;
      ref = ix.ref_recs[rid];
    }
    o.ref_id = rid;
    // genome window (:212-215): the extensions may go up to the contig copy's ends
    const int xr = L - (q + len), xl = q;
    const S g_Ar = (S)ref.end - 1 - (hr + len), g_Al = hr - (S)ref.start;
    o.gr = g_classify(g, rd + q + len, ix.text + (hr + len), 1, xr, g_Ar);
    o.gl = g_classify(g, rd + q - 1, ix.text + (hr - 1), -1, xl, g_Al);
    const bool gr_unk = o.gr.kind > SK_SHORTCUT, gl_unk = o.gl.kind > SK_SHORTCUT;

    // exon_to_tx.find(seed) (:231-236): by ascending pre-order rank.  Up to 24 entries of one or two bins, three per lane.
    const C qs = (C)hr, qe = (C)(hr + len);
    const uint32_t b0 = (uint32_t)(qs >> GRID_SHIFT), b1 = (uint32_t)((qe > qs ? qe - 1 : qs) >> GRID_SHIFT);
    const uint32_t e0 = ix.exon_grid_off[b0], e1 = ix.exon_grid_off[b1 + 1];
    const uint32_t cnt = e1 - e0;
    const ExonEntryT<C>* ent = ix.exon_grid + e0;
    int my_rank[ENT_PER_LANE];
    if (cnt > (uint32_t)(ENT_PER_LANE * GL) || b1 > b0 + 1) {
      o.flags |= HF_COMPLEX;
      o.why = 2;
    } else {
#pragma unroll
      for (int k = 0; k < ENT_PER_LANE; k++) {
        const uint32_t t = (uint32_t)(g.gl + k * GL);
        my_rank[k] = 0x7fffffff;
        if (t < cnt) {
          const C es = ent[t].start, ee = ent[t].end;
          const uint32_t rk = ent[t].rank;
          const uint32_t home = max(b0, (uint32_t)(es >> GRID_SHIFT));  // the copy listed in the interval's first queried bin counts
          if (qs < ee && es < qe && (rk & 0xffu) == (home & 0xffu)) my_rank[k] = (int)(rk >> 8);
        }
      }
    }
    o.known.tx = o.known.ent = o.known.pos = 0;
    o.known.tr = 0;
    o.known.t_q = o.known.t_len = 0;
    o.known.r = o.known.l = o.gr;
    int known_score = -1;
    // the open targets, as far as a later target that is an earlier one again is recognised by
    const uint8_t* op_seq[HIT_MAX_OPEN];
    int op_tr[HIT_MAX_OPEN];
    uint32_t op_qlen[HIT_MAX_OPEN], op_Ar[HIT_MAX_OPEN], op_Al[HIT_MAX_OPEN];
    for (int k = 0; k < HIT_MAX_OPEN; k++) {
      op_seq[k] = nullptr;
      op_tr[k] = 0;
      op_qlen[k] = op_Ar[k] = op_Al[k] = 0;
    }
    int last = -1, pos = 0;
    bool stop = (o.flags & HF_COMPLEX) != 0;
    while (!stop) {
      int cand = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < ENT_PER_LANE; k++)
        if (my_rank[k] > last) cand = min(cand, my_rank[k]);
      const int best = g.gmin(cand);
      if (best == 0x7fffffff) break;
      last = best;
      int my_ei = -1;
#pragma unroll
      for (int k = 0; k < ENT_PER_LANE; k++)
        if (my_rank[k] == best) my_ei = g.gl + k * GL;
      const unsigned own = g.ballot(my_ei >= 0);
      const uint32_t ei = (uint32_t)g.bcast(my_ei, __builtin_ctz(own));
      const ExonEntryT<C> ge = ent[ei];
      if (!(ge.prev_end <= qs)) {  // lift_mem_to_tx's general case (a seed across a short intron)
        o.flags |= HF_COMPLEX;
        o.why = 3;
        break;
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
      // the target's window, [tr - (L + bw), tr + len + L + bw + 1) cut to the transcript: which of its sides depend on the band?
      {
        o.win_fixed += (uint32_t)t_len;
        const int availl = tr_, availr = tlen - (tr_ + t_len);  // left side: min(availl, L + bw); right side: min(availr, L + bw + 1)
        bool over = false;
        if (availl >= L + bw0) {
          o.win_nl++;
        } else if (availl <= L) {
          o.win_fixed += (uint32_t)availl;
        } else if (o.win_nvl + o.win_nvr < HIT_MAX_VAR) {
          // left sides are listed before right ones
          for (int k = o.win_nvl + o.win_nvr; k > o.win_nvl; k--) o.win_var[k] = o.win_var[k - 1];
          o.win_var[o.win_nvl++] = (uint16_t)availl;
        } else {
          over = true;
        }
        if (availr >= L + bw0 + 1) {
          o.win_nr++;
        } else if (availr <= L + 1) {
          o.win_fixed += (uint32_t)availr;
        } else if (o.win_nvl + o.win_nvr < HIT_MAX_VAR) {
          o.win_var[o.win_nvl + o.win_nvr] = (uint16_t)availr;
          o.win_nvr++;
        } else {
          over = true;
        }
        if (over || o.n_tgt == 255) {
          o.flags |= HF_COMPLEX;
          o.why = 13;
          break;
        }
      }
      o.n_tgt++;
      const uint8_t* seq = ix.tx_seq + ge.seq_off;
      // extend_seed_match (src/aligner.rs:410-426)
      {
        int ext = g_match_fwd(g, seq + tr_ + t_len, rd + t_q + t_len, min(tlen - (tr_ + t_len), L - (t_q + t_len)));
        t_len += ext;
        ext = g_match_bwd(g, seq + tr_, rd + t_q, min(tr_, t_q));
        tr_ -= ext;
        t_q -= ext;
        t_len += ext;
      }
      HitTgt t;
      t.tx = ge.value;
      t.ent = e0 + ei;
      t.tr = tr_;
      t.t_q = (uint16_t)t_q;
      t.t_len = (uint16_t)t_len;
      t.pos = (uint32_t)pos;
      const int t_xr = L - (t_q + t_len), t_xl = t_q;
      const S t_Ar = (S)tlen - (S)(tr_ + t_len), t_Al = (S)tr_;
      t.r = g_classify(g, rd + t_q + t_len, seq + (tr_ + t_len), 1, t_xr, t_Ar);
      t.l = g_classify(g, rd + t_q - 1, seq + (tr_ - 1), -1, t_xl, t_Al);
      pos++;
      const bool r_unk = t.r.kind > SK_SHORTCUT, l_unk = t.l.kind > SK_SHORTCUT;
      if (!r_unk && !l_unk) {
        const int sc = side_score(t.l, t_xl) + t_len * MATCH_SCORE + side_score(t.r, t_xr);
        if (sc > known_score) {  // strictly better (:249): the first of the best
          known_score = sc;
          o.known = t;
          o.flags |= HF_KNOWN;
        }
        if (sc >= L * MATCH_SCORE) stop = true;  // cannot beat an exact match (:253-257); a target that needs a DP stays below L
        continue;
      }
      // how far the target's y agrees with the genome window's, for extensions of the same x that both need a DP: with
      // equal y lengths and an agreement at least that long the two extend() calls are one problem
      if (r_unk && gr_unk && t_q + t_len == q + len) {
        const int n = (int)max(min(min(t_Ar, g_Ar), (S)(xr + bw0 + 1)), (S)0);
        t.r.eq = (uint16_t)min(g_match_fwd(g, seq + (tr_ + t_len), ix.text + (hr + len), n), 65535);
      }
      if (l_unk && gl_unk && t_q == q) {
        const int n = (int)max(min(min(t_Al, g_Al), (S)(xl + bw0 + 1)), (S)0);
        t.l.eq = (uint16_t)min(g_match_bwd(g, seq + tr_, ix.text + hr, n), 65535);
      }
      // an earlier open target with the same seed and, as far as any band can reach, the same y on both sides: the same
      // two problems, the same score, and the earlier target keeps a tie -- nothing to add
      bool again = false;
#pragma unroll
      for (int k = 0; k < HIT_MAX_OPEN; k++) {
        if (k >= (int)o.n_open || again) continue;
        if (op_qlen[k] != ((uint32_t)t_q | ((uint32_t)t_len << 16))) continue;
        const long long nr = (long long)(t_xr + bw0 + 1), nl = (long long)(t_xl + bw0 + 1);
        const long long u_Ar = (long long)op_Ar[k], u_Al = (long long)op_Al[k], v_Ar = (long long)t.r.A, v_Al = (long long)t.l.A;
        bool same = (u_Ar == v_Ar || (u_Ar >= nr && v_Ar >= nr)) && (u_Al == v_Al || (u_Al >= nl && v_Al >= nl));
        if (same && t_xr > 0) {
          const int n = (int)min(v_Ar, nr);
          same = n <= 0 || g_match_fwd(g, seq + (tr_ + t_len), op_seq[k] + (op_tr[k] + t_len), n) == n;
        }
        if (same && t_xl > 0) {
          const int n = (int)min(v_Al, nl);
          same = n <= 0 || g_match_bwd(g, seq + tr_, op_seq[k] + op_tr[k], n) == n;
        }
        again = same;
      }
      if (again) continue;
      if (o.n_open >= HIT_MAX_OPEN) {
        o.flags |= HF_COMPLEX;
        o.why = 14;
        break;
      }
#pragma unroll
      for (int k = 0; k < HIT_MAX_OPEN; k++)
        if (k == (int)o.n_open) {
          op_seq[k] = seq;
          op_tr[k] = tr_;
          op_qlen[k] = (uint32_t)t_q | ((uint32_t)t_len << 16);
          op_Ar[k] = t.r.A;
          op_Al[k] = t.l.A;
        }
      if (g.gl == 0) out->open[o.n_open] = t;
      o.n_open++;
    }
    if (g.gl == 0) {
      out->hr = hd.hr;
      out->ref_id = o.ref_id;
      out->q = hd.q;
      out->len = hd.len;
      out->gr = o.gr;
      out->gl = o.gl;
      out->win_fixed = o.win_fixed;
      out->win_nl = o.win_nl;
      out->win_nr = o.win_nr;
      out->win_nvl = o.win_nvl;
      out->win_nvr = o.win_nvr;
      for (int k = 0; k < HIT_MAX_VAR; k++) out->win_var[k] = o.win_var[k];
      out->n_tgt = o.n_tgt;
      out->n_open = o.n_open;
      out->flags = o.flags;
      out->why = o.why;
      out->known = o.known;
    }
  }
}

}  // namespace dev

#ifdef THM_HIT_MAIN
template <class C>
static hipError_t launch_hit_expand_t(const HitParamsT<C>& p, hipStream_t s) {
  const unsigned blocks = (unsigned)((p.reads.n_reads + 255) / 256);
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(dev::hit_expand_kernel<C>, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_hit_expand(const HitParamsT<uint32_t>& p, hipStream_t s) { return launch_hit_expand_t(p, s); }
hipError_t launch_hit_expand(const HitParamsT<uint64_t>& p, hipStream_t s) { return launch_hit_expand_t(p, s); }

#endif
#define launch_hit_summaries_gl HIT_CAT(launch_hit_summaries_gl, THM_HIT_GL)
template <class C>
static hipError_t launch_hit_summaries_t(const HitParamsT<C>& p, int n_blocks, hipStream_t s) {
  if (n_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(dev::hit_summary_kernel_gl<C>, dim3(n_blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_hit_summaries_gl(const HitParamsT<uint32_t>& p, int n_blocks, hipStream_t s) { return launch_hit_summaries_t(p, n_blocks, s); }
hipError_t launch_hit_summaries_gl(const HitParamsT<uint64_t>& p, int n_blocks, hipStream_t s) { return launch_hit_summaries_t(p, n_blocks, s); }

#ifdef THM_HIT_MAIN
// lanes per hit: 4 unless THM_HIT_GL = 1 | 2 | 8 says otherwise (tuning)
#define HIT_DECL(n)                                                                                 \
  hipError_t launch_hit_summaries_gl##n(const HitParamsT<uint32_t>& p, int n_blocks, hipStream_t s); \
  hipError_t launch_hit_summaries_gl##n(const HitParamsT<uint64_t>& p, int n_blocks, hipStream_t s);
HIT_DECL(1) HIT_DECL(2) HIT_DECL(8)
static int hit_gl() {
  static const int v = [] {
    const char* e = getenv("THM_HIT_GL");
    const int x = e ? atoi(e) : 0;
    return (x == 1 || x == 2 || x == 8) ? x : 4;
  }();
  return v;
}
template <class C>
static hipError_t hit_dispatch(const HitParamsT<C>& p, int n_blocks, hipStream_t s) {
  switch (hit_gl()) {
    case 1: return launch_hit_summaries_gl1(p, n_blocks, s);
    case 2: return launch_hit_summaries_gl2(p, n_blocks, s);
    case 8: return launch_hit_summaries_gl8(p, n_blocks, s);
    default: return launch_hit_summaries_gl4(p, n_blocks, s);
  }
}
hipError_t launch_hit_summaries(const HitParamsT<uint32_t>& p, int n_blocks, hipStream_t s) { return hit_dispatch(p, n_blocks, s); }
hipError_t launch_hit_summaries(const HitParamsT<uint64_t>& p, int n_blocks, hipStream_t s) { return hit_dispatch(p, n_blocks, s); }
#endif

}  // namespace thm
