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

// One SwgExtend::extend call as the control kernel sees it.  rec < 0: the result is known in closed form (ops in
// walk order from the seed outwards: Subst if sp == 0, then Match).  rec >= 0: a DP record.
struct Side {
  int score, xend, yend, n, sp;
  int rec;     // DP record index, -1: closed form
  int ub;      // upper bound of the score (== score when known)
  bool known;  // closed form
};

// Classification of one extension.  x0 / y0: the first symbols as the extension walks them; dir = +1 (right) or -1.
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

// the ops of a path that are not Match: (path index << 2 | op kind), ascending
constexpr int TPR_MAX_ED = 2 * DP_MAX_EDITS;
struct EdList {
  uint16_t e[TPR_MAX_ED];
  int n;
};

// what the kernel keeps of an accepted candidate until the read is finished
struct TCand {
  uint64_t ch0, ch1, ylen;  // chromosome coordinates
  int score, xstart, xend, nops;
  uint16_t ed[TPR_MAX_ED];  // the path: rev(left.ops) ++ Match x len ++ right.ops (src/aligner.rs:388-394), Match but for these
  uint32_t ref_id, name_rank, type_idx;
  uint32_t mk_k[TPR_MAX_MK], ycl[TPR_MAX_MK];  // introns: path index they precede, length
  int tx_ystart, tx_yend;
  uint32_t tx_ylen;
  uint8_t strand, aln_type, rev, n_y, n_ed;
};

// One op stream: [Xclip(lead)] path with introns [Xclip(trail)], mirrored as a list when `rev`
// (kernels_extend.hip::emit_alignment).  Paths are Match almost everywhere: zero fill, then patches.
// `o` is 4-byte aligned and the allocation is rounded up to 4 bytes.
__device__ __forceinline__ void emit_stream(uint8_t* o, int total, const uint16_t* ed, int n_ed, int lead, int trail, bool rev, int n_y,
                                            const uint32_t* mk_k, const uint32_t* ycl) {
  static_assert(OPK_MATCH == 0, "zero fill");
  uint32_t* o4 = (uint32_t*)o;
  for (int t = 0; t * 4 < total; t++) o4[t] = 0u;
  const int lead5 = lead > 0 ? 5 : 0;
  auto put5 = [&](int fpos, uint8_t kind, uint32_t v) {
    uint8_t* d = o + (rev ? total - (fpos + 5) : fpos);
    d[0] = kind;
    d[1] = (uint8_t)v;
    d[2] = (uint8_t)(v >> 8);
    d[3] = (uint8_t)(v >> 16);
    d[4] = (uint8_t)(v >> 24);
  };
  for (int i = 0; i < TPR_MAX_ED; i++) {
    if (i >= n_ed) break;
    const int k = (int)(ed[i] >> 2);
    int b = 0;  // introns before op k
    for (int m = 0; m < TPR_MAX_MK; m++) b += (m < n_y && (int)mk_k[m] <= k) ? 1 : 0;
    const int fpos = lead5 + k + 5 * b;
    o[rev ? total - (fpos + 1) : fpos] = (uint8_t)(ed[i] & 3u);
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
// quantities a workgroup allocates in one go: cand op bytes, records, sleepers, bails to the heavy list,
// bails to the team list, queue slots of the DP classes
constexpr int NALLOC = 5 + DP_NQ;
enum { A_OPS = 0, A_REC = 1, A_ACT = 2, A_BAIL = 3, A_BAILT = 4, A_Q0 = 5 };

// the outcome of one hit (align_seed_hit) as the control kernel assembles it from the hit's summary and its DP results
template <class S>
struct HitOut {
  TPath<S> gx, best;
  bool have_best, gx_dead;
  uint32_t best_tx, best_ent;  // transcript and exon-grid entry of the best target
  uint32_t ref_id;
  S hr;
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
  __shared__ unsigned s_stats[16];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  if (threadIdx.x < THM_N_COUNTERS) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 16) s_stats[threadIdx.x] = 0;
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

    // One hit, from its summary (kernels_hit.hip: everything about the hit that does not depend on band, X-drop and best
    // score) and the state in force.
    //   compute  take the DP results from the memo and produce the hit's outcome; returns 1 when they are missing
    //   else     (a hit behind the frontier, asked for speculatively) only collect the problems
    // Problems whose results are missing are collected in pend[].  Order of the records of a hit: the unknown sides of
    // the open transcript targets in yield order (right, left), then those of the genome window -- which are requested
    // only if the genome problem is not dead, known by then.
    auto walk_hit = [&](const bool compute, const uint64_t slot, const int bw, const int xd, HitOut<S>& o) -> int {
      const HitSum* sp = tp.sums + slot;
      const uint8_t flags = sp->flags;
      o.why = 0;
      if (flags & HF_COMPLEX) {
        o.why = sp->why;
        return 2;
      }
      const S hr = (S)sp->hr;
      const int q = sp->q, len = sp->len;
      const HitSide hgr = sp->gr, hgl = sp->gl;
      const int n_open = sp->n_open, n_tgt = sp->n_tgt;
      o.hr = hr;
      o.ref_id = sp->ref_id;
      bool need = false, bad = false;
      o.cells = o.cols = 0;
      o.calls = 2u + 2u * (unsigned)n_tgt;
      const int xr = L - (q + len), xl = q;
      // window bytes: the genome window (:212-215) and the targets' (see HitSum::win_*)
      {
        const unsigned lb = (unsigned)(L + bw);
        unsigned w = (unsigned)len + min(lb, hgr.A) + min(lb, hgl.A);
        w += sp->win_fixed + (unsigned)sp->win_nl * lb + (unsigned)sp->win_nr * (lb + 1u);
        const int nvl = sp->win_nvl, nvr = sp->win_nvr;
        for (int k = 0; k < HIT_MAX_VAR; k++) {
          if (k < nvl) w += min((unsigned)sp->win_var[k], lb);
          else if (k < nvl + nvr) w += min((unsigned)sp->win_var[k], lb + 1u);
        }
        o.win = w;
      }
      // a side as the summary classified it, under the X-drop in force
      auto mk_side = [&](const HitSide& hs, int xlen) -> Side {
        Side s;
        s.score = s.xend = s.yend = s.n = 0;
        s.sp = -1;
        s.rec = -1;
        s.ub = 0;
        s.known = true;
        if (hs.kind == SK_SHORTCUT) {
          if (xd < 1) bad = true;  // (the shortcut needs x_drop >= 1: the wave-per-read kernels take such a read)
          s.score = s.ub = xlen - 2;
          s.xend = s.yend = s.n = xlen;
          s.sp = 0;
        } else if (hs.kind > SK_SHORTCUT) {
          s.known = false;
          s.ub = hs.kind == SK_UNK_DEL ? xlen - 1 : (hs.kind == SK_UNK ? max(xlen - 2, 0) : xlen);
        }
        return s;
      };
      // an extension that needs a DP: its result from the memo, or one more problem to ask for
      auto unknown_side = [&](Side& s, const uint8_t* x0, const uint8_t* y0, int dir, int xlen, unsigned A) {
        const int ylen = (int)min(A, (unsigned)(xlen + bw + 1));
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
          e.cls = (slots <= DPT_SLOTS && (uint32_t)ylen + 1u <= tp.dpt_cols) ? (uint8_t)0 : (uint8_t)((slots + 63) / 64);
          e.pad_ = 0;
        } else {
          pend_over = true;
        }
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
      o.gx.r = mk_side(hgr, xr);
      o.gx.l = mk_side(hgl, xl);
      const bool gx_known = o.gx.l.known && o.gx.r.known;
      const int gx_ub = o.gx.l.ub + len * MATCH_SCORE + o.gx.r.ub;
      // ---- the open targets: each unknown side is the genome window's problem again (same x, same number of y symbols,
      // the same symbols: one DP serves both), or a problem of its own ----
      TPath<S> op[HIT_MAX_OPEN];
      HitTgt ot[HIT_MAX_OPEN];
      bool al_r[HIT_MAX_OPEN], al_l[HIT_MAX_OPEN];
      bool any_alias = false;
      for (int j = 0; j < HIT_MAX_OPEN; j++) {
        al_r[j] = al_l[j] = false;
        if (j >= n_open) continue;
        const HitTgt T = sp->open[j];
        ot[j] = T;
        const int t_q = T.t_q, t_len = T.t_len, t_xr = L - (t_q + t_len), t_xl = t_q;
        const uint8_t* seq = nullptr;
        op[j].r = mk_side(T.r, t_xr);
        op[j].l = mk_side(T.l, t_xl);
        if (!op[j].r.known) {
          const unsigned X = (unsigned)(t_xr + bw + 1), yl = min(T.r.A, X);
          if (!o.gx.r.known && (unsigned)(t_q + t_len) == (unsigned)(q + len) && yl == (unsigned)min(hgr.A, X) && (unsigned)T.r.eq >= yl) {
            al_r[j] = any_alias = true;
          } else {
            seq = ix.tx_seq + ix.exon_grid[T.ent].seq_off;
            unknown_side(op[j].r, rd + t_q + t_len, seq + (T.tr + t_len), 1, t_xr, T.r.A);
          }
        }
        if (!op[j].l.known) {
          const unsigned X = (unsigned)(t_xl + bw + 1), yl = min(T.l.A, X);
          if (!o.gx.l.known && (unsigned)t_q == (unsigned)q && yl == (unsigned)min(hgl.A, X) && (unsigned)T.l.eq >= yl) {
            al_l[j] = any_alias = true;
          } else {
            if (!seq) seq = ix.tx_seq + ix.exon_grid[T.ent].seq_off;
            unknown_side(op[j].l, rd + t_q - 1, seq + (T.tr - 1), -1, t_xl, T.l.A);
          }
        }
      }
      // ---- the best target known in closed form ----
      TPath<S> kn;
      HitTgt kt;
      int known_score = -1;
      const bool has_known = (flags & HF_KNOWN) != 0;
      if (has_known) {
        kt = sp->known;
        kn.r = mk_side(kt.r, L - ((int)kt.t_q + (int)kt.t_len));
        kn.l = mk_side(kt.l, (int)kt.t_q);
        finish_path(kn, (S)kt.tr, (int)kt.t_q, (int)kt.t_len);
        known_score = kn.score;
      }
      // ---- the genome problem: dead when a target known in closed form reaches its upper bound and no target takes
      // its result (it is not computed then and takes no records); else its unknown sides are the hit's last requests ----
      o.gx_dead = !gx_known && !any_alias && has_known && known_score >= gx_ub;
      if (!o.gx_dead) {
        if (!o.gx.r.known) unknown_side(o.gx.r, rd + q + len, ix.text + (hr + len), 1, xr, hgr.A);
        if (!o.gx.l.known) unknown_side(o.gx.l, rd + q - 1, ix.text + (hr - 1), -1, xl, hgl.A);
      }
      if (bad) o.why = 7;
      if (o.why) return 2;
      if (!compute) return 0;
      if (need) return 1;
      finish_path(o.gx, hr, q, len);  // (a dead genome problem: never looked at)
      // ---- the best target: the first of the best in yield order (strictly better wins, :249) ----
      o.have_best = false;
      o.best_tx = o.best_ent = 0;
      uint32_t best_pos = 0;
      if (has_known) {
        o.have_best = true;
        o.best = kn;
        o.best_tx = kt.tx;
        o.best_ent = kt.ent;
        best_pos = kt.pos;
      }
      for (int j = 0; j < HIT_MAX_OPEN; j++) {
        if (j >= n_open) continue;
        if (al_r[j]) op[j].r = o.gx.r;
        if (al_l[j]) op[j].l = o.gx.l;
        finish_path(op[j], (S)ot[j].tr, (int)ot[j].t_q, (int)ot[j].t_len);
        if (!o.have_best || op[j].score > o.best.score || (op[j].score == o.best.score && ot[j].pos < best_pos)) {
          o.have_best = true;
          o.best = op[j];
          o.best_tx = ot[j].tx;
          o.best_ent = ot[j].ent;
          best_pos = ot[j].pos;
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
        uint32_t n_acc = 0;
        const uint32_t n_hits = rec.n_hits;
        for (uint32_t hno = 0; !bail && !sleep && hno < n_hits; hno++) {
          {
            // ================= align_seed_hit (src/aligner.rs:198-314) =================
            const int bw = band_width, xd = x_drop;
            // a round of the memo that was requested from this hit on: its records are for the state in force now
            while (k_round + 1 < (int)memo.n_rounds && memo.first_hit[k_round + 1] == (uint16_t)hno) {
              k_round++;
              ord_r = 0;
              round_ok = true;
            }
            HitOut<S> h;
            const int st = walk_hit(true, cand0 + hno, bw, xd, h);
            if (st == 2) {
              bail = true;
              why = h.why;
              break;
            }
            if (st == 1) {
              // results are missing: this hit is the round's frontier; its problems are in pend[]
              if (memo.n_rounds >= TPR_MAX_ROUNDS || tp.last_round != 0 || pend_over || n_pend == 0) {
                bail = true;
                why = 8;
                break;
              }
              // The first hit of a read usually is accepted and narrows the wide initial band: it is asked for alone.
              // After that the state rarely moves (and a read whose first hit was rejected tends to reject the
              // others too): all the remaining hits, as far as pend[] holds their problems.  A hit behind the frontier that this path cannot take ends the
              // batch; if the replay gets there it asks again, or leaves the read to the wave-per-read kernel.
              uint32_t left = (n_acc > 0 || memo.n_rounds > 0) ? 0xFFFFFFFFu : 1u;
              for (uint32_t h2 = hno + 1; --left && h2 < n_hits; h2++) {
                const int mark = n_pend;
                HitOut<S> hb;
                const int st2 = walk_hit(false, cand0 + h2, bw, xd, hb);
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
            const ExonEntryT<C>* ent = ix.exon_grid;  // (best_ent is the entry's index in the whole grid)
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
              why = 9;
              break;
            }
            TCand k;
            k.n_y = 0;
            k.tx_ystart = k.tx_yend = 0;
            k.tx_ylen = 0;
            for (int m = 0; m < TPR_MAX_MK; m++) k.mk_k[m] = k.ycl[m] = 0;
            k.nops = sel.nops;
            // the path's ops that are not Match: rev(left.ops) ++ Match x len ++ right.ops (:388-394); a DP result lists
            // its own from the end cell back to the seed -- the path's order on the left, the reverse of it on the right
            int n_ed = 0, n_ins = 0;
            {
              bool over = false;
              auto add = [&](int kpath, unsigned kind) {
                if (n_ed < TPR_MAX_ED) k.ed[n_ed] = (uint16_t)((kpath << 2) | (int)kind);
                n_ed++;
                n_ins += kind == (unsigned)OPK_INS;
              };
              for (int t = 0; t < TPR_MAX_ED; t++) k.ed[t] = 0;
              if (sel.l.rec >= 0) {
                const DpRec* d = tp.recs + sel.l.rec;
                const unsigned ne = d->n_edits;
                const uint16_t* e = (const uint16_t*)d;
                if (ne > (unsigned)DP_MAX_EDITS) over = true;
                for (unsigned i = 0; i < (unsigned)DP_MAX_EDITS; i++)
                  if (i < ne && !over) add((int)(e[i] >> 2), e[i] & 3u);
              } else if (sel.l.sp >= 0) {
                add(sel.nl - 1 - sel.l.sp, (unsigned)OPK_SUBST);
              }
              const int r0 = sel.nl + sel.len, nr = sel.nops - r0;
              if (sel.r.rec >= 0) {
                const DpRec* d = tp.recs + sel.r.rec;
                const unsigned ne = d->n_edits;
                const uint16_t* e = (const uint16_t*)d;
                if (ne > (unsigned)DP_MAX_EDITS) over = true;
                for (int i = DP_MAX_EDITS - 1; i >= 0; i--)
                  if ((unsigned)i < ne && !over) add(r0 + (nr - 1 - (int)(e[i] >> 2)), e[i] & 3u);
              } else if (sel.r.sp >= 0) {
                add(r0 + sel.r.sp, (unsigned)OPK_SUBST);
              }
              if (over || n_ed > TPR_MAX_ED) {  // more ops beside Match than a candidate keeps: the wave-per-read kernels take the read
                bail = true;
                why = 10;
                break;
              }
            }
            k.n_ed = (uint8_t)n_ed;
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
                const int n_adv = best.nops - n_ins;
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
                for (;;) {
                  const int bnd = xsum + (int)(cur.end - cur.start);
                  if (e + 1 >= ne || bnd > ye_) break;
                  const int need_adv = bnd - ys_;  // 1-based rank, among the advancing ops, of the op that reaches bnd
                  // the index behind that op: every Ins before it moves it on by one
                  int kstar = need_adv;
                  for (int i = 0; i < TPR_MAX_ED; i++)
                    if (i < n_ed && (k.ed[i] & 3u) == (unsigned)OPK_INS && (int)(k.ed[i] >> 2) < kstar) kstar++;
                  if (need_adv > n_adv) break;  // (cannot happen with a consistent count)
                  if (kstar >= best.nops && !trailing_clip) break;
                  if (n_y >= TPR_MAX_MK) {
                    bail = true;
                    why = 11;
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
            const RefRecT<C> href = ix.ref_recs[h.ref_id];  // Index::idx_to_ref(hit), looked up by the summary kernel
            RefRecT<C> cref = href;
            if (!((C)cy0 >= href.start && (C)cy0 < href.end)) {
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
            k.name_rank = href.name_rank;
            k.type_idx = type_idx;
            k.strand = href.strand != 0 ? 1 : 0;
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
        want[A_Q0 + pend[t].cls]++;
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
                                : a2 == A_ACT ? tp.n_act_out
                                : a2 == A_BAIL ? tp.bail_count
                                : a2 == A_BAILT ? tp.team_count
                                                : &tp.q_cur[a2 - A_Q0];  // (a2 - A_Q0 < DP_NQ)
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
    const bool req_ok = blk_end[A_REC] <= tp.rec_cap && blk_end[A_REC] < 0x7FFFFF00ull &&
                        blk_end[A_Q0] <= tp.q_stride && blk_end[A_Q0 + 1] <= tp.q_stride && blk_end[A_Q0 + 2] <= tp.q_stride &&
                        blk_end[A_Q0 + 3] <= tp.q_stride && blk_end[A_Q0 + 4] <= tp.q_stride;
    if (sleep && !req_ok) {
      sleep = false;
      const unsigned long long slot = atomicAdd(tp.bail_count, 1ull);  // (a late bail has no slot of this pass's allocation)
      tp.bail[slot] = idx;
      why = 12;
      tp.act_out[mine_at[A_ACT]] = 0xFFFFFFFFu;  // its slot in the next round's list stays: marked empty
      {  // ... and so do its slots in the DP queues (as far as they lie inside the lists)
        unsigned long long qs4[DP_NQ] = {mine_at[A_Q0], mine_at[A_Q0 + 1], mine_at[A_Q0 + 2], mine_at[A_Q0 + 3], mine_at[A_Q0 + 4]};
        for (int t = 0; t < n_pend; t++) {
          const int c2 = pend[t].cls;
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
      unsigned long long qs4[DP_NQ] = {mine_at[A_Q0], mine_at[A_Q0 + 1], mine_at[A_Q0 + 2], mine_at[A_Q0 + 3], mine_at[A_Q0 + 4]};
      const unsigned long long w_rec = mine_at[A_REC];
      const unsigned w_n = (unsigned)n_pend;
      for (int t = 0; t < n_pend; t++) {
        const Pend& e = pend[t];
        DpRec d;
        d.x0 = e.x0;
        d.y0 = e.y0;
        d.pad0_ = 0;
        d.xlen = e.xlen;
        d.ylen = e.ylen;
        d.bw = (uint16_t)f_bw;
        d.xd = (uint16_t)min(f_xd, 65535);
        d.dir = e.dir;
        d.cls = e.cls;
        d.n_edits = 0;
        d.read = (uint32_t)idx;
        d.score = 0;
        d.xend = d.yend = d.nops = 0;
        d.done = 0;
        d.cells = d.cols = 0;
        d.pad2_ = 0;
        const uint32_t ri = (uint32_t)(w_rec + (unsigned)t);
        tp.recs[ri] = d;
        tp.q_list[(size_t)e.cls * tp.q_stride + qs4[e.cls]] = ri;
        qs4[e.cls]++;
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
        emit_stream(p.cand_ops + my_off, nb, a.ed, (int)a.n_ed, lead, trail, a.rev != 0, (int)a.n_y, a.mk_k, a.ycl);
        const unsigned long long goff = my_off;
        my_off += ((unsigned)nb + 3u) & ~3u;
        unsigned long long toff = 0;
        if (exonic) {
          emit_stream(p.cand_ops + my_off, tnb, a.ed, (int)a.n_ed, lead, trail, false, 0, a.mk_k, a.ycl);
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
        for (int w = 1; w < 16; w++) {
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
  if (tp.stats && threadIdx.x < 16 && s_stats[threadIdx.x]) atomicAdd(&tp.stats[threadIdx.x], (unsigned long long)s_stats[threadIdx.x]);
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
  const unsigned long long q0 = min(p.q_done[CPL], (unsigned long long)p.q_stride), q1 = min(p.q_cur[CPL], (unsigned long long)p.q_stride);
  if (q0 >= q1) return;
  const uint32_t* list = p.q_list + (size_t)CPL * p.q_stride;
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
    // The result's ops are Match but for a few: those few go into the record, (index << 2 | kind), in the traceback's
    // order; the control kernel rebuilds the op list from them.
    uint16_t* eds = (uint16_t*)(opsb + ops_cap - 16);  // (the ops end at least 128 bytes before)
    int n_ed = 0;
#pragma unroll 1
    for (int t0 = 0; t0 < nops; t0 += 64) {
      const int t = t0 + lane;
      const int v = t < nops ? (int)opsb[t] : 0;
      const unsigned long long m = __ballot(v != 0);
      if (v) {
        const int at = n_ed + __popcll(m & ((1ull << lane) - 1ull));
        if (at < DP_MAX_EDITS) eds[at] = (uint16_t)((t << 2) | v);
      }
      n_ed += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    DpRec* d = p.recs + ri;
    if (lane < DP_MAX_EDITS && lane < n_ed && n_ed <= DP_MAX_EDITS) ((uint16_t*)d)[lane] = eds[lane];
    if (lane == 0) {
      d->score = r.score;
      d->xend = (uint16_t)r.xend;
      d->yend = (uint16_t)r.yend;
      d->nops = (uint16_t)nops;
      d->n_edits = n_ed <= DP_MAX_EDITS ? (uint16_t)n_ed : (uint16_t)0xFFFF;
      d->cells = r.cells;  // counted by the control kernel when (and if) the read is finished with this result
      d->cols = r.cols;
      d->done = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Thread-per-problem DP kernel (class 0): SwgExtend::extend + trace (reference src/swg.rs:31-207)
// statement by statement, one problem per THREAD, for the problems whose columns never hold more than DPT_SLOTS
// cells: min(2 bw + 1, |x| + 1) <= 16.  That is every extension after a read's first accepted hit (the band is
// L + range - score then: a few slots) and the short ends of reads -- the problems a wavefront-per-problem kernel does
// worst at (a dozen live lanes of 64, ~50 vector and scalar instructions per column whatever the band).  Here the band
// of a column is a thread's registers, a cell costs a dozen instructions, and a wavefront works on 64 problems.
// The trace (2 bits per slot, one 32-bit word per column) is in LDS, laid out [column][thread]: conflict-free, and the
// traceback -- a chain of dependent look-ups -- never leaves the CU.  x and y arrive eight symbols per load.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int DPT_THREADS = 128;
// symbols t = 8 c .. 8 c + 7 of a sequence walked from p0 in direction dir (+1 / -1), symbol t in byte t - 8 c; only
// symbols below len are touched
__device__ __forceinline__ unsigned long long fetch8(const uint8_t* p0, int dir, int c, int len) {
  // (one load for a chunk that holds a symbol below len: the reads, the text and the transcript sequences carry 16 bytes
  // of padding in front and more behind, and the symbols at and beyond len are never looked at)
  const int t0 = 8 * c;
  if (t0 >= len) return 0ull;  // (a chunk fetched ahead of the last one)
  unsigned long long v;
  if (dir > 0) {
    __builtin_memcpy(&v, p0 + t0, 8);
  } else {
    __builtin_memcpy(&v, p0 - t0 - 7, 8);
    v = __builtin_bswap64(v);
  }
  return v;
}
}  // namespace

__global__ __launch_bounds__(DPT_THREADS) void extend_dpt_kernel(DpParams p) {
  constexpr int W = DPT_SLOTS;
  constexpr int ge = GAP_EXTEND, go = GAP_OPEN;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_t[];
  const unsigned tid = blockIdx.x * (unsigned)DPT_THREADS + threadIdx.x, n_threads = gridDim.x * (unsigned)DPT_THREADS;
  uint32_t* const trace = (uint32_t*)smem_t + threadIdx.x;  // column j at trace[j * DPT_THREADS]
  const unsigned long long q0 = min(p.q_done[0], (unsigned long long)p.q_stride), q1 = min(p.q_cur[0], (unsigned long long)p.q_stride);
  int fault = 0;
  for (unsigned long long qi = q0 + tid; qi < q1; qi += n_threads) {
    const uint32_t ri = p.q_list[qi];
    if (ri == 0xFFFFFFFFu) continue;
    DpRec* const d = p.recs + ri;
    const uint8_t* const x0 = d->x0;
    const uint8_t* const y0 = d->y0;
    const int xlen = d->xlen, ylen = d->ylen, bw = d->bw, xd = d->xd, dir = d->dir;
    const int w = 2 * bw + 1;
    if (min(w, xlen + 1) > W || (uint32_t)ylen + 1u > p.tcols || xlen == 0 || ylen == 0) {
      fault |= 8;
      continue;
    }
    // :62-71 leftmost column (its trace is all Ins: not stored)
    int D[W + 1], C[W + 1];
#pragma unroll
    for (int b = 0; b <= W; b++) {
      D[b] = b == 0 ? 0 : b * ge + go;
      C[b] = b == 0 ? 0 : MIN_SCORE;
    }
    D[W] = C[W] = MIN_SCORE;  // (read by the last slot's look at its lower neighbour; never holds a cell)
    // the x symbols of the band's rows: xw[b] = x[top + b - 1], top = 0 in phase 1 (slot b = row b)
    int xw[W];
    {
      const unsigned long long c0 = fetch8(x0, dir, 0, xlen), c1 = fetch8(x0, dir, 1, xlen);
#pragma unroll
      for (int b = 0; b < W; b++) {
        const int t = b - 1;  // x[t]
        const int v = t < 8 ? (int)((c0 >> (8 * max(t, 0))) & 0xffull) : (int)((c1 >> (8 * (t - 8))) & 0xffull);
        xw[b] = (b >= 1 && t < xlen) ? v : 0x100;  // 0x100: equals no symbol
      }
    }
    int max_score = 0, max_i = 0, max_j = 0;
    unsigned n_cells = 0, n_cols = 0;
    // :75-113 band anchored at row 0
    const int p1_end = min(bw, ylen);
    const int rows = min(w, xlen + 1);
    // y[j - 1] is the symbol of column j; eight of them per load, the next eight fetched while these are used
    unsigned long long ych = fetch8(y0, dir, 0, ylen), ych_next = fetch8(y0, dir, 1, ylen);
    auto y_at = [&](int t) -> int {  // symbols are asked for in ascending order
      if ((t & 7) == 0 && t > 0) {
        ych = ych_next;
        ych_next = fetch8(y0, dir, (t >> 3) + 1, ylen);
      }
      return (int)((ych >> (8 * (t & 7))) & 0xffull);
    };
    int y_t = 0;  // next symbol to take
    for (int j = 1; j <= p1_end; j++) {
      const int yc = y_at(y_t++);
      int band_max = MIN_SCORE, prev_D = MIN_SCORE, Rrun = MIN_SCORE, Dleft = MIN_SCORE;
      uint32_t tw = 0;
#pragma unroll
      for (int b = 0; b < W; b++) {
        if (b < rows) {
          const int Cn = max(C[b] + ge, D[b] + ge + go);
          const int Rn = b == 0 ? MIN_SCORE : max(Rrun + ge, Dleft + ge + go);
          const bool eq = b > 0 && xw[b] == yc;
          const int dg = b == 0 ? MIN_SCORE : prev_D + (eq ? MATCH_SCORE : MISMATCH_SCORE);
          prev_D = D[b];
          const int sc = max(dg, max(Cn, Rn));
          // triple_max (:226-240): diag (Match if the bases are equal, else Subst) > Del > Ins
          const uint32_t op = sc == dg ? (eq ? (uint32_t)OPK_MATCH : (uint32_t)OPK_SUBST) : (sc == Cn ? (uint32_t)OPK_DEL : (uint32_t)OPK_INS);
          D[b] = sc;
          C[b] = Cn;
          Rrun = Rn;
          Dleft = sc;
          tw |= op << (2 * b);
          if (sc > max_score) {
            max_score = sc;
            max_i = b;
            max_j = j;
          }
          band_max = max(band_max, sc);
        }
      }
      trace[(size_t)j * DPT_THREADS] = tw;
      n_cells += (unsigned)rows;
      n_cols++;
      if (band_max < max_score - xd) break;  // leaves ONLY this loop (:110-116); cannot fire with x_drop >= band_width
    }
    // :116-154 band slides down one row per column
    if (y_t != bw && bw + 1 <= ylen) {  // phase 1 left early: cannot happen with x_drop >= band_width
      fault |= 4;
      continue;
    }
    // x[t] for the band's new last row: eight per load as well (t = W - 1, W, ...)
    int x_t = W - 1;
    unsigned long long xch = fetch8(x0, dir, x_t >> 3, xlen), xch_next = fetch8(x0, dir, (x_t >> 3) + 1, xlen);
    for (int j = bw + 1; j <= ylen; j++) {
      const int top = j - bw;
      n_cols++;
      if (top > xlen) break;  // empty row range: band_max = MIN -> X-drop
      const int yc = y_at(y_t++);
      // the band moved down a row: xw[b] = x[top + b - 1]; the new last slot holds x[top + W - 2]
#pragma unroll
      for (int b = 0; b < W - 1; b++) xw[b] = xw[b + 1];
      {
        if ((x_t & 7) == 0 && x_t > W - 1) {
          xch = xch_next;
          xch_next = fetch8(x0, dir, (x_t >> 3) + 1, xlen);
        }
        xw[W - 1] = x_t < xlen ? (int)((xch >> (8 * (x_t & 7))) & 0xffull) : 0x100;
        x_t++;
      }
      const int nvalid = min(w, xlen + 1 - top);
      int band_max = MIN_SCORE, Rrun = MIN_SCORE, Dleft = MIN_SCORE;
      uint32_t tw = 0;
#pragma unroll
      for (int b = 0; b < W; b++) {
        if (b < nvalid) {
          const int Cn = (b >= w - 1) ? MIN_SCORE : max(C[b + 1] + ge, D[b + 1] + ge + go);
          const int Rn = b == 0 ? MIN_SCORE : max(Rrun + ge, Dleft + ge + go);
          const bool eq = xw[b] == yc;
          const int dg = D[b] + (eq ? MATCH_SCORE : MISMATCH_SCORE);
          const int sc = max(dg, max(Cn, Rn));
          const uint32_t op = sc == dg ? (eq ? (uint32_t)OPK_MATCH : (uint32_t)OPK_SUBST) : (sc == Cn ? (uint32_t)OPK_DEL : (uint32_t)OPK_INS);
          D[b] = sc;
          C[b] = Cn;
          Rrun = Rn;
          Dleft = sc;
          tw |= op << (2 * b);
          if (sc > max_score) {
            max_score = sc;
            max_i = top + b;
            max_j = j;
          }
          band_max = max(band_max, sc);
        }
      }
      trace[(size_t)j * DPT_THREADS] = tw;
      n_cells += (unsigned)nvalid;
      if (band_max < max_score - xd) break;
    }
    // :170-207 traceback, the ops that are not Match as (index from the end cell << 2 | kind)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    unsigned long long ed_lo = 0, ed_hi = 0;
    int n_ed = 0, n = 0;
    {
      int i = max_i, j = max_j, tj = -1;
      uint32_t tw = 0;
      int guard = xlen + ylen + 4;
      while ((i > 0 || j > 0) && guard-- > 0) {
        uint32_t op;
        if (j == 0) {
          op = (uint32_t)OPK_INS;  // column 0 is all Ins (:65,:70)
        } else {
          if (tj != j) {
            tw = trace[(size_t)j * DPT_THREADS];
            tj = j;
          }
          const int b = i - max(j - bw, 0);
          if (b < 0 || b >= W) {
            fault |= 2;
            break;
          }
          op = (tw >> (2 * b)) & 3u;
        }
        if (op != (uint32_t)OPK_MATCH) {
          const unsigned long long v = (unsigned long long)(((unsigned)n << 2) | op) & 0xffffull;
          if (n_ed < 4)
            ed_lo |= v << (16 * n_ed);
          else if (n_ed < 8)
            ed_hi |= v << (16 * (n_ed - 4));
          n_ed++;
        }
        n++;
        if (op <= (uint32_t)OPK_SUBST) {
          if (i == 0 || j == 0) {
            fault |= 2;
            break;
          }
          i--;
          j--;
        } else if (op == (uint32_t)OPK_INS) {
          if (i == 0) {
            fault |= 2;
            break;
          }
          i--;
        } else {
          j--;
        }
      }
      if (guard <= 0) fault |= 2;
    }
    unsigned long long* d64 = (unsigned long long*)d;
    d64[0] = ed_lo;
    d64[1] = ed_hi;
    d->score = max_score;
    d->xend = (uint16_t)max_i;
    d->yend = (uint16_t)max_j;
    d->nops = (uint16_t)n;
    d->n_edits = n_ed <= DP_MAX_EDITS ? (uint16_t)n_ed : (uint16_t)0xFFFF;
    d->cells = n_cells;
    d->cols = n_cols;
    d->done = 1;
  }
  if (fault) atomicOr(p.fault, 2 | (fault << 4));
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

size_t extend_dpt_lds_bytes(uint32_t tcols) { return (size_t)dev::DPT_THREADS * tcols * 4; }
hipError_t launch_extend_dpt(const DpParams& p, int n_blocks, hipStream_t s) {
  if (n_blocks <= 0) return hipSuccess;
  const size_t lds = extend_dpt_lds_bytes(p.tcols);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)dev::extend_dpt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(dev::extend_dpt_kernel, dim3(n_blocks), dim3(dev::DPT_THREADS), lds, s, p);
  return hipGetLastError();
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
