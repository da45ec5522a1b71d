// swg_device.h -- banded Smith-Waterman-Gotoh extension, one problem per
// wavefront, for gfx950 (wave64).
//
// Semantics: SwgExtend::extend + trace, reference src/swg.rs:31-240 (SURVEY.md
// Appendix A), bit-exact on the contract x_drop >= band_width.
//
// Mapping to the hardware
//   * the band of one DP column (w = 2*bw+1 slots, reference arrays D/C/R
//     indexed by band_idx) lives on the lanes: slot b -> lane b / CPL, register
//     b % CPL (CPL = cells per lane, compile time; w <= 64*CPL);
//   * columns are walked in order, so the X-drop test (reference :110,:151) is
//     evaluated exactly where the reference evaluates it;
//   * the horizontal state C and the diagonal input come from the previous
//     column: own registers in phase 1 (band anchored at row 0, :75-113), one
//     DPP wave shift in phase 2 (band slides down, :116-154);
//   * the vertical gap state R is a serial chain in the reference
//     (R[i] = max(R[i-1]+ge, D[i-1]+ge+go)).  With gap_open <= 0 it equals
//     max_{k<i}(D'[k] + go + (i-k)*ge), D' = max(diag, C): a wave-wide exclusive
//     prefix-max of (D'[k] - k*ge), done with 7 DPP steps (row_shr 1/2/4/8,
//     row_bcast 15/31, wave_shr 1);
//   * trace directions are 2 bits per cell: two 64-bit ballots per register
//     slice per column, stored by one lane (16*CPL bytes per column; LDS for the
//     one-cell-per-lane code, a per-wave scratch in global memory for wider bands);
//   * no per-lane argmax: every cell derives from the previous column by moves worth
//     at most +1, so the running maximum (an SGPR) rises by exactly 1 iff some lane
//     beats it; the first cell attaining the final maximum is the lowest improving
//     row of the last improving column (one ballot mask kept per improving column);
//   * traceback keeps the ballot words of 64 columns in registers (lane t <-> column)
//     and resolves a whole diagonal run per step;
//   * bands of any width (more than 256 slots: reads of several hundred bases with a
//     low score threshold) run through swg_extend_tiled below: the band in tiles of
//     64 slots, the column state in a wave-private array in global memory.
//
// Scoring is the aligner's fixed Scoring::from_scores(-1,-1,1,-1)
// (reference src/aligner.rs:140).
#ifndef THERMITE_SWG_DEVICE_H
#define THERMITE_SWG_DEVICE_H

#include <hip/hip_runtime.h>

#include "thermite_internal.h"

namespace thm {
namespace dev {

constexpr int NEG = MIN_SCORE;  // scan identity; never wraps when a few hundred -1s are added

enum : int { OPK_MATCH = 0, OPK_SUBST = 1, OPK_DEL = 2, OPK_INS = 3 };

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// lane l <- lane l-1 (lane 0 <- fill)
__device__ __forceinline__ int wave_shr1(int v, int fill) { return dpp_mov<0x138, 0xf>(fill, v); }
// lane l <- lane l+1 (lane 63 <- fill)
__device__ __forceinline__ int wave_shl1(int v, int fill) { return dpp_mov<0x130, 0xf>(fill, v); }

// inclusive prefix max over the 64 lanes
__device__ __forceinline__ int wave_incl_max_scan(int v) {
  v = max(v, dpp_mov<0x111, 0xf>(NEG, v));  // row_shr:1
  v = max(v, dpp_mov<0x112, 0xf>(NEG, v));  // row_shr:2
  v = max(v, dpp_mov<0x114, 0xf>(NEG, v));  // row_shr:4
  v = max(v, dpp_mov<0x118, 0xf>(NEG, v));  // row_shr:8
  v = max(v, dpp_mov<0x142, 0xa>(NEG, v));  // row_bcast:15 -> rows 1,3
  v = max(v, dpp_mov<0x143, 0xc>(NEG, v));  // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ int wave_excl_max_scan(int v) { return wave_incl_max_scan(wave_shr1(v, NEG)); }
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_incl_max_scan(v), 63); }
__device__ __forceinline__ int wave_min(int v) { return -wave_max(-v); }
__device__ __forceinline__ int bcast_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
// one v_cmp writing an SGPR pair (no 0/1 round trip through a VGPR)
__device__ __forceinline__ unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Pins a wave-uniform value to scalar registers and hides its origin from the optimiser, which
// otherwise folds `vote(a) & mask` back into a per-lane AND (a 0/1 round trip through a VGPR).
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  asm("" : "+s"(lo), "+s"(hi));
  return ((unsigned long long)hi << 32) | lo;
}
// (v > t) & mask and (v >= t) & mask as lane masks: one compare into an SGPR pair and one scalar AND
// (written out because the optimiser turns `vote(v > t) & mask` into a per-lane AND plus a re-vote)
__device__ __forceinline__ unsigned long long mask_gt(int v, int t, unsigned long long mask) {
  unsigned long long m;
  asm("v_cmp_gt_i32_e64 %0, %1, %2\n\ts_and_b64 %0, %0, %3" : "=&s"(m) : "v"(v), "s"(t), "s"(mask) : "scc");
  return m;
}
__device__ __forceinline__ unsigned long long mask_ge(int v, int t, unsigned long long mask) {
  unsigned long long m;
  asm("v_cmp_ge_i32_e64 %0, %1, %2\n\ts_and_b64 %0, %0, %3" : "=&s"(m) : "v"(v), "s"(t), "s"(mask) : "scc");
  return m;
}
__device__ __forceinline__ int sgpr32(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm("" : "+s"(v));
  return v;
}

// Load through the constant address space: with a wave-uniform address this
// becomes an s_load (scalar cache, result in SGPRs, no EXEC games in the control
// flow that depends on it).  Only for data no kernel instance writes while it
// runs (the index, the read batch, the outputs of earlier kernels).
template <class T>
__device__ __forceinline__ T uload(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) T* cptr;
  // readfirstlane pins the address as wave-uniform for the compiler as well
  const uintptr_t a = (uintptr_t)p;
  const uintptr_t u = ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                      (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  return *(cptr)u;
#else
  return *p;
#endif
}

struct SwgResult {
  int score, xend, yend;
  unsigned cells, cols;  // cells / columns actually computed (the early exit below makes this <= the reference's count)
  // What the result depends on: y[0..jmax) and, when `broke` is false (the loop ran
  // out of columns), on ylen itself.  A second problem with the same x, band and
  // X-drop whose y agrees on [0, jmax) -- and has ylen >= jmax if broke, the same
  // ylen otherwise -- has the same result (used to skip the transcript extension
  // when it would repeat the genome extension).
  int jmax;
  bool broke;
};

// ---- fused DPP steps for the per-column scan (the hot loop) ----
// `v_max_i32_dpp v, v, v <ctrl>`: lanes whose DPP source is out of range (or masked
// out) are simply not written, so they keep their own v -- exactly max(v, identity).
// One instruction per step instead of mov-identity + mov_dpp + max.  The two wait
// states a DPP read needs after a VALU write are in the asm (hipcc pads nothing
// inside an asm statement).
// `r` is carried by the caller across columns: lane 0 (which no step writes) keeps the identity it
// was initialised with, all other lanes are overwritten by the first step.
__device__ __forceinline__ int wave_excl_max_scan_fast(int v, int& r) {
#if defined(THM_EXP_NOSCAN)
  return wave_shr1(v, NEG);
#elif defined(THM_EXP_NONOP)
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(r)
      : "v"(v));
  return r;
#endif
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(r)
      : "v"(v));
  return r;
}

static_assert(MATCH_SCORE == 1 && MISMATCH_SCORE <= 0 && GAP_OPEN <= 0 && GAP_EXTEND <= 0,
              "the running-max and early-exit logic below relies on +1 per column at most");

// One extension.  x[i] = xs[i*dx], y[j] = ys[j*dy] (dx, dy = +1 or -1: a left
// extension walks the read and the window backwards, reference
// src/aligner.rs:364-375) are wave-private LDS bytes; trace is wave-private LDS
// with room for (ylen+1)*CPL*2 u64.  Every lane returns the same SwgResult.
// Contract: 2*bw+1 <= 64*CPL, xd >= bw.
//
// Per column there is no wave reduction besides the scan for R:
//   * every cell of column j derives from column j-1 by moves worth at most +1,
//     so the running maximum rises by exactly 1 iff some lane beats it (a ballot);
//   * X-drop (reference :151: band_max < max_score - x_drop) is "no lane has
//     D >= max_score - x_drop" (a ballot);
//   * early exit: if no cell of this column can still exceed max_score even by
//     matching all of its remaining x (D + (|x| - i) <= max_score for every band
//     cell), no later cell can, because every path to a later column crosses this
//     one.  The reference would keep computing until its X-drop fires, but its
//     result (strict `>` on max_score) cannot change any more.  Same score, same
//     end cell, same trace cells on the path.
template <int CPL>
__device__ SwgResult swg_extend_wave(const uint8_t* xs, int dx, int xlen, const uint8_t* ys, int dy, int ylen, int bw,
                                     int xd, unsigned long long* trace) {
  SwgResult res;
  res.score = 0;
  res.xend = 0;
  res.yend = 0;
  res.cells = 0;
  res.cols = 0;
  res.jmax = 0;
  res.broke = (xlen == 0);  // an empty x gives the empty result for any y; an empty y is "ran out of columns"
  if (xlen == 0 || ylen == 0) return res;  // reference :39-55

  const int lane = lane_id();
  const int w = 2 * bw + 1;
  constexpr int ge = GAP_EXTEND, go = GAP_OPEN;

  int Dv[CPL], Cv[CPL];
  // reference :62-71 leftmost column
#pragma unroll
  for (int c = 0; c < CPL; c++) {
    int b = lane * CPL + c;
    Dv[c] = (b == 0) ? 0 : b * ge + go;
    Cv[c] = (b == 0) ? 0 : MIN_SCORE;
  }
  // Running maximum (reference max_score, starts at 0) and where it was first reached.
  // Every rise is by exactly +1, so the first cell attaining the final maximum lies in the
  // last column that raised it, at the lowest improving row of that column.
  int run_max = 0;
  int best_j = 0;
  unsigned long long best_mask[CPL];
#pragma unroll
  for (int c = 0; c < CPL; c++) best_mask[c] = 0;
  bool finished = false;
  unsigned long long* tr = trace + (size_t)CPL * 2;  // column 1
  int scan_acc = NEG;       // see wave_excl_max_scan_fast
  int d_shift = MIN_SCORE;  // phase 1: D of the slot above, shifted in from the neighbouring lane; lane 0 (row 0 has no
                            // diagonal, reference :84) keeps MIN_SCORE because the shift never writes it
  // The counters (cells, columns, last column with cells) are derived after the walk from the
  // last column visited, not maintained per column.
  int last_j = 0;          // last column that computed cells
  bool empty_col = false;  // phase 2 ended on a column whose row range was empty

  // ---------------- phase 1: band rows 0..w-1 (reference :75-113) ----------------
  // Per-lane constants: slot index b, R offset go + b*ge, and -b*ge for the scan key.
  const int p1_end = min(bw, ylen);
  const int rows1 = min(w, xlen + 1);
  {
    int xc[CPL];
    bool valid[CPL];
    unsigned long long vmask[CPL];  // lanes whose slot holds a cell: masks are combined on the scalar unit
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int b = lane * CPL + c;
      // slot 0 (row 0) has no diagonal and slots past |x| hold no cell: their x byte is a don't-care
      xc[c] = (int)xs[min(max(b - 1, 0), xlen - 1) * dx];
      valid[c] = b < rows1;
      vmask[c] = vote(valid[c]);
    }
    // the y character of the next column is fetched one column ahead, so that the LDS
    // latency is off the column-to-column critical path
    const uint8_t* yp = ys;
    int yc_next = (int)*yp;
    for (int j = 1; j <= p1_end; j++) {
      const int yc = yc_next;
      yp += dy;
      yc_next = (int)*yp;  // one past the last column at most: still inside the staged window / LDS
      int d[CPL], Cn[CPL], key[CPL];
      unsigned long long meq[CPL];
      d_shift = wave_shr1(Dv[CPL - 1], d_shift);  // D[b-1] of the previous column for register 0
      const int d_in = d_shift;
      int lane_tot = NEG;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int b = lane * CPL + c;
        Cn[c] = max(Cv[c], Dv[c] + go) + ge;
        const int dprev = (c == 0) ? d_in : Dv[c - 1];
        const bool eq = (uint8_t)xc[c] == (uint8_t)yc;
        meq[c] = vote(eq);
        d[c] = dprev + (eq ? MATCH_SCORE : MISMATCH_SCORE);  // slot 0: MIN_SCORE +- 1, never the maximum
        const int dp = max(d[c], Cn[c]);
        key[c] = dp - b * ge;  // slots without a cell sit above every slot with one: the exclusive scan never feeds them down
        lane_tot = (c == 0) ? key[c] : max(lane_tot, key[c]);
      }
      int run = wave_excl_max_scan_fast(lane_tot, scan_acc);
      unsigned long long m_imp = 0, m_alive = 0, m_c[CPL];
      const int alive_floor = run_max - xlen;  // D + (xlen - i) > run_max  <=>  D - i > run_max - xlen
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int b = lane * CPL + c;
        const int R = run + (go + b * ge);
        run = max(run, key[c]);
        const int Dn = max(max(d[c], Cn[c]), R);
        // direction bits (Match 0, Subst 1, Del 2, Ins 3; priority diag > Del > Ins, reference :226-240)
        const unsigned long long hi = vote(Dn != d[c]);
        const unsigned long long lo = (~hi & ~meq[c]) | (hi & vote(Dn != Cn[c]));
#ifndef THM_EXP_NOTRACE
        if (lane == 0) {
          tr[c * 2 + 0] = lo;
          tr[c * 2 + 1] = hi;
        }
#else
        if (lo == 0x123456789ull && hi == 0x3ull && lane == 0) tr[0] = lo;
#endif
        m_c[c] = mask_gt(Dn, run_max, vmask[c]);
        m_imp |= m_c[c];
        m_alive |= mask_gt(Dn - b, alive_floor, vmask[c]);
        Dv[c] = Dn;  // slots without a cell hold don't-care values: a cell only ever reads slots that held cells
        Cv[c] = Cn[c];
      }
      tr += CPL * 2;
      last_j = j;
      const bool imp = m_imp != 0ull;
      if (imp) {
        best_j = j;
#pragma unroll
        for (int c = 0; c < CPL; c++) best_mask[c] = m_c[c];
      }
      run_max += imp ? MATCH_SCORE : 0;  // wave-uniform: the compares above take it as a scalar operand
      // reference :110: with x_drop >= band_width the X-drop test cannot fire in
      // phase 1; the early exit is ours (see above).
      // (An improving lane is alive by construction.)
      if (m_alive == 0ull) {
        finished = true;
        res.broke = true;
        break;
      }
    }
  }

  // ---------------- phase 2: band slides down (reference :116-154) ----------------
  if (!finished && bw + 1 <= ylen) {
    // x / y characters of the next column are fetched one column ahead (see phase 1);
    // per-lane pointers advance by one character per column.  Reads past the end of x
    // land in other LDS bytes (or return 0 out of range) and are masked by `valid`.
    const uint8_t* yp = ys + bw * dy;
    int yc_next = (int)*yp;
    const uint8_t* xp[CPL];
    int xv_next[CPL];
    bool last_slot[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int b = lane * CPL + c;
      xp[c] = xs + b * dx;  // x[i-1] with i = top + b, top = 1
      xv_next[c] = (int)*xp[c];
      last_slot[c] = b >= w - 1;
    }
    for (int j = bw + 1; j <= ylen; j++) {
      const int top = j - bw;
      if (top > xlen) {  // empty row range: band_max = MIN -> X-drop (reference :117-153)
        res.broke = true;
        empty_col = true;
        break;
      }
      last_j = j;
      const int nvalid = min(w, xlen + 1 - top);  // slots b < nvalid hold a cell
      const int yc = yc_next;
      yp += dy;
      yc_next = (int)*yp;
      int d[CPL], Cn[CPL], key[CPL], xc[CPL], base[CPL];
      bool valid[CPL];
      unsigned long long meq[CPL], vmask[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c++) base[c] = max(Cv[c], Dv[c] + go) + ge;
      const int c_in = wave_shl1(base[0], MIN_SCORE);  // slot b+1 of the previous column for the last register
      int lane_tot = NEG;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int b = lane * CPL + c;
        valid[c] = b < nvalid;
        vmask[c] = vote(valid[c]);
        xc[c] = xv_next[c];  // beyond |x| the byte is arbitrary: it only feeds slots without a cell
        xp[c] += dx;
        xv_next[c] = (int)*xp[c];
        const int cnext = (c == CPL - 1) ? c_in : base[c + 1];
        Cn[c] = last_slot[c] ? MIN_SCORE : cnext;
        const bool eq = (uint8_t)xc[c] == (uint8_t)yc;
        meq[c] = vote(eq);
        d[c] = Dv[c] + (eq ? MATCH_SCORE : MISMATCH_SCORE);
        const int dp = max(d[c], Cn[c]);
        key[c] = dp - b * ge;  // slots without a cell sit above every slot with one: the exclusive scan never feeds them down
        lane_tot = (c == 0) ? key[c] : max(lane_tot, key[c]);
      }
      int run = wave_excl_max_scan_fast(lane_tot, scan_acc);
      unsigned long long m_imp = 0, m_alive = 0, m_x = 0, m_c[CPL];
      const int xfloor = run_max - xd;               // X-drop survivors: D >= max_score - x_drop
      const int alive_floor = run_max - xlen + top;  // D + (xlen - i) > run_max  <=>  D - b > run_max - xlen + top
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int b = lane * CPL + c;
        const int R = run + (go + b * ge);
        run = max(run, key[c]);
        const int Dn = max(max(d[c], Cn[c]), R);
        const unsigned long long hi = vote(Dn != d[c]);
        const unsigned long long lo = (~hi & ~meq[c]) | (hi & vote(Dn != Cn[c]));
#ifndef THM_EXP_NOTRACE
        if (lane == 0) {
          tr[c * 2 + 0] = lo;
          tr[c * 2 + 1] = hi;
        }
#else
        if (lo == 0x123456789ull && hi == 0x3ull && lane == 0) tr[0] = lo;
#endif
        m_c[c] = mask_gt(Dn, run_max, vmask[c]);
        m_imp |= m_c[c];
        m_x |= mask_ge(Dn, xfloor, vmask[c]);
        m_alive |= mask_gt(Dn - b, alive_floor, vmask[c]);
        Dv[c] = Dn;
        Cv[c] = Cn[c];
      }
      tr += CPL * 2;
      const bool imp = m_imp != 0ull;
      if (imp) {
        best_j = j;
#pragma unroll
        for (int c = 0; c < CPL; c++) best_mask[c] = m_c[c];
      }
      run_max += imp ? MATCH_SCORE : 0;  // wave-uniform: the compares above take it as a scalar operand
      // stop: reference :151 (band_max < max_score - x_drop) or our early exit; an improving
      // column equals the new maximum, so neither applies to it
      if (!imp && (m_x == 0ull || m_alive == 0ull)) {
        res.broke = true;
        break;
      }
    }
  }

  // ---------------- counters ----------------
  {
    res.jmax = last_j;
    res.cols = (unsigned)last_j + (empty_col ? 1u : 0u);
    const int n1 = min(last_j, p1_end);  // phase-1 columns: rows1 cells each
    unsigned cells = (unsigned)n1 * (unsigned)rows1;
    if (last_j > p1_end) {
      // phase-2 column with top = j - bw holds min(w, |x| + 1 - top) cells, top = 1 .. t1
      const int t1 = last_j - bw;
      const int tfull = min(t1, xlen + 1 - w);  // tops up to here hold w cells
      if (tfull >= 1) cells += (unsigned)tfull * (unsigned)w;
      const int ta = max(tfull, 0) + 1;  // first top of the shrinking part
      if (t1 >= ta) {
        const int hi_cells = xlen + 1 - ta, lo_cells = xlen + 1 - t1;
        cells += (unsigned)((hi_cells + lo_cells) * (t1 - ta + 1) / 2);
      }
    }
    res.cells = cells;
  }

  // ---------------- first (j, i) attaining the maximum ----------------
  if (run_max > 0) {
    int bmin = 64 * CPL;
#pragma unroll
    for (int c = 0; c < CPL; c++)
      if (best_mask[c]) bmin = min(bmin, (int)__builtin_ctzll(best_mask[c]) * CPL + c);
    res.score = run_max;
    res.xend = max(best_j - bw, 0) + bmin;  // top of the band in column best_j
    res.yend = best_j;
  }
  return res;
}

// Reference trace(), src/swg.rs:170-207, without the leading Xclip.  Walks from
// (i, j) back to the origin; op k of the walk (k = 0 is the cell at the max) is
// written to ops[k * stride] with stride = +1 or -1 (so a caller can lay the
// path out in either direction).  Returns the number of ops, or -1 on an
// inconsistent trace.
//
// The walk is a serial dependency chain in the reference.  Here the ballot words
// of 64 consecutive columns are held in registers (lane t holds column jhi - t)
// and a whole diagonal run is resolved at once: every lane assumes the path
// reaches its column along the current diagonal, reads its own cell's direction,
// and one ballot tells how many consecutive lanes really continue diagonally.
// Those lanes store their Match/Subst ops together; only gap steps are taken one
// at a time.  An alignment is typically a handful of runs.
template <int CPL>
__device__ int swg_traceback_wave(const unsigned long long* trace, int i, int j, int bw, uint8_t* ops, int stride,
                                  int max_ops) {
  const int lane = lane_id();
  i = bcast_first(i);
  j = bcast_first(j);
  bw = bcast_first(bw);
  int n = 0;
  int jhi = -1;
  unsigned long long w_lo[CPL], w_hi[CPL];
#pragma unroll
  for (int c = 0; c < CPL; c++) w_lo[c] = w_hi[c] = 0;
  while (i > 0 || j > 0) {
    if (j == 0) {  // column 0 is all Ins (reference :65,:70): i more insertions
      if (n + i > max_ops) return -1;
      #pragma unroll 1
      for (int s = lane; s < i; s += 64) ops[(n + s) * stride] = (uint8_t)OPK_INS;
      n += i;
      break;
    }
    if (jhi < 0 || j < jhi - 63) {
      jhi = j;
      const int col = j - lane;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        unsigned long long lo = 0, hi = 0;
        if (col >= 1) {
          lo = trace[((size_t)col * CPL + c) * 2 + 0];
          hi = trace[((size_t)col * CPL + c) * 2 + 1];
        }
        w_lo[c] = lo;
        w_hi[c] = hi;
      }
    }
    const int t0 = jhi - j;
    // hypothesis: the path runs diagonally from (i, j); this lane looks at (i - s, j - s)
    const int s = lane - t0;
    const int ii = i - s, jj = j - s;
    const int top = max(jj - bw, 0);
    const int b = ii - top;
    const bool cell_ok = (s >= 0) && (jj >= 1) && (ii >= 0) && (b >= 0) && (b < 64 * CPL);
    const int bb = cell_ok ? b : 0;
    const int l = bb / CPL, cc = bb % CPL;
    unsigned long long wl = w_lo[0], wh = w_hi[0];
#pragma unroll
    for (int c = 1; c < CPL; c++) {
      wl = (cc == c) ? w_lo[c] : wl;
      wh = (cc == c) ? w_hi[c] : wh;
    }
    const int dir = (int)((wl >> l) & 1ull) | ((int)((wh >> l) & 1ull) << 1);
    const unsigned long long okm = __ballot(cell_ok);
    if (!((okm >> t0) & 1ull)) return -1;
    const unsigned long long dm = __ballot(cell_ok && ii >= 1 && dir <= OPK_SUBST) >> t0;
    int run = (~dm == 0ull) ? 64 : __builtin_ctzll(~dm);
    run = min(run, 64 - t0);
    if (run > 0) {
      if (n + run > max_ops) return -1;
      if (s >= 0 && s < run) ops[(n + s) * stride] = (uint8_t)dir;
      n += run;
      i -= run;
      j -= run;
      continue;
    }
    const int op = __builtin_amdgcn_readlane(dir, t0);
    if (op <= OPK_SUBST) return -1;  // a diagonal move out of row 0
    if (n >= max_ops) return -1;
    if (lane == 0) ops[n * stride] = (uint8_t)op;
    n++;
    if (op == OPK_INS) {
      if (i == 0) return -1;
      i--;
    } else {
      j--;
    }
  }
  return n;
}


// ---------------------------------------------------------------------------------------------
// Shortcut without DP -- the commonest extension by far: the seed ended on a substitution and the rest of the
// read matches (x[0] != y[0], x[1..] == y[1..|x|)).  Then SwgExtend::extend's result is known:
//     score |x| - 2 at (|x|, |x|), ops = Subst, Match x (|x| - 1)
// provided |x| >= 3 (else the maximum stays 0 at the origin), |y| >= |x|, x_drop >= 1 (no X-drop on the
// way: the diagonal holds j - 2 in column j against a running maximum of max(0, j - 2)) and x is not one
// repeated base.  Proof sketch (scores: match +1, mismatch -1, gap of g costs 1 + g, a leading deletion run
// only g -- SURVEY.md Appendix A.2): a cell (i, j) scores at most i minus its penalties.  Reaching
// |x| - 2 or more needs i >= |x| - 2 with penalties <= i - |x| + 2.  Row |x| - 2: no penalty at all, i.e.
// the bare diagonal, which starts with the mismatch -- impossible.  Row |x| - 1: penalty <= 1, i.e. one
// leading deletion and then x[0..|x|-1) == y[1..|x|) -- that makes x one repeated base (x[k] == y[k+1] ==
// x[k+1]): excluded.  Row |x|: penalty <= 2: the diagonal with its one Subst (ends in column |x|), or a
// leading deletion run of g <= 2 and an exact match behind it (ends in column |x| + g with |x| - g: g = 1
// would beat the diagonal, but again needs x[k] == y[k+1] for all k: one repeated base), or one inserted
// base (penalty 2) with x[1..] == y[0..]: |x| - 3.  So the maximum is |x| - 2, first attained -- in the
// reference's scan order, columns then rows -- at (|x|, |x|); on the diagonal every cell holds i - 2 through
// the diagonal move (any other path to (i, i) pairs a deletion with an insertion: penalty >= 3), and
// triple_max prefers the diagonal on ties, so the trace is the diagonal.  The result depends on y[0..|x|)
// and on |y| >= |x| only (jmax, broke).  37.7 % of the DP columns of the benchmark workload were such
// extensions (tools/perf.py).
__device__ __forceinline__ bool swg_one_mismatch_shortcut(const uint8_t* xs, int dx, int xlen, const uint8_t* ys, int dy,
                                                          int ylen, int xd, uint8_t* ops, int stride, int max_ops, SwgResult& r,
                                                          int& n_ops) {
  if (xlen < 3 || ylen < xlen || xd < 1 || xlen > max_ops) return false;
  const int lane = lane_id();
  const int x0 = (int)xs[0];
  bool bad = (lane == 0) && (x0 == (int)ys[0]);  // the shape starts with a mismatch
  bool other = false;                            // some base of x differs from x[0]
#pragma unroll 1
  for (int t0 = 1; t0 < xlen; t0 += 64) {
    const int t = t0 + lane;
    if (t < xlen) {
      const int xc = (int)xs[t * dx];
      bad = bad || xc != (int)ys[t * dy];
      other = other || xc != x0;
    }
  }
  if (__ballot(bad) != 0ull || __ballot(other) == 0ull) return false;
  // walk order: op k belongs to the cell reached after k diagonal steps back from (|x|, |x|)
#pragma unroll 1
  for (int k = lane; k < xlen; k += 64) ops[k * stride] = (uint8_t)(k == xlen - 1 ? OPK_SUBST : OPK_MATCH);
  r.score = xlen - 2;
  r.xend = xlen;
  r.yend = xlen;
  r.cells = 0;
  r.cols = 0;
  r.jmax = xlen;
  r.broke = true;
  n_ops = xlen;
  return true;
}


// ---------------------------------------------------------------------------------------------
// Band of any width (the reference allocates whatever 2*max_band_width+1 asks for, src/swg.rs:17-26):
// the slots of a column are walked in tiles of 64 (slot b = 64*t + lane); the column state D, C of the
// previous and the current column lives in four wave-private int arrays of `stride` entries each
// (dp[0..stride) = D even columns, then C even, D odd, C odd; stride >= 2*bw + 66), the vertical chain
// is the same exclusive prefix-max with a carry from tile to tile, the trace is two ballots per tile and
// column at trace[(j * T + t) * 2], T = ceil((2*bw+1) / 64).  Same results as swg_extend_wave: same
// recurrences, tie rules, X-drop and exact early exit.  This is the slow path: it serves the rare reads
// whose band exceeds what the register-resident kernels hold, not the benchmark.
__device__ __forceinline__ void tiled_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ inline SwgResult swg_extend_tiled(const uint8_t* xs, int dx, int xlen, const uint8_t* ys, int dy, int ylen, int bw, int xd,
                                             unsigned long long* trace, int* dp, int stride) {
  SwgResult res;
  res.score = 0;
  res.xend = 0;
  res.yend = 0;
  res.cells = 0;
  res.cols = 0;
  res.jmax = 0;
  res.broke = (xlen == 0);
  if (xlen == 0 || ylen == 0) return res;  // reference :39-55
  const int lane = lane_id();
  const int w = 2 * bw + 1;
  const int T = (w + 63) >> 6;
  constexpr int ge = GAP_EXTEND, go = GAP_OPEN;
  int* Dbuf[2] = {dp, dp + 2 * (size_t)stride};
  int* Cbuf[2] = {dp + stride, dp + 3 * (size_t)stride};
  // reference :62-71 leftmost column (one slot past the band is written too: phase 2 reads slot b + 1)
  for (int b = lane; b < T * 64 + 1; b += 64) {
    Dbuf[0][b] = (b == 0) ? 0 : b * ge + go;
    Cbuf[0][b] = (b == 0) ? 0 : MIN_SCORE;
    Dbuf[1][b] = MIN_SCORE;
    Cbuf[1][b] = MIN_SCORE;
  }
  tiled_sync();
  int run_max = 0, best_j = 0, best_b = 0;
  bool finished = false;
  int last_j = 0;
  bool empty_col = false;
  const int p1_end = min(bw, ylen);
  const int rows1 = min(w, xlen + 1);

  // one column; returns true when the walk stops after it
  auto column = [&](int j, bool phase2) -> bool {
    const int top = phase2 ? j - bw : 0;
    const int nvalid = phase2 ? min(w, xlen + 1 - top) : rows1;
    const int tv = (nvalid + 63) >> 6;
    const int* Dp = Dbuf[(j - 1) & 1];
    const int* Cp = Cbuf[(j - 1) & 1];
    int* Dc = Dbuf[j & 1];
    int* Cc = Cbuf[j & 1];
    const int yc = (int)ys[(j - 1) * dy];
    const int xfloor = run_max - xd;
    const int alive_floor = run_max - xlen + top;
    int carry = NEG;
    unsigned long long any_imp = 0, any_x = 0, any_alive = 0;
    int col_bmin = -1;
    unsigned long long* tr = trace + (size_t)j * T * 2;
    for (int t = 0; t < tv; t++) {
      const int b = t * 64 + lane;
      const bool valid = b < nvalid;
      int d, Cn;
      bool eq;
      if (!phase2) {
        // reference :80-98: C and D of the same slot, diagonal from the slot above
        const int dprev = Dp[b], cprev = Cp[b];
        const int dabove = (b > 0) ? Dp[b - 1] : MIN_SCORE;
        Cn = max(cprev, dprev + go) + ge;
        const int xi = min(max(b - 1, 0), xlen - 1);
        eq = (int)xs[xi * dx] == yc;
        d = dabove + (eq ? MATCH_SCORE : MISMATCH_SCORE);
      } else {
        // reference :119-140: C from slot b + 1 of the previous column (same row), diagonal from the same slot
        const int dprev = Dp[b];
        const int dnext = Dp[b + 1], cnext = Cp[b + 1];
        Cn = (b >= w - 1) ? MIN_SCORE : max(cnext, dnext + go) + ge;
        const int xi = min(max(top + b - 1, 0), xlen - 1);
        eq = (int)xs[xi * dx] == yc;
        d = dprev + (eq ? MATCH_SCORE : MISMATCH_SCORE);
      }
      // slots without a cell (only at the top of the last tile) sit above every slot with one: nothing feeds down from them
      const int key = max(d, Cn) - b * ge;
      const int incl = wave_incl_max_scan(key);
      const int run = max(carry, wave_shr1(incl, NEG));
      carry = max(carry, __builtin_amdgcn_readlane(incl, 63));
      const int R = run + (go + b * ge);
      const int Dn = max(max(d, Cn), R);
      const unsigned long long vm = vote(valid);
      // direction bits (Match 0, Subst 1, Del 2, Ins 3; priority diag > Del > Ins, reference :226-240)
      const unsigned long long hi = vote(Dn != d);
      const unsigned long long lo = (~hi & ~vote(eq)) | (hi & vote(Dn != Cn));
      if (lane == 0) {
        tr[t * 2 + 0] = lo;
        tr[t * 2 + 1] = hi;
      }
      const unsigned long long m_c = vote(Dn > run_max) & vm;
      if (m_c && col_bmin < 0) col_bmin = t * 64 + (int)__builtin_ctzll(m_c);
      any_imp |= m_c;
      any_x |= vote(Dn >= xfloor) & vm;
      any_alive |= vote(Dn - b > alive_floor) & vm;
      Dc[b] = Dn;
      Cc[b] = Cn;
    }
    tiled_sync();
    last_j = j;
    const bool imp = any_imp != 0ull;
    if (imp) {
      best_j = j;
      best_b = col_bmin;
      run_max += MATCH_SCORE;
    }
    if (!phase2) return any_alive == 0ull;  // reference :110 cannot fire in phase 1 (x_drop >= band_width); the early exit is ours
    return !imp && (any_x == 0ull || any_alive == 0ull);
  };

  for (int j = 1; j <= p1_end; j++) {
    if (column(j, false)) {
      finished = true;
      res.broke = true;
      break;
    }
  }
  if (!finished && bw + 1 <= ylen) {
    for (int j = bw + 1; j <= ylen; j++) {
      if (j - bw > xlen) {  // empty row range: band_max = MIN -> X-drop (reference :117-153)
        res.broke = true;
        empty_col = true;
        break;
      }
      if (column(j, true)) {
        res.broke = true;
        break;
      }
    }
  }
  // counters (same closed form as swg_extend_wave)
  {
    res.jmax = last_j;
    res.cols = (unsigned)last_j + (empty_col ? 1u : 0u);
    const int n1 = min(last_j, p1_end);
    unsigned cells = (unsigned)n1 * (unsigned)rows1;
    if (last_j > p1_end) {
      const int t1 = last_j - bw;
      const int tfull = min(t1, xlen + 1 - w);
      if (tfull >= 1) cells += (unsigned)tfull * (unsigned)w;
      const int ta = max(tfull, 0) + 1;
      if (t1 >= ta) {
        const int hi_cells = xlen + 1 - ta, lo_cells = xlen + 1 - t1;
        cells += (unsigned)(((long long)(hi_cells + lo_cells) * (t1 - ta + 1)) / 2);
      }
    }
    res.cells = cells;
  }
  if (run_max > 0) {
    res.score = run_max;
    res.xend = max(best_j - bw, 0) + best_b;
    res.yend = best_j;
  }
  return res;
}

// trace() for the tiled layout: every lane tests the cell its diagonal hypothesis lands on (one round trip to
// the trace per diagonal run); gap steps one at a time.
__device__ inline int swg_traceback_tiled(const unsigned long long* trace, int i, int j, int bw, uint8_t* ops, int stride, int max_ops) {
  const int lane = lane_id();
  i = bcast_first(i);
  j = bcast_first(j);
  bw = bcast_first(bw);
  const int w = 2 * bw + 1;
  const int T = (w + 63) >> 6;
  int n = 0;
  while (i > 0 || j > 0) {
    if (j == 0) {  // column 0 is all Ins (reference :65,:70)
      if (n + i > max_ops) return -1;
#pragma unroll 1
      for (int s = lane; s < i; s += 64) ops[(n + s) * stride] = (uint8_t)OPK_INS;
      n += i;
      break;
    }
    const int ii = i - lane, jj = j - lane;
    const int top = max(jj - bw, 0);
    const int b = ii - top;
    const bool cell_ok = (jj >= 1) && (ii >= 0) && (b >= 0) && (b < w);
    int dir = 0;
    if (cell_ok) {
      const unsigned long long* tw = trace + ((size_t)jj * T + (b >> 6)) * 2;
      dir = (int)((tw[0] >> (b & 63)) & 1ull) | ((int)((tw[1] >> (b & 63)) & 1ull) << 1);
    }
    const unsigned long long okm = __ballot(cell_ok);
    if (!(okm & 1ull)) return -1;
    const unsigned long long dm = __ballot(cell_ok && ii >= 1 && dir <= OPK_SUBST);
    const int run = (~dm == 0ull) ? 64 : (int)__builtin_ctzll(~dm);
    if (run > 0) {
      if (n + run > max_ops) return -1;
      if (lane < run) ops[(n + lane) * stride] = (uint8_t)dir;
      n += run;
      i -= run;
      j -= run;
      continue;
    }
    const int op = __builtin_amdgcn_readlane(dir, 0);
    if (op <= OPK_SUBST) return -1;  // a diagonal move out of row 0
    if (n >= max_ops) return -1;
    if (lane == 0) ops[n * stride] = (uint8_t)op;
    n++;
    if (op == OPK_INS) {
      if (i == 0) return -1;
      i--;
    } else {
      j--;
    }
  }
  return n;
}

}  // namespace dev
}  // namespace thm
#endif
