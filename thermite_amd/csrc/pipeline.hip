// pipeline.hip -- the batched read-level entry points of include/thermite.h:
//   thm_batch_upload / thm_batch_run / thm_batch_sync / thm_batch_fetch,
//   thm_align_batch (= the three in sequence) and thm_smems_batch.
//
// One batch = count -> scan -> fill on the device:
//   seed kernels    SMEMs per read (pool + per-read run), hit counts
//   plan kernel     lists: reads with many hits (longest jobs first), reads of the slow class
//   scan            hit counts -> per-read slice of the candidate array
//   extend kernel   align_read per read; accepted alignments into the slice,
//                   op streams into a bump-allocated pool; final order list.
//                   Reads whose band or length exceeds what the register-resident kernel
//                   holds (the slow class) run afterwards in the any-width kernel.
//   scans           alignment counts / op bytes -> output offsets
//   compact kernel  canonical output (alignments in read order, op streams
//                   back to back), so the D2H copy is a plain memcpy
// Pool capacities are heuristics; a kernel that runs out sets a fault bit and
// thm_batch_sync grows the pools and replays the batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "aligner_internal.h"

using namespace thm;

namespace {

// bw0(L) of reference src/aligner.rs:130-138; non-decreasing in L
int band_for_len(const thm_align_opts& o, uint32_t L) {
  volatile float prod = o.min_aln_score_percent * (float)L;
  float pv = prod;
  int pct = (pv != pv) ? 0 : (pv >= 2147483648.0f ? 2147483647 : (pv <= -2147483648.0f ? (-2147483647 - 1) : (int)pv));
  int ms = std::max(pct, o.min_aln_score);
  if (ms < 0) return 0;
  return std::max((int)L - ms, 0);
}

int blocks_for(const thm_aligner* a, uint64_t n, size_t lds_per_block) {
  int per_cu = 8;
  if (lds_per_block > 0) per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds_per_block);
  if (per_cu < 1) per_cu = 1;
  return grid_blocks(a, n, 4, per_cu);
}

// device memory the any-width kernel may take for its wave-private buffers (traces grow with length x band)
constexpr uint64_t SLOW_SCRATCH_BUDGET = 24ull << 30;

template <class C>
const DeviceIndexT<C>& dev_view(const thm_aligner* a);
template <>
const DeviceIndexT<uint32_t>& dev_view<uint32_t>(const thm_aligner* a) {
  return a->dix->view;
}
template <>
const DeviceIndexT<uint64_t>& dev_view<uint64_t>(const thm_aligner* a) {
  return a->dix->view64;
}

template <class C>
int enqueue_seed_t(thm_aligner* a, uint32_t min_seed_len) {
  const uint64_t n = a->n_reads;
  hipStream_t s = a->stream;
  HIPCHK(a, a->s_off.ensure((n + 1) * 8));
  HIPCHK(a, a->s_cnt.ensure((n + 1) * 4));
  HIPCHK(a, a->s_hits.ensure((n + 1) * 8));
  HIPCHK(a, a->s_cand_off.ensure((n + 2) * 8));
  HIPCHK(a, a->scan_tmp.ensure(scan_tmp_entries(n + 1) * 8 + 64));
  HIPCHK(a, a->r_status.ensure((n + 1) * 4));
  HIPCHK(a, hipMemsetAsync(a->r_status.p, 0, (n + 1) * 4, s));
  // typical: 1-2 SMEMs per read; waves take the pool in 256-entry slices, hence the fixed slack
  const uint64_t smem_min = a->dbg_smem_cap ? a->dbg_smem_cap : n * 4 + (4u << 20);
  if (a->smem_cap < smem_min) a->smem_cap = smem_min;
  HIPCHK(a, a->s_smems.ensure(a->smem_cap * sizeof(SmemT<C>)));
  if (!a->dbg_smem_cap) a->smem_cap = std::max<uint64_t>(a->smem_cap, a->s_smems.cap / sizeof(SmemT<C>));
  // to_ascii_uppercase (src/aligner.rs:125) + sanitising, once per run for both kernels
  // (16 bytes in front: a left extension is read in whole 8-byte words that may begin a few bytes before the first read)
  HIPCHK(a, a->r_san.ensure(a->n_bases + 256 + 16));
  HIPCHK(a, hipMemsetAsync(a->r_san.p, 0, 16, s));
  HIPCHK(a, launch_sanitize(a->r_bases.as<uint8_t>(), a->r_san.as<uint8_t>() + 16, a->n_bases, a->n_bases + 128, s));
  // length classes of the seed stage (launch.h): short reads take the byte-per-position paths
  uint64_t n_long = 0;
  uint32_t max_short = 0, max_long = 0;
  for (const auto& lc : a->len_hist) {
    if (lc.first <= SHORT_READ_MAX) {
      max_short = std::max(max_short, lc.first);
    } else {
      max_long = std::max(max_long, lc.first);
      n_long += lc.second;
    }
  }
  const uint64_t n_short = n - n_long;
  // ragged per-position rows (launch.h, ms_row)
  const uint64_t items = ms_row(a->n_bases, n) + 64;
  HIPCHK(a, a->s_ms_end.ensure(items * 2 + 64));
  HIPCHK(a, a->s_ms_lo.ensure(items * sizeof(C) + 64));
  HIPCHK(a, a->s_ms_hi.ensure(items * sizeof(C) + 64));
  auto cells_of = [&](uint32_t max_len) -> uint64_t {
    const uint32_t P = (max_len >= min_seed_len) ? max_len - min_seed_len + 1 : 1;
    return (P + 7) / 8;
  };
  const uint64_t cells = n_short * cells_of(max_short) + n_long * cells_of(max_long);
  HIPCHK(a, a->s_work_reads.ensure((n + 1) * 8));
  HIPCHK(a, a->s_work_long.ensure((n_long + 1) * 8));
  HIPCHK(a, a->s_work_cells.ensure((std::max(cells, n) + 1) * 8));
  HIPCHK(a, a->s_work_counts.ensure(64));
  HIPCHK(a, hipMemsetAsync(a->s_work_counts.p, 0, 64, s));
  int rc = reset_queue(a);
  if (rc != THM_OK) return rc;
  HIPCHK(a, hipMemsetAsync(a->d_cursors.p, 0, 64, s));
  const int n_blocks = blocks_for(a, n, std::min(seed_select_lds_bytes(std::max(max_short, 1u)), SEED_SELECT_LDS_LIMIT));
  SeedParamsT<C> sp;
  sp.ix = dev_view<C>(a);
  sp.reads.bases = a->r_san.as<uint8_t>() + 16;
  sp.reads.offsets = a->r_offsets.as<uint64_t>();
  sp.reads.n_reads = n;
  sp.min_seed_len = min_seed_len;
  sp.max_len_short = max_short;
  sp.max_len_long = max_long;
  sp.n_long = n_long;
  sp.ms_end = a->s_ms_end.as<uint16_t>();
  sp.ms_lo = a->s_ms_lo.as<C>();
  sp.ms_hi = a->s_ms_hi.as<C>();
  sp.work_short = a->s_work_reads.as<unsigned long long>();
  sp.work_long = a->s_work_long.as<unsigned long long>();
  sp.work_cells = a->s_work_cells.as<unsigned long long>();
  sp.work_counts = a->s_work_counts.as<unsigned long long>();
  sp.smems = a->s_smems.as<SmemT<C>>();
  sp.smem_cap = a->smem_cap;
  sp.cursor = a->d_cursors.as<unsigned long long>();
  sp.read_smem_off = a->s_off.as<uint64_t>();
  sp.read_smem_cnt = a->s_cnt.as<uint32_t>();
  sp.read_hits = a->s_hits.as<uint64_t>();
  sp.read_status = a->r_status.as<int32_t>();
  sp.counters = a->d_counters.as<unsigned long long>();
  sp.queue = a->d_queue.as<unsigned int>();
  sp.fault = a->d_fault.as<int>();
  sp.sel_scratch = nullptr;
  sp.sel_scratch_per_wave = 0;
  static const uint32_t fill_mode = [] {
    const char* e = getenv("THM_SEED_FILL");
    const int v = e ? atoi(e) : 0;
    return (uint32_t)(v >= 0 && v <= 2 ? v : 0);
  }();
  sp.fill_mode = fill_mode;
  sp.fill_keys = nullptr;
  sp.fill_perm = nullptr;
  sp.fill_hist = nullptr;
  if (fill_mode == 2) {
    const uint64_t slots = (std::max(cells, n) + 1) * 8;
    HIPCHK(a, a->s_fill_keys.ensure(slots * 2));
    HIPCHK(a, a->s_fill_perm.ensure(slots * 4));
    HIPCHK(a, a->s_fill_hist.ensure((2 * (FILL_BUCKETS + 1) + 1) * 4));
    HIPCHK(a, hipMemsetAsync(a->s_fill_hist.p, 0, (2 * (FILL_BUCKETS + 1) + 1) * 4, s));
    sp.fill_keys = a->s_fill_keys.as<uint16_t>();
    sp.fill_perm = a->s_fill_perm.as<uint32_t>();
    sp.fill_hist = a->s_fill_hist.as<unsigned int>();
  }
  if (n_long && seed_select_lds_bytes(max_long) > SEED_SELECT_LDS_LIMIT) {
    // reads of thousands of bases: the selection kernel's per-read lists go to global memory
    const int nb = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_blocks, (n_long + 3) / 4));
    sp.sel_scratch_per_wave = (seed_select_scratch_bytes(max_long) + 255) & ~255ull;
    HIPCHK(a, a->s_sel_scratch.ensure((size_t)nb * 4 * sp.sel_scratch_per_wave + 256));
    sp.sel_scratch = a->s_sel_scratch.as<uint8_t>();
  }
  HIPCHK(a, launch_seed(sp, n_blocks, s));
  // hit counts -> offsets of each read's slice (also the Mem offsets of thm_smems_batch)
  HIPCHK(a, launch_exclusive_scan_u64(a->s_hits.as<uint64_t>(), a->s_cand_off.as<uint64_t>(), n,
                                      a->scan_tmp.as<uint64_t>(), s));
  return THM_OK;
}

int enqueue_seed(thm_aligner* a, uint32_t min_seed_len) {
  return a->dix->wide ? enqueue_seed_t<uint64_t>(a, min_seed_len) : enqueue_seed_t<uint32_t>(a, min_seed_len);
}

// Length classes of the extend stage for the current options.  Band and buffer sizes grow with the read
// length, so each class is a range of lengths:
//   fast   L <= fast_max: band within +-127 and the wave-private buffers within LDS -> register-resident kernel;
//   slow   fast_max < L <= slow_max: any-width kernel (band in tiles, buffers in global memory);
//   beyond slow_max: the DP trace alone would exceed the memory budget -> per-read THM_ERR_UNSUPPORTED.
struct ExtClasses {
  uint32_t fast_max = 0, fast_len = 0, fast_bw = 0;  // threshold; longest fast read present and its band
  uint32_t slow_max = 0, slow_len = 0, slow_bw = 0;
  uint64_t n_slow = 0;
};
ExtClasses classify(const thm_aligner* a, uint32_t mk_cap_slow) {
  ExtClasses c;
  bool fast_open = true;
  for (const auto& lc : a->len_hist) {
    const uint32_t L = lc.first;
    const uint32_t bw = (uint32_t)band_for_len(a->opts, L);
    const int cpl = (int)((2 * bw + 1 + 63) / 64);
    if (fast_open && cpl <= 4 && extend_lds_bytes(L, bw, cpl) <= EXTEND_LDS_LIMIT) {
      c.fast_len = L;
      c.fast_bw = bw;
      continue;
    }
    fast_open = false;
    if (extend_slow_scratch_bytes(L, bw, mk_cap_slow) > SLOW_SCRATCH_BUDGET) break;
    c.slow_len = L;
    c.slow_bw = bw;
    c.n_slow += lc.second;
  }
  c.fast_max = c.fast_len;
  c.slow_max = std::max(c.slow_len, c.fast_len);
  return c;
}

template <class C>
int enqueue_extend_t(thm_aligner* a, const ExtClasses& cls, uint32_t mk_cap_slow, bool retry_possible) {
  const uint64_t n = a->n_reads;
  hipStream_t s = a->stream;
  ExtendParamsT<C> ep;
  ep.ix = dev_view<C>(a);
  ep.reads.bases = a->r_san.as<uint8_t>() + 16;
  ep.reads.offsets = a->r_offsets.as<uint64_t>();
  ep.reads.n_reads = n;
  ep.opts = a->opts;
  ep.smems = a->s_smems.as<SmemT<C>>();
  // one record per read (offsets, SMEM run, candidate slice, first SMEM and its first occurrence)
  HIPCHK(a, a->e_recs.ensure((n + 1) * sizeof(ReadRecT<C>)));
  {
    PackParamsT<C> pk;
    pk.sa = ep.ix.sa;
    pk.offsets = a->r_offsets.as<uint64_t>();
    pk.n_reads = n;
    pk.smems = ep.smems;
    pk.read_smem_off = a->s_off.as<uint64_t>();
    pk.read_smem_cnt = a->s_cnt.as<uint32_t>();
    pk.read_cand_off = a->s_cand_off.as<uint64_t>();
    pk.fault_seed = a->d_fault.as<int>();
    pk.recs = a->e_recs.as<ReadRecT<C>>();
    HIPCHK(a, launch_pack_reads(pk, s));
  }
  ep.read_recs = a->e_recs.as<ReadRecT<C>>();
  ep.heavy = a->s_heavy.as<unsigned long long>();
  ep.heavy_count = a->s_work_counts.as<unsigned long long>() + 2;
  ep.team = a->s_team.as<unsigned long long>();
  ep.team_count = a->s_work_counts.as<unsigned long long>() + 7;
  ep.team_limit = 2u * (uint32_t)a->n_cu;
  ep.cands = a->e_cands.as<Cand>();
  ep.cand_cap = a->cand_cap;
  ep.order = a->e_order.as<uint32_t>();
  ep.cand_ops = a->e_ops.as<uint8_t>();
  ep.cand_ops_cap = a->cand_ops_cap;
  ep.ops_cursor = a->d_cursors.as<unsigned long long>() + 1;
  ep.read_n_alns = a->e_nalns.as<uint32_t>();
  ep.read_op_bytes = a->e_opbytes.as<uint64_t>();
  ep.read_status = a->r_status.as<int32_t>();
  ep.retry = a->s_slow.as<unsigned long long>();
  ep.retry_count = a->s_work_counts.as<unsigned long long>() + 5;
  ep.n_contract = a->s_work_counts.as<unsigned long long>() + 6;
  ep.counters = a->d_counters.as<unsigned long long>();
  ep.queue = a->d_queue.as<unsigned int>();
  ep.fault = a->d_fault.as<int>() + 1;
  ep.fault_seed = a->d_fault.as<int>();
  ep.prof = a->d_counters.as<unsigned long long>() + 2 * THM_N_COUNTERS;
  ep.slow_scratch = nullptr;
  ep.slow_scratch_per_wave = 0;
  ep.trace_scratch = nullptr;
  // ---- fast class ----
  ep.max_read_len = cls.fast_len;
  ep.max_bw = cls.fast_bw;
  if (a->dbg_band_clip) ep.max_bw = std::min<uint32_t>(ep.max_bw, a->dbg_band_clip - 1);  // test hook: reads beyond get THM_ERR_INTERNAL
  ep.mk_cap = FAST_MAX_YCLIPS;
  ep.list_only = 0;
  const int cpl = std::max(1, (int)((2 * cls.fast_bw + 1 + 63) / 64));
  const size_t lds = extend_lds_bytes(cls.fast_len, cls.fast_bw, cpl);
  // workgroups that fit the machine at once: LDS and the kernel's register budget
  const int ext_blocks = std::min(blocks_for(a, n, lds), a->n_cu * extend_waves_per_simd(cpl, sizeof(C) == 8));
  const bool team_ok = cpl <= 2 && team_fits_lds(lds);
  // global trace scratch (extensions over more than 64 band slots): the team kernel runs BESIDE the main kernel, so
  // its waves have their own slices behind the main kernel's
  const size_t trace_per_wave = extend_trace_scratch_bytes(cls.fast_len, cls.fast_bw, cpl);
  const size_t main_trace_waves = (size_t)ext_blocks * 4, team_trace_waves = team_ok ? (size_t)a->n_cu * TEAM_WAVES : 0;
  HIPCHK(a, a->e_trace.ensure((main_trace_waves + team_trace_waves) * trace_per_wave + 64));
  ep.trace_scratch = a->e_trace.as<unsigned long long>();
  // reads with very many hits: a workgroup per read (speculative chunks of hits, kernels_extend.hip TEAM) BESIDE the
  // wave-per-read kernel: such reads are the long jobs of a batch, a launch behind the main kernel would put them on
  // the critical path.  A team workgroup needs a whole CU's registers, so it must be placed before the persistent
  // waves of the main kernel fill the machine: the team kernel goes first on this stream, the main kernel on the second
  // stream behind an event (the event's latency is the team's head start; the other way round the team kernel started
  // 10 us late and waited for the main kernel to drain -- rocprofv3 kernel trace, tools/overlap_trace.py).
  // rows for the waves' counters: [main | team | slow] (launch.h, ExtendParamsT::wave_counters)
  uint64_t slow_waves = 0, slow_per_wave = 0;
  if (cls.n_slow || retry_possible) {
    const uint32_t sl_len = std::max(cls.slow_len, cls.fast_len), sl_bw = std::max(cls.slow_bw, cls.fast_bw);
    slow_per_wave = (extend_slow_scratch_bytes(sl_len, sl_bw, mk_cap_slow) + 255) & ~255ull;
    slow_waves = std::max<uint64_t>(1, std::min<uint64_t>(SLOW_SCRATCH_BUDGET / std::max<uint64_t>(slow_per_wave, 1), (uint64_t)a->n_cu * 8));
    if (!retry_possible) slow_waves = std::min(slow_waves, std::max<uint64_t>(cls.n_slow, 1));
    slow_waves = (slow_waves + 3) / 4 * 4;
  }
  const uint64_t main_rows = (uint64_t)ext_blocks * 4, team_rows = team_ok ? (uint64_t)a->n_cu * TEAM_WAVES : 0;
  // ---- extension problems as the unit of wavefront work (kernels_tpr.hip): rounds of [control kernel, thread per
  // read | DP kernel, wave per request]; what it cannot take goes on the wave-per-read kernel's list ----
  const bool tpr = a->use_tpr && n > 0 && cpl <= 4;
  const int ctl_blocks = tpr ? (int)std::min<uint64_t>((n + 255) / 256, (uint64_t)a->n_cu * TPR_CTL_BLOCKS_PER_CU) : 0;
  // DP workgroups that fit the machine at once (LDS: x, y, trace and op buffer of four waves)
  const uint32_t dp_x_cap = (cls.fast_len + 64u + 15u) & ~15u, dp_y_cap = (cls.fast_len + cls.fast_bw + 2u + 64u + 15u) & ~15u;
  const int dp_per_cu = (int)std::max<size_t>(1, std::min<size_t>(TPR_DP_BLOCKS_PER_CU, EXTEND_LDS_LIMIT / std::max<size_t>(1, extend_dp_lds_bytes(dp_x_cap, dp_y_cap))));
  const int dp_blocks = tpr ? a->n_cu * dp_per_cu : 0;
  // thread-per-problem kernel (class 0: at most DPT_SLOTS band slots hold cells): columns of its LDS trace -- such a problem
  // has a band of at most +-7 (|y| <= |x| + 8) or an x of at most 15 symbols (|y| <= 16 + bw); workgroups that fit a CU
  // (at most 128 columns, 64 KiB of LDS per workgroup: longer ones are the wave-per-problem kernel's)
  const uint32_t dpt_tcols = std::min<uint32_t>(std::min<uint32_t>(dp_y_cap, 128u), std::max<uint32_t>(cls.fast_len + DPT_SLOTS / 2 + 2, DPT_SLOTS + cls.fast_bw + 2));
  const int dpt_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, EXTEND_LDS_LIMIT / std::max<size_t>(1, extend_dpt_lds_bytes(dpt_tcols))));
  const int dpt_blocks = tpr ? a->n_cu * dpt_per_cu : 0;
  // rows: [main | team | slow | control kernel's workgroups | the wave-per-read launch for what the control kernel leaves]
  const int bail_blocks = tpr ? std::min(ext_blocks, a->n_cu) : 0;
  const uint64_t tpr_rows = (uint64_t)ctl_blocks + (uint64_t)bail_blocks * 4;
  const uint64_t n_rows = main_rows + team_rows + slow_waves + tpr_rows;
  HIPCHK(a, a->e_wcnt.ensure(n_rows * THM_N_COUNTERS * 8 + 64));
  HIPCHK(a, hipMemsetAsync(a->e_wcnt.p, 0, n_rows * THM_N_COUNTERS * 8, s));
  ep.wave_counters = a->e_wcnt.as<unsigned long long>();
  ep.skip_scan = 0;
  if (tpr) {
    const uint64_t rec_cap = std::min<uint64_t>(4 * n + 65536, 64ull << 20);
    HIPCHK(a, a->t_memos.ensure(n * sizeof(ReadMemo) + 64));
    HIPCHK(a, a->t_recs.ensure(rec_cap * sizeof(DpRec) + 64));
    HIPCHK(a, a->t_qlist.ensure((size_t)DP_NQ * rec_cap * 4 + 64));
    HIPCHK(a, a->t_act[0].ensure(n * 4 + 64));
    HIPCHK(a, a->t_act[1].ensure(n * 4 + 64));
    HIPCHK(a, a->t_bail.ensure((n + 1) * 8));
    HIPCHK(a, a->t_queue2.ensure(thm::QUEUE_BYTES));
    HIPCHK(a, hipMemsetAsync(a->t_queue2.p, 0, thm::QUEUE_BYTES, s));
    HIPCHK(a, a->t_ctl.ensure(TPRC_BYTES));
    HIPCHK(a, hipMemsetAsync(a->t_ctl.p, 0, TPRC_BYTES, s));
    unsigned long long* ctl = a->t_ctl.as<unsigned long long>();
    TprParamsT<C> tq;
    tq.recs_rw = a->e_recs.as<ReadRecT<C>>();
    tq.memos = a->t_memos.as<ReadMemo>();
    tq.recs = a->t_recs.as<DpRec>();
    tq.rec_cap = rec_cap;
    tq.rec_cursor = ctl + TPRC_REC_CUR;
    tq.q_list = a->t_qlist.as<uint32_t>();
    tq.q_stride = rec_cap;
    tq.q_cur = ctl + TPRC_Q_CUR;
    tq.bail = a->t_bail.as<unsigned long long>();
    tq.bail_count = ctl + TPRC_BAIL_CNT;
    tq.team = nullptr;  // (the team kernel is running already; a read of this path has fewer hits than it takes anyway)
    tq.team_count = nullptr;
    tq.max_hits = TPR_MAX_HITS;
    tq.dpt_cols = dpt_tcols;
    tq.stats = ctl + TPRC_STATS;
    ExtendParamsT<C> zp = ep;
    zp.wave_counters = ep.wave_counters + (main_rows + team_rows + slow_waves) * THM_N_COUNTERS;
    DpParams dq;
    dq.recs = tq.recs;
    dq.q_list = tq.q_list;
    dq.q_stride = tq.q_stride;
    dq.q_cur = ctl + TPRC_Q_CUR;
    dq.q_done = ctl + TPRC_Q_DONE;
    dq.fault = ep.fault;
    dq.x_cap = dp_x_cap;
    dq.y_cap = dp_y_cap;
    {
      const size_t tb = extend_dp_trace_bytes(dp_y_cap, cpl);
      HIPCHK(a, a->t_trace.ensure((size_t)dp_blocks * 4 * tb + 64));
      dq.trace_scratch = a->t_trace.as<unsigned long long>();
      dq.trace_per_wave = tb / 8;
      dq.tcols = dpt_tcols;
    }
    // hit summaries: everything about a hit that does not depend on the state align_read carries, all hits side by side
    {
      const uint64_t slot_cap = a->cand_cap;
      HIPCHK(a, a->t_hdr.ensure(slot_cap * sizeof(HitHdr) + 64));
      HIPCHK(a, a->t_sums.ensure(slot_cap * sizeof(HitSum) + 64));
      HIPCHK(a, hipMemsetAsync(a->t_hdr.p, 0xFF, slot_cap * sizeof(HitHdr), s));
      HitParamsT<C> hq;
      hq.ix = ep.ix;
      hq.reads = ep.reads;
      hq.opts = ep.opts;
      hq.smems = ep.smems;
      hq.read_recs = ep.read_recs;
      hq.max_read_len = cls.fast_len;
      hq.max_hits = TPR_MAX_HITS;
      hq.hdr = a->t_hdr.as<HitHdr>();
      hq.sums = a->t_sums.as<HitSum>();
      hq.total_hits = a->s_cand_off.as<uint64_t>() + n;
      hq.slot_cap = slot_cap;
      hq.fault_seed = a->d_fault.as<int>();
      HIPCHK(a, launch_hit_expand(hq, s));
      HIPCHK(a, launch_hit_summaries(hq, a->n_cu * 12, s));
      tq.sums = a->t_sums.as<HitSum>();
    }
    // The reads with HEAVY_HITS hits and more (the lists of plan_kernel) are the wave-per-read / workgroup-per-read
    // kernels': they run BESIDE the rounds below, on their own streams.
    {
      ExtendParamsT<C> hp = ep;
      hp.skip_scan = 1;
      hp.team_limit = 16u * (uint32_t)a->n_cu;  // no scan of the batch to share the machine with: the team kernel keeps every read it can take
      HIPCHK(a, hipEventRecord(a->ev_fork, s));
      HIPCHK(a, hipStreamWaitEvent(a->stream2, a->ev_fork, 0));
      HIPCHK(a, hipStreamWaitEvent(a->stream3, a->ev_fork, 0));
      if (team_ok) {
        ExtendParamsT<C> tp = hp;
        tp.list_only = 1;
        tp.wave_counters = ep.wave_counters + main_rows * THM_N_COUNTERS;
        tp.trace_scratch = ep.trace_scratch + main_trace_waves * trace_per_wave / 8;
        HIPCHK(a, launch_extend(tp, cpl, a->n_cu, a->stream2, true));
      }
      HIPCHK(a, launch_extend(hp, cpl, std::min(ext_blocks, a->n_cu), a->stream3));  // (a few thousand reads: one workgroup per CU leaves the machine to the rounds)
      HIPCHK(a, hipEventRecord(a->ev_join, a->stream2));
      HIPCHK(a, hipEventRecord(a->ev_join3, a->stream3));
    }
    // round 0's list: the reads of this path by descending hit count
    HIPCHK(a, launch_tpr_order(a->e_recs.as<ReadRecT<C>>(), n, cls.fast_len, TPR_MAX_HITS, ctl + TPRC_BINS, a->t_act[0].as<uint32_t>(), ctl + TPRC_N_ACT,
                               a->d_fault.as<int>(), s));
    const int n_rounds = a->tpr_rounds;  // rounds of requests a read may take (then: the wave-per-read kernel)
    for (int r = 0; r <= n_rounds; r++) {
      tq.round = (uint32_t)r;
      tq.act_in = a->t_act[r & 1].as<uint32_t>();
      tq.n_act_in = ctl + TPRC_N_ACT + r;
      tq.act_out = a->t_act[(r + 1) & 1].as<uint32_t>();
      tq.n_act_out = ctl + TPRC_N_ACT + r + 1;
      tq.last_round = r == n_rounds ? 1u : 0u;
      HIPCHK(a, launch_extend_ctl(zp, tq, ctl_blocks, s));
      if (r == n_rounds) break;
      dq.work = (unsigned int*)((uint8_t*)a->t_ctl.p + TPRC_WORK_BYTES + (size_t)r * 4 * 64);
      // the two DP kernels of a round side by side: the thread-per-problem one has a few hundred wavefronts of long
      // serial work (one column after the other, per thread), the wave-per-problem one fills the machine
      HIPCHK(a, hipEventRecord(a->ev_dpt_fork, s));
      HIPCHK(a, hipStreamWaitEvent(a->stream4, a->ev_dpt_fork, 0));
      HIPCHK(a, launch_extend_dpt(dq, dpt_blocks, a->stream4));
      HIPCHK(a, hipEventRecord(a->ev_dpt_join, a->stream4));
      HIPCHK(a, launch_extend_dp(dq, cpl, dp_blocks, s));
      HIPCHK(a, hipStreamWaitEvent(s, a->ev_dpt_join, 0));
      HIPCHK(a, hipMemcpyAsync(ctl + TPRC_Q_DONE, ctl + TPRC_Q_CUR, DP_NQ * 8, hipMemcpyDeviceToDevice, s));
    }
    // what the control kernel left (capacities, rounds): one more wave-per-read launch over that list, behind the others
    HIPCHK(a, hipStreamWaitEvent(s, a->ev_join, 0));
    HIPCHK(a, hipStreamWaitEvent(s, a->ev_join3, 0));
    {
      ExtendParamsT<C> bp = ep;
      bp.skip_scan = 1;
      bp.heavy = a->t_bail.as<unsigned long long>();
      bp.heavy_count = ctl + TPRC_BAIL_CNT;
      bp.team = nullptr;
      bp.team_count = nullptr;
      bp.queue = a->t_queue2.as<unsigned int>();
      bp.wave_counters = zp.wave_counters + (uint64_t)ctl_blocks * THM_N_COUNTERS;
      HIPCHK(a, launch_extend(bp, cpl, bail_blocks, s));
    }
  } else if (team_ok) {
    ExtendParamsT<C> tp = ep;
    tp.list_only = 1;
    tp.wave_counters = ep.wave_counters + main_rows * THM_N_COUNTERS;
    tp.trace_scratch = ep.trace_scratch + main_trace_waves * trace_per_wave / 8;
    HIPCHK(a, hipEventRecord(a->ev_fork, s));
    HIPCHK(a, launch_extend(tp, cpl, a->n_cu, s, true));
    HIPCHK(a, hipStreamWaitEvent(a->stream2, a->ev_fork, 0));
    HIPCHK(a, launch_extend(ep, cpl, ext_blocks, a->stream2));
    HIPCHK(a, hipEventRecord(a->ev_join, a->stream2));
    HIPCHK(a, hipStreamWaitEvent(s, a->ev_join, 0));
  } else {
    HIPCHK(a, launch_extend(ep, cpl, ext_blocks, s));
  }
  // ---- slow class (and the fast kernel's retries) ----
  if (cls.n_slow || retry_possible) {
    const uint32_t sl_len = std::max(cls.slow_len, cls.fast_len), sl_bw = std::max(cls.slow_bw, cls.fast_bw);
    const uint64_t per_wave = slow_per_wave;
    const int blocks = (int)(slow_waves / 4);
    ep.wave_counters = a->e_wcnt.as<unsigned long long>() + (main_rows + team_rows) * THM_N_COUNTERS;
    HIPCHK(a, a->e_slow.ensure((size_t)blocks * 4 * per_wave + 256));
    ep.max_read_len = sl_len;
    ep.max_bw = sl_bw;
    ep.mk_cap = mk_cap_slow;
    ep.list_only = 1;
    ep.heavy = a->s_slow.as<unsigned long long>();
    ep.heavy_count = a->s_work_counts.as<unsigned long long>() + 5;
    ep.team = nullptr;
    ep.team_count = nullptr;
    ep.slow_scratch = a->e_slow.as<uint8_t>();
    ep.slow_scratch_per_wave = per_wave;
    HIPCHK(a, launch_extend(ep, 0, blocks, s));
  }
  HIPCHK(a, launch_counters_reduce(a->e_wcnt.as<unsigned long long>(), (uint32_t)n_rows, a->d_counters.as<unsigned long long>(), s));
  return THM_OK;
}

int enqueue_run(thm_aligner* a) {
  const uint64_t n = a->n_reads;
  hipStream_t s = a->stream;
  // counters as they stood before this attempt, so that a replay after a pool overflow does not count twice
  HIPCHK(a, hipMemcpyAsync(a->d_counters.as<uint8_t>() + THM_N_COUNTERS * 8, a->d_counters.p, THM_N_COUNTERS * 8,
                           hipMemcpyDeviceToDevice, s));
  HIPCHK(a, hipEventRecord(a->ev[0], s));
  int rc = enqueue_seed(a, (uint32_t)a->opts.min_seed_len);
  if (rc != THM_OK) return rc;
  HIPCHK(a, hipEventRecord(a->ev[1], s));

  const uint64_t cand_min = a->dbg_cand_cap ? a->dbg_cand_cap : n * 3 + 1024;
  // op bytes: about 2 L per exonic alignment (genome + transcript op streams), 384 per read at least
  const uint64_t ops_min = a->dbg_ops_cap ? a->dbg_ops_cap : std::max<uint64_t>(n * 384, a->n_bases * 3) + 65536;
  if (a->cand_cap < cand_min) a->cand_cap = cand_min;
  if (a->cand_ops_cap < ops_min) a->cand_ops_cap = ops_min;
  HIPCHK(a, a->e_cands.ensure(a->cand_cap * sizeof(Cand)));
  HIPCHK(a, a->e_order.ensure(a->cand_cap * 2 * 4));
  HIPCHK(a, a->e_heavy.ensure((a->cand_cap / 2 + 16) * 2 * 8));
  HIPCHK(a, a->e_rel.ensure(a->cand_cap * 4));
  HIPCHK(a, a->e_ops.ensure(a->cand_ops_cap + 64));
  HIPCHK(a, a->e_nalns.ensure((n + 1) * 4));
  HIPCHK(a, a->e_nalns64.ensure((n + 1) * 8));
  HIPCHK(a, a->e_opbytes.ensure((n + 1) * 8));
  HIPCHK(a, a->e_aln_off.ensure((n + 2) * 8));
  HIPCHK(a, a->e_ops_off.ensure((n + 2) * 8));
  HIPCHK(a, a->o_alns.ensure(a->cand_cap * sizeof(thm_aln)));
  HIPCHK(a, a->o_ops.ensure(a->cand_ops_cap + 64));
  HIPCHK(a, a->s_heavy.ensure((n + 1) * 8));
  HIPCHK(a, a->s_slow.ensure((n + 1) * 8));
  HIPCHK(a, a->s_team.ensure((n + 1) * 8));
  HIPCHK(a, hipMemsetAsync(a->d_queue.p, 0, thm::QUEUE_BYTES, s));

  // length classes for the current options; lists for the extend stage
  const uint32_t mk_cap_slow = std::max<uint32_t>(a->ix->max_tx_exons, 1);
  const bool retry_possible = a->ix->max_tx_exons > (uint32_t)FAST_MAX_YCLIPS + 1;
  const ExtClasses cls = classify(a, mk_cap_slow);
  a->n_slow_host = cls.n_slow;
  a->fast_max_len = cls.fast_max;
  a->slow_max_len = cls.slow_max;
  PlanParams pp;
  pp.offsets = a->r_offsets.as<uint64_t>();
  pp.read_hits = a->s_hits.as<uint64_t>();
  pp.n_reads = n;
  pp.fast_max_len = cls.fast_max;
  pp.slow_max_len = cls.slow_max;
  pp.heavy = a->s_heavy.as<unsigned long long>();
  pp.slow = a->s_slow.as<unsigned long long>();
  pp.team = a->s_team.as<unsigned long long>();
  {
    const int cpl_f = std::max(1, (int)((2 * cls.fast_bw + 1 + 63) / 64));
    pp.team_ok = (cpl_f <= 2 && team_fits_lds(extend_lds_bytes(cls.fast_len, cls.fast_bw, cpl_f))) ? 1u : 0u;
  }
  pp.total_hits = a->s_cand_off.as<uint64_t>() + n;
  {
    static const unsigned div_env = [] {
      const char* e = getenv("THM_TEAM_DIV_PER_CU");  // tuning knob
      const int v = e ? atoi(e) : 0;
      return v > 0 ? (unsigned)v : TEAM_DIV_PER_CU;
    }();
    pp.team_div = div_env * (uint32_t)std::max(a->n_cu, 1);
  }
  pp.tpr_max_hits = (a->use_tpr && std::max(1, (int)((2 * cls.fast_bw + 1 + 63) / 64)) <= 4) ? TPR_MAX_HITS : 0u;
  pp.counts = a->s_work_counts.as<unsigned long long>();
  pp.read_status = a->r_status.as<int32_t>();
  pp.read_n_alns = a->e_nalns.as<uint32_t>();
  pp.read_op_bytes = a->e_opbytes.as<uint64_t>();
  HIPCHK(a, launch_plan(pp, s));
  HIPCHK(a, hipEventRecord(a->ev[2], s));

  rc = a->dix->wide ? enqueue_extend_t<uint64_t>(a, cls, mk_cap_slow, retry_possible)
                    : enqueue_extend_t<uint32_t>(a, cls, mk_cap_slow, retry_possible);
  if (rc != THM_OK) return rc;
  HIPCHK(a, hipEventRecord(a->ev[3], s));

  HIPCHK(a, launch_widen_u32_to_u64(a->e_nalns.as<uint32_t>(), a->e_nalns64.as<uint64_t>(), n, s));
  HIPCHK(a, launch_exclusive_scan_u64(a->e_nalns64.as<uint64_t>(), a->e_aln_off.as<uint64_t>(), n,
                                      a->scan_tmp.as<uint64_t>(), s));
  HIPCHK(a, launch_exclusive_scan_u64(a->e_opbytes.as<uint64_t>(), a->e_ops_off.as<uint64_t>(), n,
                                      a->scan_tmp.as<uint64_t>(), s));
  CompactParams cp;
  cp.n_reads = n;
  cp.read_cand_off = a->s_cand_off.as<uint64_t>();
  cp.cands = a->e_cands.as<Cand>();
  cp.order = a->e_order.as<uint32_t>();
  cp.cand_ops = a->e_ops.as<uint8_t>();
  cp.read_n_alns = a->e_nalns.as<uint32_t>();
  cp.read_aln_off = a->e_aln_off.as<uint64_t>();
  cp.read_ops_off = a->e_ops_off.as<uint64_t>();
  cp.read_offsets = a->r_offsets.as<uint64_t>();
  cp.alns = a->o_alns.as<thm_aln>();
  cp.ops = a->o_ops.as<uint8_t>();
  cp.fault = a->d_fault.as<int>();
  cp.alns_cap = a->cand_cap;
  cp.ops_cap = a->cand_ops_cap;
  // (a heavy read has more than 8 alignments and a descriptor stands for 4 of them: cand_cap / 2 entries hold either list)
  cp.heavy_cap = a->cand_cap / 2 + 16;
  cp.heavy_cnt = a->d_cursors.as<unsigned long long>() + 2;  // zeroed with the cursors when the batch starts
  cp.heavy_list = a->e_heavy.as<uint64_t>();
  cp.heavy_desc = a->e_heavy.as<uint64_t>() + cp.heavy_cap;
  cp.rel = a->e_rel.as<uint32_t>();
  HIPCHK(a, launch_compact(cp, s));
  HIPCHK(a, hipEventRecord(a->ev[4], s));
  return THM_OK;
}

// host-visible status words of the last enqueue
struct RunStatus {
  int fault_seed = 0, fault_ext = 0;
  unsigned long long smem_used = 0, ops_used = 0, total_hits = 0;
};
int read_status(thm_aligner* a, RunStatus* st) {
  hipStream_t s = a->stream;
  int f[2] = {0, 0};
  unsigned long long cur[2] = {0, 0};
  HIPCHK(a, hipMemcpyAsync(f, a->d_fault.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipMemcpyAsync(cur, a->d_cursors.p, 16, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipMemcpyAsync(&st->total_hits, a->s_cand_off.as<uint64_t>() + a->n_reads, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipStreamSynchronize(s));
  st->fault_seed = f[0];
  st->fault_ext = f[1];
  st->smem_used = cur[0];
  st->ops_used = cur[1];
  return THM_OK;
}

template <class C>
int expand_mems_t(thm_aligner* a, uint64_t n) {
  ExpandParamsT<C> xp;
  xp.ix = dev_view<C>(a);
  xp.n_reads = n;
  xp.smems = a->s_smems.as<SmemT<C>>();
  xp.read_smem_off = a->s_off.as<uint64_t>();
  xp.read_smem_cnt = a->s_cnt.as<uint32_t>();
  xp.read_mem_off = a->s_cand_off.as<uint64_t>();
  xp.mems = a->o_mems.as<thm_mem>();
  HIPCHK(a, launch_expand(xp, a->stream));
  return THM_OK;
}

}  // namespace

extern "C" {

int32_t thm_batch_upload(thm_aligner* a, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads) {
  if (!a || !offsets || (!bases && n_reads && offsets[n_reads] > 0)) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  a->uploaded = a->ran = a->synced = false;
  if (n_reads >= 0xFFFFFFF0ull) return fail(a, THM_ERR_UNSUPPORTED, "more than 2^32-16 reads in one batch");
  if (offsets[0] != 0) return fail(a, THM_ERR_INVALID_ARG, "offsets[0] must be 0");
  // the lengths present in the batch (usually one or a handful): histogram by sorting the distinct ones
  uint32_t lmax = 0;
  uint64_t n_over = 0;
  a->len_hist.clear();
  {
    std::vector<uint64_t> cnt(1024, 0);
    std::vector<std::pair<uint32_t, uint64_t>> big;
    for (uint64_t i = 0; i < n_reads; i++) {
      if (offsets[i + 1] < offsets[i]) return fail(a, THM_ERR_INVALID_ARG, "read offsets are not monotone");
      const uint64_t L = offsets[i + 1] - offsets[i];
      if (L > MAX_READ_LEN) {  // the read gets its own status; the batch goes on
        n_over++;
        continue;
      }
      if (L < cnt.size())
        cnt[L]++;
      else
        big.emplace_back((uint32_t)L, 1);
      lmax = std::max<uint32_t>(lmax, (uint32_t)L);
    }
    for (uint32_t L = 0; L < cnt.size(); L++)
      if (cnt[L]) a->len_hist.emplace_back(L, cnt[L]);
    std::sort(big.begin(), big.end());
    for (const auto& b : big) {
      if (!a->len_hist.empty() && a->len_hist.back().first == b.first)
        a->len_hist.back().second++;
      else
        a->len_hist.push_back(b);
    }
  }
  a->n_over = n_over;
  a->n_reads = n_reads;
  a->n_bases = offsets[n_reads];
  a->max_read_len = lmax;
  HIPCHK(a, a->r_bases.ensure(a->n_bases + 64));
  HIPCHK(a, a->r_offsets.ensure((n_reads + 1) * 8));
  if (a->n_bases) HIPCHK(a, hipMemcpyAsync(a->r_bases.p, bases, a->n_bases, hipMemcpyHostToDevice, a->stream));
  HIPCHK(a, hipMemcpyAsync(a->r_offsets.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, a->stream));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  a->uploaded = true;
  return THM_OK;
}

int32_t thm_batch_run(thm_aligner* a) {
  if (!a) return THM_ERR_INVALID_ARG;
  if (!a->uploaded) return fail(a, THM_ERR_INVALID_ARG, "thm_batch_run before thm_batch_upload");
  HIPCHK(a, hipSetDevice(a->device));
  a->ran = false;
  a->synced = false;
  int rc = enqueue_run(a);
  if (rc == THM_OK) a->ran = true;
  return rc;
}

int32_t thm_batch_sync(thm_aligner* a) {
  if (!a) return THM_ERR_INVALID_ARG;
  if (!a->ran) return fail(a, THM_ERR_INVALID_ARG, "thm_batch_sync before thm_batch_run");
  if (a->synced) return THM_OK;
  HIPCHK(a, hipSetDevice(a->device));
  for (int attempt = 0; attempt < 6; attempt++) {
    RunStatus st;
    int rc = read_status(a, &st);
    if (rc != THM_OK) return rc;
    // a seed-pool overflow comes first: the extend kernel did not run on that attempt (its fault word means nothing)
    if (!st.fault_seed && (st.fault_ext & 2)) return fail(a, THM_ERR_INTERNAL, "extend kernel reported an internal inconsistency (fault word 0x%x)", st.fault_ext);
    const bool grow = st.fault_seed || (st.fault_ext & 1);
    if (!grow) {
      float ms = 0;
      for (int k = 0; k < 4; k++) {
        if (hipEventElapsedTime(&ms, a->ev[k], a->ev[k + 1]) == hipSuccess) a->timings[k] = ms;
      }
      if (hipEventElapsedTime(&ms, a->ev[0], a->ev[4]) == hipSuccess) a->timings[THM_T_TOTAL] = ms;
      a->synced = true;
      return THM_OK;
    }
    // grow whatever overflowed and replay (counters of the failed attempt are discarded by the caller's reset)
    if (st.fault_seed) a->smem_cap = std::max<uint64_t>(a->smem_cap * 2, st.smem_used + 1024);
    if (st.total_hits > a->cand_cap) a->cand_cap = st.total_hits + st.total_hits / 8 + 1024;
    if (st.ops_used > a->cand_ops_cap) a->cand_ops_cap = st.ops_used + st.ops_used / 2 + 65536;
    if (a->cand_cap * sizeof(Cand) > (160ull << 30))
      return fail(a, THM_ERR_OOM, "batch has %llu seed hits: candidate pool would exceed 160 GiB", st.total_hits);
    HIPCHK(a, hipMemcpyAsync(a->d_counters.p, a->d_counters.as<uint8_t>() + THM_N_COUNTERS * 8, THM_N_COUNTERS * 8,
                             hipMemcpyDeviceToDevice, a->stream));
    a->n_replays++;
    rc = enqueue_run(a);
    if (rc != THM_OK) return rc;
  }
  return fail(a, THM_ERR_INTERNAL, "pools kept overflowing after 6 attempts");
}

int32_t thm_batch_fetch(thm_aligner* a, thm_batch_view* out) {
  if (!a || !out) return THM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  int rc = thm_batch_sync(a);
  if (rc != THM_OK) return rc;
  const uint64_t n = a->n_reads;
  hipStream_t s = a->stream;
  const int k = a->r_cur ^= 1;  // the other set still backs the previous view
  HBuf& h_off = a->r_off[k];
  HBuf& h_alns = a->r_alns[k];
  HBuf& h_ops = a->r_ops[k];
  HBuf& h_stat = a->r_stat[k];
  HIPCHK(a, h_off.ensure((n + 2) * 8));
  // offsets, and behind them the op-pool size, in one round trip
  HIPCHK(a, hipMemcpyAsync(h_off.p, a->e_aln_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipMemcpyAsync(h_off.as<uint64_t>() + n + 1, a->e_ops_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
  // per-read statuses travel only when a read can have failed: reads beyond the classes are known to the host,
  // out-of-contract reads are counted by the device (the word behind the list counters)
  unsigned long long n_contract = 0;
  HIPCHK(a, hipMemcpyAsync(&n_contract, a->s_work_counts.as<unsigned long long>() + 6, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipStreamSynchronize(s));
  const uint64_t n_alns = h_off.as<uint64_t>()[n];
  const uint64_t n_ops = h_off.as<uint64_t>()[n + 1];
  HIPCHK(a, h_alns.ensure(n_alns * sizeof(thm_aln)));
  HIPCHK(a, h_ops.ensure(n_ops));
  if (n_alns) HIPCHK(a, hipMemcpyAsync(h_alns.p, a->o_alns.p, n_alns * sizeof(thm_aln), hipMemcpyDeviceToHost, s));
  if (n_ops) HIPCHK(a, hipMemcpyAsync(h_ops.p, a->o_ops.p, n_ops, hipMemcpyDeviceToHost, s));
  uint64_t n_beyond = a->n_over;
  for (const auto& lc : a->len_hist)
    if (lc.first > a->slow_max_len) n_beyond += lc.second;
  const bool any_failed = n_beyond || n_contract;
  if (any_failed) {
    HIPCHK(a, h_stat.ensure((n + 1) * 4));
    HIPCHK(a, hipMemcpyAsync(h_stat.p, a->r_status.p, n * 4, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(a, hipStreamSynchronize(s));
  out->n_reads = n;
  out->n_alns = n_alns;
  out->n_op_bytes = n_ops;
  out->read_aln_off = h_off.as<uint64_t>();
  out->alns = h_alns.as<thm_aln>();
  out->ops = h_ops.as<uint8_t>();
  out->n_failed_reads = 0;
  out->read_status = nullptr;
  if (any_failed) {
    const int32_t* st = h_stat.as<int32_t>();
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) bad += st[i] != THM_OK;
    out->n_failed_reads = bad;
    out->read_status = bad ? st : nullptr;
  }
  return THM_OK;
}

int32_t thm_align_batch(thm_aligner* a, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                        thm_batch_view* out) {
  int rc = thm_batch_upload(a, bases, offsets, n_reads);
  if (rc != THM_OK) return rc;
  rc = thm_batch_run(a);
  if (rc != THM_OK) return rc;
  return thm_batch_fetch(a, out);
}

int32_t thm_smems_batch(thm_aligner* a, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                        uint64_t min_seed_len, thm_mems_view* out) {
  if (!a || !out) return THM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  if (min_seed_len < 1 || min_seed_len > 65535) return fail(a, THM_ERR_INVALID_ARG, "min_seed_len out of range");
  int rc = thm_batch_upload(a, bases, offsets, n_reads);
  if (rc != THM_OK) return rc;
  const uint64_t n = n_reads;
  hipStream_t s = a->stream;
  RunStatus st;
  for (int attempt = 0;; attempt++) {
    if (attempt == 0)
      HIPCHK(a, hipMemcpyAsync(a->d_counters.as<uint8_t>() + THM_N_COUNTERS * 8, a->d_counters.p, THM_N_COUNTERS * 8,
                               hipMemcpyDeviceToDevice, s));
    else
      HIPCHK(a, hipMemcpyAsync(a->d_counters.p, a->d_counters.as<uint8_t>() + THM_N_COUNTERS * 8, THM_N_COUNTERS * 8,
                               hipMemcpyDeviceToDevice, s));
    rc = enqueue_seed(a, (uint32_t)min_seed_len);
    if (rc != THM_OK) return rc;
    rc = read_status(a, &st);
    if (rc != THM_OK) return rc;
    if (!st.fault_seed) break;
    if (attempt >= 6) return fail(a, THM_ERR_INTERNAL, "smem pool kept overflowing");
    a->n_replays++;
    a->smem_cap = std::max<uint64_t>(a->smem_cap * 2, st.smem_used + 1024);
  }
  a->uploaded = false;  // the seed-only pass leaves no aligned batch behind
  const uint64_t n_mems = st.total_hits;
  if (n_mems * sizeof(thm_mem) > (64ull << 30)) return fail(a, THM_ERR_OOM, "%llu seed hits in one batch", st.total_hits);
  HIPCHK(a, a->o_mems.ensure(n_mems * sizeof(thm_mem) + 64));
  rc = a->dix->wide ? expand_mems_t<uint64_t>(a, n) : expand_mems_t<uint32_t>(a, n);
  if (rc != THM_OK) return rc;
  a->h_off.assign(n + 1, 0);
  a->h_mems.resize(n_mems);
  HIPCHK(a, hipMemcpyAsync(a->h_off.data(), a->s_cand_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, s));
  if (n_mems) HIPCHK(a, hipMemcpyAsync(a->h_mems.data(), a->o_mems.p, n_mems * sizeof(thm_mem), hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipStreamSynchronize(s));
  out->n_reads = n;
  out->n_mems = n_mems;
  out->read_mem_off = a->h_off.data();
  out->mems = a->h_mems.data();
  return THM_OK;
}

}  // extern "C"
