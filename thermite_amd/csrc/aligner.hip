// aligner.hip -- host side of the C ABI (include/thermite.h): device upload of
// the index, the per-aligner stream / scratch, and the batch entry points that
// stand where aligner::align_read, Index::all_smems and SwgExtend::extend stand
// in the reference (src/aligner.rs:123, src/index.rs:228, src/swg.rs:31).
//
// There is no CPU fallback: without a HIP device every compute entry point
// fails with THM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "aligner_internal.h"

namespace thm {
const char* global_error_cstr();
}
using namespace thm;

template <class T>
static hipError_t upload(DBuf& b, const std::vector<T>& v, hipStream_t s) {
  size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
  hipError_t e = b.ensure(bytes);
  if (e != hipSuccess) return e;
  if (!v.empty()) return hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
  return hipSuccess;
}

static int get_dev_copy(thm_aligner* a) {
  thm_index* ix = const_cast<thm_index*>(a->ix);
  std::lock_guard<std::mutex> g(*(std::mutex*)ix->dev_mu);
  if ((int)ix->dev.size() <= a->device) ix->dev.resize(a->device + 1, nullptr);
  if (!ix->dev[a->device]) {
    auto* d = new thm_index::DevCopy();
    d->device = a->device;
    hipStream_t s = a->stream;
    hipError_t e = hipSuccess;
    auto up = [&](auto& buf, const auto& vec) {
      if (e == hipSuccess) e = upload(buf, vec, s);
    };
    d->wide = ix->wide;
    // the text gets 16 bytes in front (left extensions are read in whole 8-byte words that may begin a few bytes before the
    // first symbol; the window staging of the wave-per-read kernels rounds its addresses down to 16 bytes)
    if (e == hipSuccess) e = d->text.ensure(ix->text.size() + 16 + 16);
    if (e == hipSuccess) e = hipMemsetAsync(d->text.p, '$', 16, s);
    if (e == hipSuccess && !ix->text.empty())
      e = hipMemcpyAsync(d->text.as<uint8_t>() + 16, ix->text.data(), ix->text.size(), hipMemcpyHostToDevice, s);
    if (ix->wide) {
      up(d->sa, ix->sa64);
      up(d->lut, ix->lut64);
    } else {
      up(d->sa, ix->sa);
      up(d->lut, ix->lut);
    }
    up(d->refs, ix->refs);
    up(d->name_rank, ix->name_rank);
    up(d->txs, ix->txs);
    up(d->exons, ix->exons);
    up(d->exon_txoff, ix->exon_txoff);
    // transcript sequences get 16 bytes of front padding: window staging reads whole 16-byte
    // granules and may start up to 15 bytes before a transcript
    if (e == hipSuccess) e = d->tx_seq.ensure(ix->tx_seq.size() + 32);
    if (e == hipSuccess) e = hipMemsetAsync(d->tx_seq.p, '$', 16, s);
    if (e == hipSuccess && !ix->tx_seq.empty())
      e = hipMemcpyAsync(d->tx_seq.as<uint8_t>() + 16, ix->tx_seq.data(), ix->tx_seq.size(), hipMemcpyHostToDevice, s);
    up(d->exon_grid_off, ix->exon_grid_off);
    up(d->gene_grid_off, ix->gene_grid_off);
    up(d->ref_bin, ix->ref_bin);
    if (ix->wide)
      up(d->ref_recs, ix->ref_recs64);
    else
      up(d->ref_recs, ix->ref_recs);
    if (ix->wide) {
      up(d->exon_grid, ix->exon_grid64);
      up(d->gene_grid, ix->gene_grid64);
    } else {
      up(d->exon_grid, ix->exon_grid);
      up(d->gene_grid, ix->gene_grid);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      free_dev_copy(d);
      return fail(a, e == hipErrorOutOfMemory ? THM_ERR_OOM : THM_ERR_HIP, "index upload failed: %s",
                  hipGetErrorString(e));
    }
    auto fill_view = [&](auto& v) {
      typedef typename std::remove_reference<decltype(v)>::type::coord_t C;
      v.text = d->text.as<uint8_t>() + 16;
      v.sa = d->sa.as<C>();
      v.lut = d->lut.as<LutEntryT<C>>();
      v.refs = d->refs.as<thm_ref>();
      v.name_rank = d->name_rank.as<uint32_t>();
      v.ref_recs = d->ref_recs.as<RefRecT<C>>();
      v.ref_bin = d->ref_bin.as<uint32_t>();
      v.txs = d->txs.as<thm_tx>();
      v.exons = d->exons.as<thm_exon>();
      v.exon_txoff = d->exon_txoff.as<uint64_t>();
      v.tx_seq = d->tx_seq.as<uint8_t>() + 16;
      v.exon_grid_off = d->exon_grid_off.as<uint32_t>();
      v.exon_grid = d->exon_grid.as<ExonEntryT<C>>();
      v.gene_grid_off = d->gene_grid_off.as<uint32_t>();
      v.gene_grid = d->gene_grid.as<GridEntryT<C>>();
      v.n = ix->n;
      v.n_refs = (uint32_t)ix->refs.size();
      v.n_txs = (uint32_t)ix->txs.size();
      v.kt = ix->kt;
      v.max_tx_exons = ix->max_tx_exons;
    };
    memset(&d->view, 0, sizeof d->view);
    memset(&d->view64, 0, sizeof d->view64);
    if (ix->wide)
      fill_view(d->view64);
    else
      fill_view(d->view);
    ix->dev[a->device] = d;
  }
  a->dix = ix->dev[a->device];
  return THM_OK;
}

int reset_queue(thm_aligner* a) {
  HIPCHK(a, hipMemsetAsync(a->d_queue.p, 0, thm::QUEUE_BYTES, a->stream));
  HIPCHK(a, hipMemsetAsync(a->d_fault.p, 0, 64, a->stream));
  return THM_OK;
}

int grid_blocks(const thm_aligner* a, uint64_t n_items, int waves_per_block, int blocks_per_cu) {
  uint64_t need = (n_items + waves_per_block - 1) / waves_per_block;
  uint64_t cap = (uint64_t)a->n_cu * blocks_per_cu;
  return (int)std::max<uint64_t>(1, std::min(need, cap));
}

extern "C" {

const char* thm_version(void) { return "thermite_amd 0.1 (gfx950)"; }

int32_t thm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void thm_index_free(thm_index* ix) {
  if (!ix) return;
  for (auto* d : ix->dev) free_dev_copy(d);
  delete (std::mutex*)ix->dev_mu;
  delete ix;
}

const char* thm_last_error(const thm_aligner* a) { return a ? a->err.c_str() : global_error_cstr(); }

int32_t thm_aligner_create(const thm_index* ix, const thm_align_opts* opts, int32_t device_id, thm_aligner** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!ix || !opts) return fail(nullptr, THM_ERR_INVALID_ARG, "thm_aligner_create: null index or opts");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, THM_ERR_NO_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(nullptr, THM_ERR_NO_DEVICE, "device id %d out of range", device_id);
  thm_aligner* a = new thm_aligner();
  a->ix = ix;
  a->device = device_id;
  a->opts = *opts;
  {
    const char* e = getenv("THM_TPR");  // 0 / 1: the problem-parallel path off / on (A/B measurements)
    if (e && e[0] == '0') a->use_tpr = false;
    if (e && e[0] == '1') a->use_tpr = true;
    e = getenv("THM_TPR_ROUNDS");
    if (e && atoi(e) >= 1 && atoi(e) <= TPR_MAX_ROUNDS) a->tpr_rounds = atoi(e);
  }
  auto bail = [&](int code) {
    thm_aligner_free(a);
    return code;
  };
  if (hipSetDevice(device_id) != hipSuccess) return bail(fail(nullptr, THM_ERR_HIP, "hipSetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) a->n_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(nullptr, THM_ERR_HIP, "hipStreamCreate failed"));
  for (auto& e : a->ev)
    if (hipEventCreate(&e) != hipSuccess) return bail(fail(nullptr, THM_ERR_HIP, "hipEventCreate failed"));
  if (hipStreamCreateWithFlags(&a->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&a->stream3, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&a->stream4, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_dpt_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_dpt_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_join3, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming) != hipSuccess)
    return bail(fail(nullptr, THM_ERR_HIP, "second stream / events: creation failed"));
  if (a->d_counters.ensure(THM_N_COUNTERS * 8 * 3) != hipSuccess || a->d_queue.ensure(thm::QUEUE_BYTES) != hipSuccess ||
      a->d_fault.ensure(64) != hipSuccess || a->d_cursors.ensure(64) != hipSuccess)
    return bail(fail(nullptr, THM_ERR_OOM, "scratch allocation failed"));
  (void)hipMemsetAsync(a->d_counters.p, 0, THM_N_COUNTERS * 8 * 3, a->stream);
  int rc = get_dev_copy(a);
  if (rc != THM_OK) return bail(rc);
  rc = thm_aligner_set_opts(a, opts);
  if (rc != THM_OK) return bail(rc);
  *out = a;
  return THM_OK;
}

void thm_aligner_free(thm_aligner* a) {
  if (!a) return;
  (void)hipSetDevice(a->device);
  if (a->stream) (void)hipStreamSynchronize(a->stream);
  if (a->stream2) (void)hipStreamSynchronize(a->stream2);
  if (a->stream3) (void)hipStreamSynchronize(a->stream3);
  if (a->stream4) (void)hipStreamSynchronize(a->stream4);
  DBuf* all[] = {&a->d_counters, &a->d_queue, &a->d_fault, &a->d_cursors, &a->b0, &a->b1, &a->b2, &a->b3, &a->b4,
                 &a->b5, &a->b6, &a->b7, &a->b8, &a->r_bases, &a->r_offsets, &a->r_san, &a->s_ms_end, &a->s_ms_lo, &a->s_ms_hi, &a->s_work_reads, &a->s_work_long, &a->s_work_cells, &a->s_work_counts, &a->s_fill_keys, &a->s_fill_perm, &a->s_fill_hist, &a->s_sel_scratch, &a->s_heavy, &a->s_slow, &a->s_team, &a->r_status, &a->e_slow, &a->e_recs, &a->e_wcnt, &a->t_memos, &a->t_recs, &a->t_dpops, &a->t_qlist, &a->t_act[0], &a->t_act[1], &a->t_ctl, &a->t_bail, &a->t_queue2, &a->t_trace, &a->t_ttrace, &a->t_hdr, &a->t_sums, &a->s_smems, &a->s_off, &a->s_cnt,
                 &a->s_hits, &a->s_cand_off, &a->scan_tmp, &a->e_cands, &a->e_heavy, &a->e_rel, &a->e_order, &a->e_ops, &a->e_nalns,
                 &a->e_nalns64, &a->e_opbytes, &a->e_aln_off, &a->e_ops_off, &a->e_trace, &a->o_alns, &a->o_ops, &a->o_mems};
  for (DBuf* b : all) b->release();
  for (int k = 0; k < 2; k++) {
    a->r_off[k].release();
    a->r_alns[k].release();
    a->r_ops[k].release();
    a->r_stat[k].release();
  }
  for (auto& e : a->ev)
    if (e) (void)hipEventDestroy(e);
  if (a->ev_fork) (void)hipEventDestroy(a->ev_fork);
  if (a->ev_join) (void)hipEventDestroy(a->ev_join);
  if (a->ev_join3) (void)hipEventDestroy(a->ev_join3);
  if (a->ev_dpt_fork) (void)hipEventDestroy(a->ev_dpt_fork);
  if (a->ev_dpt_join) (void)hipEventDestroy(a->ev_dpt_join);
  if (a->stream4) (void)hipStreamDestroy(a->stream4);
  if (a->stream3) (void)hipStreamDestroy(a->stream3);
  if (a->stream2) (void)hipStreamDestroy(a->stream2);
  if (a->stream) (void)hipStreamDestroy(a->stream);
  delete a;
}

int32_t thm_aligner_set_opts(thm_aligner* a, const thm_align_opts* o) {
  if (!a || !o) return THM_ERR_INVALID_ARG;
  if (o->min_seed_len < 1 || o->min_seed_len > 65535) return fail(a, THM_ERR_INVALID_ARG, "min_seed_len out of range");
  if (!(o->min_aln_score_percent >= 0.0f && o->min_aln_score_percent <= 1.0f))
    return fail(a, THM_ERR_INVALID_ARG, "min_aln_score_percent must be within [0,1] (src/main.rs:46-49)");
  a->opts = *o;
  return THM_OK;
}

void* thm_aligner_stream(thm_aligner* a) { return a ? (void*)a->stream : nullptr; }

const thm_index* thm_aligner_index(const thm_aligner* a) { return a ? a->ix : nullptr; }

int32_t thm_counters_get(thm_aligner* a, uint64_t out[THM_N_COUNTERS]) {
  if (!a || !out) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  HIPCHK(a, hipMemcpyAsync(out, a->d_counters.p, THM_N_COUNTERS * 8, hipMemcpyDeviceToHost, a->stream));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  return THM_OK;
}
int32_t thm_counters_reset(thm_aligner* a) {
  if (!a) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  HIPCHK(a, hipMemsetAsync(a->d_counters.p, 0, THM_N_COUNTERS * 8, a->stream));
  return THM_OK;
}
void* thm_counters_device_ptr(thm_aligner* a) { return a ? a->d_counters.p : nullptr; }

int32_t thm_timings_get(thm_aligner* a, float out[THM_N_TIMINGS]) {
  if (!a || !out) return THM_ERR_INVALID_ARG;
  memcpy(out, a->timings, sizeof(a->timings));
  return THM_OK;
}

// ------------------------------------------------- SwgExtend::extend batch
int32_t thm_swg_extend_batch(thm_aligner* a, const uint8_t* x_bases, const uint64_t* x_off, const uint8_t* y_bases,
                             const uint64_t* y_off, const uint32_t* band_width, const int32_t* x_drop,
                             uint32_t max_band_width, uint64_t n, thm_swg_view* out) {
  if (!a || !out || !x_off || !y_off || !band_width || !x_drop) return THM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  if (n == 0) return THM_OK;
  if (n >= 0xFFFFFFFFull) return fail(a, THM_ERR_UNSUPPORTED, "more than 2^32-1 problems in one call");
  HIPCHK(a, hipSetDevice(a->device));
  uint32_t bw_max = 0, x_max = 0, y_max = 0;
  std::vector<uint64_t> ops_off(n + 1, 0);
  for (uint64_t i = 0; i < n; i++) {
    if (band_width[i] > max_band_width)  // assert!, src/swg.rs:32
      return fail(a, THM_ERR_OUT_OF_CONTRACT, "problem %llu: band_width %u > max_band_width %u (src/swg.rs:32)",
                  (unsigned long long)i, band_width[i], max_band_width);
    if (x_drop[i] < (int64_t)band_width[i])
      return fail(a, THM_ERR_OUT_OF_CONTRACT, "problem %llu: x_drop < band_width is undefined in the reference",
                  (unsigned long long)i);
    uint64_t xl = x_off[i + 1] - x_off[i], yl = y_off[i + 1] - y_off[i];
    if (x_off[i + 1] < x_off[i] || y_off[i + 1] < y_off[i]) return fail(a, THM_ERR_INVALID_ARG, "offsets not monotone");
    if (xl > MAX_READ_LEN || band_width[i] > 2 * MAX_READ_LEN)
      return fail(a, THM_ERR_UNSUPPORTED, "x longer than %u or band wider than +-%u", MAX_READ_LEN, 2 * MAX_READ_LEN);
    uint64_t cols = std::min<uint64_t>(yl, xl + band_width[i] + 1);
    bw_max = std::max(bw_max, band_width[i]);
    x_max = std::max<uint32_t>(x_max, (uint32_t)xl);
    y_max = std::max<uint32_t>(y_max, (uint32_t)cols);
    ops_off[i + 1] = ops_off[i] + xl + cols + 8;
  }
  int cpl = (int)((2 * bw_max + 1 + 63) / 64);
  SwgBatchParams p;
  p.x_cap = (x_max + 15u) & ~15u;
  p.y_cap = (y_max + 15u) & ~15u;
  if (p.x_cap == 0) p.x_cap = 16;
  if (p.y_cap == 0) p.y_cap = 16;
  p.max_bw = bw_max;
  p.scratch = nullptr;
  p.scratch_per_wave = 0;
  // bands over +-127 or problems beyond the LDS budget: the any-width kernel (SwgExtend::new takes any band, src/swg.rs:17-26)
  if (cpl > 4 || swg_batch_lds_bytes(p, cpl) > EXTEND_LDS_LIMIT) cpl = 0;
  const uint64_t xb_n = x_off[n], yb_n = y_off[n], pool = ops_off[n];
  HIPCHK(a, a->b0.ensure(xb_n + 16));
  HIPCHK(a, a->b1.ensure((n + 1) * 8));
  HIPCHK(a, a->b2.ensure(yb_n + 16));
  HIPCHK(a, a->b3.ensure((n + 1) * 8));
  HIPCHK(a, a->b4.ensure(n * 4));
  HIPCHK(a, a->b5.ensure(n * 4));
  HIPCHK(a, a->b6.ensure((n + 1) * 8));
  HIPCHK(a, a->b7.ensure(pool + 16));
  HIPCHK(a, a->b8.ensure(n * sizeof(thm_swg_aln)));
  hipStream_t s = a->stream;
  if (xb_n) HIPCHK(a, hipMemcpyAsync(a->b0.p, x_bases, xb_n, hipMemcpyHostToDevice, s));
  HIPCHK(a, hipMemcpyAsync(a->b1.p, x_off, (n + 1) * 8, hipMemcpyHostToDevice, s));
  if (yb_n) HIPCHK(a, hipMemcpyAsync(a->b2.p, y_bases, yb_n, hipMemcpyHostToDevice, s));
  HIPCHK(a, hipMemcpyAsync(a->b3.p, y_off, (n + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(a, hipMemcpyAsync(a->b4.p, band_width, n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(a, hipMemcpyAsync(a->b5.p, x_drop, n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(a, hipMemcpyAsync(a->b6.p, ops_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
  int rc = reset_queue(a);
  if (rc != THM_OK) return rc;
  p.xb = a->b0.as<uint8_t>();
  p.xo = a->b1.as<uint64_t>();
  p.yb = a->b2.as<uint8_t>();
  p.yo = a->b3.as<uint64_t>();
  p.bw = a->b4.as<uint32_t>();
  p.xd = a->b5.as<int32_t>();
  p.ops_off = a->b6.as<uint64_t>();
  p.ops = a->b7.as<uint8_t>();
  p.out = a->b8.as<thm_swg_aln>();
  p.counters = a->d_counters.as<unsigned long long>();
  p.queue = a->d_queue.as<unsigned int>();
  p.fault = a->d_fault.as<int>();
  p.n = n;
  static const int bpc = [] {  // tuning knob: resident workgroups per CU
    const char* e = getenv("THM_SWG_BPC");
    const int v = e ? atoi(e) : 4;
    return (v >= 1 && v <= 8) ? v : 4;
  }();
  int n_blocks = grid_blocks(a, n, 4, bpc);
  if (cpl == 0) {
    p.scratch_per_wave = (swg_batch_scratch_bytes(p) + 255) & ~255ull;
    const uint64_t budget = 16ull << 30;
    if (p.scratch_per_wave > budget) return fail(a, THM_ERR_UNSUPPORTED, "DP trace of %llu bytes per problem exceeds the device-memory budget",
                                                 (unsigned long long)p.scratch_per_wave);
    n_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_blocks, budget / p.scratch_per_wave / 4));
    HIPCHK(a, a->e_slow.ensure((size_t)n_blocks * 4 * p.scratch_per_wave + 256));
    p.scratch = a->e_slow.as<uint8_t>();
  }
  HIPCHK(a, launch_swg_batch(p, cpl, n_blocks, s));
  std::vector<thm_swg_aln> raw(n);
  std::vector<uint8_t> pool_h(pool);
  int fault = 0;
  HIPCHK(a, hipMemcpyAsync(raw.data(), a->b8.p, n * sizeof(thm_swg_aln), hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipMemcpyAsync(pool_h.data(), a->b7.p, pool, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipMemcpyAsync(&fault, a->d_fault.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(a, hipStreamSynchronize(s));
  if (fault) return fail(a, THM_ERR_INTERNAL, "inconsistent trace in swg_batch_kernel");
  // canonical layout: op streams back to back in problem order
  a->h_swg.resize(n);
  a->h_ops.clear();
  for (uint64_t i = 0; i < n; i++) {
    thm_swg_aln r = raw[i];
    uint64_t off = a->h_ops.size();
    a->h_ops.insert(a->h_ops.end(), pool_h.begin() + r.ops_off, pool_h.begin() + r.ops_off + r.ops_len);
    r.ops_off = off;
    a->h_swg[i] = r;
  }
  out->n = n;
  out->n_op_bytes = a->h_ops.size();
  out->alns = a->h_swg.data();
  out->ops = a->h_ops.data();
  return THM_OK;
}

// tuning hook: per-section shader clocks of the extend kernel (all zero unless built with -DTHM_PROF)
int32_t thm_debug_prof_get(thm_aligner* a, uint64_t out[16], int32_t reset) {
  if (!a || !out) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  uint8_t* p = a->d_counters.as<uint8_t>() + 2 * THM_N_COUNTERS * 8;
  HIPCHK(a, hipMemcpyAsync(out, p, 16 * 8, hipMemcpyDeviceToHost, a->stream));
  if (reset) HIPCHK(a, hipMemsetAsync(p, 0, 16 * 8, a->stream));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  return THM_OK;
}

// test hook: start the next batches with these pool sizes (entries, entries, bytes; 0 = the usual heuristics) so that
// the grow-and-replay path of thm_batch_sync is exercised; returns the number of replays so far through *n_replays
int32_t thm_debug_set_pool_caps(thm_aligner* a, uint64_t smem_cap, uint64_t cand_cap, uint64_t ops_cap, uint32_t* n_replays) {
  if (!a) return THM_ERR_INVALID_ARG;
  a->dbg_smem_cap = smem_cap;
  a->dbg_cand_cap = cand_cap;
  a->dbg_ops_cap = ops_cap;
  a->smem_cap = a->cand_cap = a->cand_ops_cap = 0;
  if (n_replays) *n_replays = a->n_replays;
  return THM_OK;
}

// test / tuning hook.  flags bit 0: the problem-parallel path (kernels_tpr.hip) off -- every read takes the wave-per-read
// kernels; bit 1: on (the parity tests run both ways); bits 8..11: rounds of requests (0: keep).
// thm_debug_tpr_stats: 32 words of the last run -- [0] reads of the fast class left to the wave-per-read kernel,
// [1..15] why (1 band, 2 grid, 3 lift, 7 other, 8 rounds / requests per round, 9 candidates, 10 ops beside Match, 11 introns,
// 12 request pools, 13 window ends, 14 open targets), [16] DP requests, [17] of them narrow (thread-per-problem kernel),
// [18..21] by band class, [22] reads still waiting when the rounds ran out.
int32_t thm_debug_set_flags(thm_aligner* a, uint32_t flags) {
  if (!a) return THM_ERR_INVALID_ARG;
  if (flags & 1u) a->use_tpr = false;
  if (flags & 2u) a->use_tpr = true;
  const int r = (int)((flags >> 8) & 15u);
  if (r >= 1 && r <= TPR_MAX_ROUNDS) a->tpr_rounds = r;
  return THM_OK;
}
// test hook: the register-resident kernels pretend their class holds bands up to `max_bw` only (max_bw + 1 is stored; 0 turns
// the hook off): a read whose band is wider gets the per-read status THM_ERR_INTERNAL and no alignments -- the condition
// the kernels guard against but cannot meet while the classes are cut by read length (src/swg.rs:32 asserts it per call)
int32_t thm_debug_set_band_clip(thm_aligner* a, uint32_t max_bw_plus_1) {
  if (!a) return THM_ERR_INVALID_ARG;
  a->dbg_band_clip = max_bw_plus_1;
  return THM_OK;
}
int32_t thm_debug_tpr_stats(thm_aligner* a, uint64_t stats[32]) {
  if (!a || !stats) return THM_ERR_INVALID_ARG;
  memset(stats, 0, 256);
  if (!a->t_ctl.p) return THM_OK;
  HIPCHK(a, hipSetDevice(a->device));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  unsigned long long c[40];
  HIPCHK(a, hipMemcpy(c, a->t_ctl.p, sizeof c, hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; k++) stats[k] = c[TPRC_STATS + k];
  stats[16] = c[TPRC_REC_CUR];
  stats[17] = c[TPRC_Q_CUR];  // class 0: thread-per-problem kernel
  for (int k = 0; k < 4; k++) stats[18 + k] = c[TPRC_Q_CUR + 1 + k];
  stats[22] = c[TPRC_N_ACT + a->tpr_rounds + 1];
  return THM_OK;
}

// profiling hook (tools/calib_fetch.py): a gather of known size over the index's k-mer table, in the access patterns of the
// hot kernels, so that rocprofv3's FETCH_SIZE can be calibrated against a byte count (launch.h: launch_calib_gather).
// Returns the bytes the lanes ask for through *bytes_requested.
int32_t thm_debug_calib_gather(thm_aligner* a, int32_t pattern, uint64_t n_threads, uint64_t* bytes_requested) {
  if (!a || pattern < 0 || pattern > 2) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  const uint64_t span = a->dix->lut.cap & ~255ull;
  if (span < 4096) return fail(a, THM_ERR_INVALID_ARG, "k-mer table too small for the calibration gather");
  HIPCHK(a, launch_calib_gather(a->dix->lut.as<uint8_t>(), span, n_threads, pattern, a->d_cursors.as<unsigned long long>() + 4, a->stream));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  if (bytes_requested) *bytes_requested = n_threads * (pattern == 0 ? 8 : (pattern == 1 ? 4 : 16));
  return THM_OK;
}

// debug hook used by tests/test_gpu_swg.py: wave scan / shift primitives
int32_t thm_debug_wave_prims(thm_aligner* a, const int32_t in[64], int32_t out[384]) {
  if (!a || !in || !out) return THM_ERR_INVALID_ARG;
  HIPCHK(a, hipSetDevice(a->device));
  HIPCHK(a, a->b0.ensure(64 * 4));
  HIPCHK(a, a->b1.ensure(384 * 4));
  HIPCHK(a, hipMemcpyAsync(a->b0.p, in, 64 * 4, hipMemcpyHostToDevice, a->stream));
  HIPCHK(a, launch_wave_prims(a->b0.as<int>(), a->b1.as<int>(), a->stream));
  HIPCHK(a, hipMemcpyAsync(out, a->b1.p, 384 * 4, hipMemcpyDeviceToHost, a->stream));
  HIPCHK(a, hipStreamSynchronize(a->stream));
  return THM_OK;
}

}  // extern "C"
