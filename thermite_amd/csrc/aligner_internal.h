// aligner_internal.h -- host-side state shared by aligner.hip and pipeline.hip.
#ifndef THERMITE_ALIGNER_INTERNAL_H
#define THERMITE_ALIGNER_INTERNAL_H
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "launch.h"
#include "thermite_internal.h"

// grow-only device buffer
struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

// grow-only pinned host buffer (results land here: D2H at full PCIe rate, no value-initialisation)
struct HBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

struct thm_index::DevCopy {
  int device = -1;
  DBuf text, sa, lut, refs, name_rank, ref_recs, ref_bin, txs, exons, exon_txoff, tx_seq, exon_grid_off, exon_grid, gene_grid_off, gene_grid;
  bool wide = false;                     // which of the two views is valid (thermite_internal.h, "Coordinate width")
  thm::DeviceIndexT<uint32_t> view;
  thm::DeviceIndexT<uint64_t> view64;
};

inline void free_dev_copy(thm_index::DevCopy* d) {
  if (!d) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(d->device);
  DBuf* all[] = {&d->text, &d->sa,         &d->lut,    &d->refs,      &d->name_rank, &d->ref_recs, &d->ref_bin, &d->txs,
                 &d->exons, &d->exon_txoff, &d->tx_seq, &d->exon_grid_off, &d->exon_grid, &d->gene_grid_off, &d->gene_grid};
  for (DBuf* b : all) b->release();
  (void)hipSetDevice(cur);
  delete d;
}

struct thm_aligner {
  const thm_index* ix = nullptr;
  thm_index::DevCopy* dix = nullptr;
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // the team kernel runs beside the wave-per-read kernel
  hipStream_t stream3 = nullptr;  // ... and both beside the rounds of the problem-parallel path
  hipEvent_t ev_join3 = nullptr;
  hipStream_t stream4 = nullptr;  // the thread-per-problem DP kernel of a round runs beside the wave-per-problem one
  hipEvent_t ev_dpt_fork = nullptr, ev_dpt_join = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  thm_align_opts opts;
  std::string err;

  DBuf d_counters, d_queue, d_fault, d_cursors;  // cursors: [0] smem pool head, [1] op pool head (u64 each)
  DBuf b0, b1, b2, b3, b4, b5, b6, b7, b8;       // operator-level scratch

  // ---- read-level pipeline ----
  DBuf r_bases, r_offsets, r_san;  // raw reads, offsets, upper-cased + sanitised copy (made by every run)
  DBuf r_status;                   // per-read status (i32), zeroed by every run
  uint64_t n_reads = 0, n_bases = 0;
  uint32_t max_read_len = 0;
  // lengths present in the batch, ascending, with their read counts (thm_batch_upload): the length classes of a
  // run (which reads the register-resident kernels take) depend on the options, which may change between runs
  std::vector<std::pair<uint32_t, uint64_t>> len_hist;
  uint64_t n_over = 0;  // reads longer than MAX_READ_LEN (per-read status THM_ERR_UNSUPPORTED)
  bool uploaded = false;
  // seeds
  DBuf s_smems, s_off, s_cnt, s_hits, s_cand_off, scan_tmp, s_ms_end, s_ms_lo, s_ms_hi, s_work_reads, s_work_long, s_work_cells,
      s_work_counts, s_sel_scratch, s_heavy, s_slow, s_team, s_fill_keys, s_fill_perm, s_fill_hist;
  uint64_t smem_cap = 0;
  // extension
  DBuf e_heavy, e_rel;  // compact stage: lists of reads with many alignments, op offsets of their alignments
  DBuf e_cands, e_order, e_ops, e_nalns, e_nalns64, e_opbytes, e_aln_off, e_ops_off, e_trace, e_slow, e_recs, e_wcnt;
  // Extension problems as the unit of wavefront work (kernels_tpr.hip: thread-per-read control kernel + wave-per-request
  // DP kernel, in rounds), ahead of the wave-per-read kernels, which take what is left.  THM_TPR=0 or
  // thm_debug_set_flags turn it off (every read then takes the wave-per-read path); THM_TPR_ROUNDS = 1..8.
  DBuf t_memos, t_recs, t_dpops, t_qlist, t_act[2], t_ctl, t_bail, t_queue2, t_trace, t_ttrace, t_hdr, t_sums;
  bool use_tpr = false;  // (until the path is the faster one on the headline workload)
  int tpr_rounds = 8;
  uint64_t n_slow_host = 0;     // reads of the slow class in the last enqueue (host count)
  uint32_t fast_max_len = 0, slow_max_len = 0;
  uint64_t cand_cap = 0, cand_ops_cap = 0;
  // compacted outputs
  DBuf o_alns, o_ops, o_mems;
  uint64_t out_alns_cap = 0, out_ops_cap = 0;
  // test hook (thm_debug_set_pool_caps): initial pool sizes instead of the heuristics, to force the grow-and-replay path
  uint64_t dbg_smem_cap = 0, dbg_cand_cap = 0, dbg_ops_cap = 0;
  uint32_t dbg_band_clip = 0;  // test hook (thm_debug_set_band_clip): pretend the fast class holds bands up to this only (0: off)
  uint32_t n_replays = 0;  // pool-overflow replays since the aligner was created
  bool ran = false, synced = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float timings[THM_N_TIMINGS] = {0};

  // host results of the read-level path: two pinned sets used alternately, so that the view
  // thm_batch_fetch returned stays valid while the next batch is uploaded, run and fetched
  HBuf r_off[2], r_alns[2], r_ops[2], r_stat[2];
  int r_cur = 0;
  // host results of the operator- and seed-level calls
  std::vector<uint64_t> h_off;
  std::vector<thm_aln> h_alns;
  std::vector<uint8_t> h_ops;
  std::vector<thm_mem> h_mems;
  std::vector<thm_swg_aln> h_swg;
};

inline int fail(thm_aligner* a, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (a) a->err = buf;
  thm::set_global_error(buf);
  return code;
}

#define HIPCHK(a, call)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(a, e_ == hipErrorOutOfMemory ? THM_ERR_OOM : THM_ERR_HIP, "%s failed: %s (%s:%d)", \
                  #call, hipGetErrorString(e_), __FILE__, __LINE__);                                \
  } while (0)

int reset_queue(thm_aligner* a);
int grid_blocks(const thm_aligner* a, uint64_t n_items, int waves_per_block, int blocks_per_cu);
#endif
