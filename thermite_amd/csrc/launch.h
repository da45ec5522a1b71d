// launch.h -- kernel parameter blocks and host-callable launch wrappers.
//
// Everything that holds a text position or a suffix-array rank is templated on the
// coordinate type C (uint32_t / uint64_t, see thermite_internal.h "Coordinate width");
// the launch wrappers are overloaded on it.
#ifndef THERMITE_LAUNCH_H
#define THERMITE_LAUNCH_H

#include <hip/hip_runtime.h>

#include "thermite_internal.h"

namespace thm {

struct SwgBatchParams {
  const uint8_t* xb;
  const uint64_t* xo;
  const uint8_t* yb;
  const uint64_t* yo;
  const uint32_t* bw;
  const int32_t* xd;
  const uint64_t* ops_off;  // per problem: start of its slot in the op pool
  uint8_t* ops;
  thm_swg_aln* out;
  unsigned long long* counters;  // THM_N_COUNTERS
  unsigned int* queue;           // work-queue head, zeroed before launch
  int* fault;
  uint64_t n;
  uint32_t x_cap, y_cap;  // per-wave bytes for x and y (multiples of 16)
  // cpl == 0 (band of any width, swg_extend_tiled): per-wave scratch in global memory
  uint8_t* scratch;         // [waves in the grid * scratch_per_wave]
  uint64_t scratch_per_wave;
  uint32_t max_bw;          // sizes the tiled trace and the column state
};
size_t swg_batch_lds_bytes(const SwgBatchParams& p, int cpl);
size_t swg_batch_scratch_bytes(const SwgBatchParams& p);  // per wave, cpl == 0
hipError_t launch_swg_batch(const SwgBatchParams& p, int cpl, int n_blocks, hipStream_t s);
hipError_t launch_wave_prims(const int* in, int* out, hipStream_t s);

// ---- read-level pipeline ----
struct ReadBatch {
  const uint8_t* bases;     // upper-cased, sanitised reads (bytes outside ACGTN -> 0), 128 B zero padding, device
  const uint64_t* offsets;  // [n_reads+1]
  uint64_t n_reads;
};

// Longest read the 16-bit fields of the seed stage hold (ends and lengths); longer reads get the per-read
// status THM_ERR_UNSUPPORTED and no alignments.
constexpr uint32_t MAX_READ_LEN = 65535;
// Reads up to this length take the byte-per-position paths of the seed stage ("short"); longer ones ("long")
// are listed separately so that a few long reads in a batch of short ones do not size anybody else's launch.
constexpr uint32_t SHORT_READ_MAX = 255;

// Row of read r in the per-position arrays of the seed stage (ends and intervals of the matching statistics):
// ragged, 8-slot aligned, one slot per base of the read.
__host__ __device__ inline uint64_t ms_row(uint64_t base_off, uint64_t r) { return (base_off + 8 * r) & ~7ull; }

template <class C>
struct SeedParamsT {
  DeviceIndexT<C> ix;
  ReadBatch reads;
  uint32_t min_seed_len;
  uint32_t max_len_short;  // longest read of at most SHORT_READ_MAX bases in the batch (0: none)
  uint32_t max_len_long;   // longest read above SHORT_READ_MAX (0: none), at most MAX_READ_LEN
  uint64_t n_long;         // reads above SHORT_READ_MAX (host count: sizes the launches of the long class)
  uint16_t* ms_end;        // [ms_row(n_bases, n_reads)] end of the longest match from this position (0: < k)
  C* ms_lo;                // its suffix-array interval
  C* ms_hi;
  unsigned long long* work_short;   // [n_reads] short reads that need more than the probe at position 0
  unsigned long long* work_long;    // [n_long] long reads, the same
  unsigned long long* work_cells;   // [cells] (read << 16 | cell) of grid cells to probe
  unsigned long long* work_counts;  // [8] list lengths: 0 short, 1 cells, 2 heavy, 3 select overflow, 4 long, 5 slow; zeroed before launch
  SmemT<C>* smems;             // pool
  uint64_t smem_cap;           // pool capacity (entries)
  unsigned long long* cursor;  // bump allocator head (entries), zeroed before launch
  uint64_t* read_smem_off;     // [n_reads] first entry of the read's run in the pool
  uint32_t* read_smem_cnt;     // [n_reads]
  uint64_t* read_hits;         // [n_reads] total occurrences = sum(hi-lo)
  int32_t* read_status;        // [n_reads] per-read status (0 ok), zeroed before launch
  unsigned long long* counters;
  unsigned int* queue;         // two words: one per launch of the wavefront-per-read selection kernel
  int* fault;  // 1 = smem pool overflow
  uint8_t* sel_scratch;        // global scratch of the wavefront-per-read selection when its lists do not fit LDS
  uint64_t sel_scratch_per_wave;
  // seed_fill_kernel's probes (THM_SEED_FILL): 0 = one thread per slot of the worst-case grid, 1 = a fixed grid striding
  // over the listed cells, 2 = the probes bucketed by the leading FILL_KEY_BASES bases of their k-mer first (the
  // probe-ordering experiment of DESIGN.md section 4.1: table and suffix-array reads of a wave become neighbours)
  uint32_t fill_mode;
  uint16_t* fill_keys;     // [fill slots] bucket of the probe, 0xFFFF: no probe
  uint32_t* fill_perm;     // [fill slots] probes in bucket order
  unsigned int* fill_hist; // [FILL_BUCKETS] counts, [FILL_BUCKETS] cursors, [1] total; zeroed before launch
};
constexpr int FILL_KEY_BASES = 7;
constexpr unsigned FILL_BUCKETS = 1u << (2 * FILL_KEY_BASES);  // + one bucket for k-mers with a byte outside ACGT
size_t seed_select_lds_bytes(uint32_t max_read_len);         // per workgroup (4 waves)
size_t seed_select_scratch_bytes(uint32_t max_read_len);     // per wave
constexpr size_t SEED_SELECT_LDS_LIMIT = 64 * 1024;          // above this the lists go to global scratch
hipError_t launch_seed(const SeedParamsT<uint32_t>& p, int n_blocks, hipStream_t s);
hipError_t launch_seed(const SeedParamsT<uint64_t>& p, int n_blocks, hipStream_t s);
hipError_t launch_sanitize(const uint8_t* in, uint8_t* out, uint64_t n, uint64_t n_padded, hipStream_t s);

// After the seed stage: list the reads of the fast class with >= HEAVY_HITS hits (heavy[0 .. counts[2])), the reads
// of the slow class (slow[0 .. counts[5])), and give reads beyond every class their status.
struct PlanParams {
  const uint64_t* offsets;
  const uint64_t* read_hits;
  uint64_t n_reads;
  uint32_t fast_max_len;  // reads up to this length run in the register-resident extend kernel
  uint32_t slow_max_len;  // longer ones up to this length in the any-width kernel; beyond: THM_ERR_UNSUPPORTED
  unsigned long long* heavy;
  unsigned long long* slow;
  unsigned long long* team;    // reads with >= the team threshold of hits (counts[7]) when team_ok
  uint32_t team_ok;
  const uint64_t* total_hits;  // hits of the whole batch (the last element of the scan of read_hits)
  uint32_t team_div;           // team threshold = clamp(total_hits / team_div, TEAM_MIN_HITS, TEAM_HITS); see TEAM_HITS
  uint32_t tpr_max_hits;       // > 0: the problem-parallel path takes the reads with fewer hits (kernels_tpr.hip): the heavy
                               // list starts there and so does the team list
  unsigned long long* counts;  // work_counts of the seed stage
  int32_t* read_status;
  uint32_t* read_n_alns;       // zeroed for unsupported reads
  uint64_t* read_op_bytes;
};
hipError_t launch_plan(const PlanParams& p, hipStream_t s);

// Everything the extend kernel needs to start on a read, in one 64-byte record (one scalar load instead of a chain of
// dependent ones: offsets, SMEM run, candidate slice, first SMEM, its first suffix-array entry).  Written by
// pack_reads_kernel after the seed stage and the hit-count scan.
template <class C>
struct ReadRecT {
  uint64_t base_off;   // first base of the read in the sanitised batch
  uint64_t smem_off;   // the read's SMEM run in the pool
  uint64_t cand_off;   // the read's slice of cands[] (and, doubled, of order[])
  uint32_t len;        // read length; > max length of every class: 0xFFFFFFFF
  uint32_t smem_cnt;
  uint32_t n_hits;     // min(seed hits, 2^32 - 1): size of the slice
  uint16_t qpos0, len0;  // the first SMEM of the run ...
  C lo0, hi0;
  C sa0;               // ... and its first occurrence in align_read's order, sa[hi0 - 1]
};
static_assert(sizeof(ReadRecT<uint32_t>) == 56 && sizeof(ReadRecT<uint64_t>) == 64, "ReadRec layout");
template <class C>
struct PackParamsT {
  const C* sa;
  const uint64_t* offsets;
  uint64_t n_reads;
  const SmemT<C>* smems;
  const uint64_t* read_smem_off;
  const uint32_t* read_smem_cnt;
  const uint64_t* read_cand_off;  // [n_reads + 1]
  const int* fault_seed;
  ReadRecT<C>* recs;
};
hipError_t launch_pack_reads(const PackParamsT<uint32_t>& p, hipStream_t s);
hipError_t launch_pack_reads(const PackParamsT<uint64_t>& p, hipStream_t s);

// expand SMEMs into Mem lists (thm_smems_batch)
template <class C>
struct ExpandParamsT {
  DeviceIndexT<C> ix;
  uint64_t n_reads;
  const SmemT<C>* smems;
  const uint64_t* read_smem_off;
  const uint32_t* read_smem_cnt;
  const uint64_t* read_mem_off;  // exclusive prefix sum of read_hits, [n_reads+1]
  thm_mem* mems;
};
hipError_t launch_expand(const ExpandParamsT<uint32_t>& p, hipStream_t s);
hipError_t launch_expand(const ExpandParamsT<uint64_t>& p, hipStream_t s);

// exclusive prefix sum of u64 (n entries -> n+1 entries), single launch for moderate n
hipError_t launch_exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* block_tmp,
                                     hipStream_t s);
size_t scan_tmp_entries(uint64_t n);

// candidate alignment as the extend kernel stores it (device scratch)
struct Cand {
  uint64_t ystart, yend, ylen;      // chromosome coords
  uint64_t tx_ystart, tx_yend, tx_ylen;
  uint64_t ops_off;                 // into the candidate op pool
  uint64_t tx_ops_off;
  int32_t score;
  uint32_t ref_id;
  uint32_t xstart, xend;
  uint32_t ops_len, tx_ops_len;     // bytes (serialised)
  uint32_t tx_or_gene_idx;
  int32_t tx_score;
  uint32_t tx_xstart, tx_xend;
  uint32_t name_rank;
  uint8_t strand, aln_type, primary, pad_;
};

// work counters of the extend kernel: EXT_NQ of them, EXT_QSTRIDE u32 apart (separate cache lines)
constexpr unsigned EXT_NQ = 8, EXT_QSTRIDE = 64;
constexpr size_t QUEUE_BYTES = (EXT_NQ + 3) * EXT_QSTRIDE * 4;  // + the counters of the heavy-read list, the slow list and the team list
constexpr unsigned HEAVY_HITS = 8;  // reads with at least this many seed hits are scheduled first
// Reads with very many hits are worked on by a whole workgroup (extend_kernel, TEAM): speculative chunks of hits.
// Which reads: a wave takes ~20 us per hit (a chain of dependent memory round trips) whatever else runs, the whole
// machine ~4.5 ns per hit of a batch (256 CUs) -- a read whose hits take one wave longer than the rest of the batch
// takes the machine is the tail of the launch (measured, tools/tail_diag.py: the benchmark's batches of ~815 k hits
// finish in 3.65 ms without their reads of >= 64 hits, in 3.9 - 5.0 ms with them, set by the one longest read of 190 -
// 280 hits).  So the threshold follows the batch: total_hits / (TEAM_DIV_PER_CU x #CU), ~145 hits for the benchmark's
// batches, within [TEAM_MIN_HITS, TEAM_HITS].
constexpr unsigned TEAM_HITS = 256;
constexpr unsigned TEAM_MIN_HITS = 32;
constexpr unsigned TEAM_DIV_PER_CU = 22;
constexpr int TEAM_WAVES = 16;
constexpr unsigned TEAM_MAX_HITS = 60000;   // beyond that the team's per-chunk book (16384 chunks of 4 hits, less one per SMEM) does not fit: sequential path
// intron markers one alignment can carry in the register-resident kernel (LDS); an alignment across more
// introns sends its read to the any-width kernel, whose marker list is sized by the longest transcript
constexpr int FAST_MAX_YCLIPS = 64;

template <class C>
struct ExtendParamsT {
  DeviceIndexT<C> ix;
  ReadBatch reads;
  thm_align_opts opts;
  const SmemT<C>* smems;
  const ReadRecT<C>* read_recs;   // [n_reads] (pack_reads_kernel)
  const unsigned long long* heavy;        // the list this launch goes through first: reads with >= HEAVY_HITS hits
  const unsigned long long* heavy_count;  // (fast kernel) or the reads of the slow class and the retries (any-width kernel)
  // reads with >= TEAM_HITS hits.  A workgroup per read pays while such reads are few (the tail of the launch); when
  // there are more of them than team_limit the wave-per-read kernel takes them as further heavy reads (enough of
  // them to fill the machine) and the team kernel leaves at once.  Both kernels decide from *team_count.
  const unsigned long long* team;
  const unsigned long long* team_count;
  uint32_t team_limit;
  Cand* cands;
  uint64_t cand_cap;  // entries in cands[] (order[] holds twice as many u32)
  uint32_t* order;  // [total hits] per-read scratch for the final ordering (indices into the read's slice)
  uint8_t* cand_ops;
  uint64_t cand_ops_cap;
  unsigned long long* ops_cursor;
  uint32_t* read_n_alns;      // [n_reads] final alignment count
  uint64_t* read_op_bytes;    // [n_reads] serialised op bytes of the final alignments
  int32_t* read_status;       // [n_reads]
  unsigned long long* retry;        // fast kernel: reads it could not finish (intron markers), appended to the slow list
  unsigned long long* retry_count;
  unsigned long long* n_contract;   // reads that ended with THM_ERR_OUT_OF_CONTRACT (tells the host to fetch the statuses)
  unsigned long long* counters;
  // [waves of this launch][THM_N_COUNTERS]: every wave leaves its counts in its own row (plain stores) and
  // launch_counters_reduce adds the rows to `counters` afterwards.  (Twelve atomics per wave on one cache line, 61 000 per
  // launch at ~88 M/s, all at the end of the launch when the waves leave: 0.17 ms of a 4 ms launch.)
  unsigned long long* wave_counters;
  unsigned int* queue;
  int* fault;  // bit 0: op pool overflow, bit 1: internal inconsistency
  const int* fault_seed;  // set by the seed kernels on an SMEM pool overflow: the SMEM runs are incomplete, nothing may be read
  uint32_t max_read_len;  // reads of this launch are at most this long (the fast kernel skips longer ones)
  uint32_t max_bw;
  uint32_t mk_cap;        // intron markers per alignment
  uint32_t list_only;     // 1: only the reads of the list (any-width kernel)
  uint32_t skip_scan;     // 1: the wave-per-read kernel takes its lists only (the rest of the batch went through kernels_tpr.hip)
  unsigned long long* trace_scratch;  // fast kernel: [waves in the grid * extend_trace_scratch_bytes / 8] (unused when cpl == 1)
  uint8_t* slow_scratch;              // any-width kernel: [waves in the grid * slow_scratch_per_wave]
  uint64_t slow_scratch_per_wave;
  unsigned long long* prof;  // 16 slots of shader clocks per section (THM_PROF builds), else unused
};
size_t extend_lds_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl);
int extend_waves_per_simd(int cpl, bool wide);
size_t extend_trace_scratch_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl);
size_t extend_slow_scratch_bytes(uint32_t max_read_len, uint32_t max_bw, uint32_t mk_cap);  // per wave
constexpr size_t EXTEND_LDS_LIMIT = 160 * 1024;  // gfx950: one workgroup may take the whole LDS of its CU
// does a team workgroup (TEAM_WAVES waves' buffers + the team's own static LDS: per-chunk book, round results) fit?
// lds4 = extend_lds_bytes(...) of an ordinary 4-wave workgroup
constexpr size_t TEAM_STATIC_LDS = 16384 + TEAM_WAVES * 32 + 64;  // t_nacc[TEAM_MAX_CHUNKS], t_res, t_state, t_ctl (kernels_extend.hip)
inline bool team_fits_lds(size_t lds4) { return lds4 / 4 * TEAM_WAVES + TEAM_STATIC_LDS + 256 <= EXTEND_LDS_LIMIT; }
// team = true: the workgroup-per-read variant for reads with very many hits (cpl 1 or 2 only; TEAM_WAVES waves per workgroup)
hipError_t launch_extend(const ExtendParamsT<uint32_t>& p, int cpl, int n_blocks, hipStream_t s, bool team = false);
hipError_t launch_extend(const ExtendParamsT<uint64_t>& p, int cpl, int n_blocks, hipStream_t s, bool team = false);
// ---- hit summaries (kernels_hit.hip): the part of align_seed_hit that does not depend on the state align_read carries
// from hit to hit (band, X-drop, best score), computed once per hit by a group of eight lanes ----
// What an extension looks like before any DP (swg_device.h::swg_one_mismatch_shortcut and the bound argument in
// kernels_tpr.hip).  y has min(A, |x| + bw + 1) symbols; only A is kept, the rest is the reader's state.
enum : uint16_t {
  SK_EMPTY = 0,     // |x| == 0 or no y: score 0, nothing aligned (src/swg.rs:39-55)
  SK_SINGLE = 1,    // one base that mismatches: score 0, nothing aligned
  SK_SHORTCUT = 2,  // first pair mismatches, the rest of x matches exactly, x not one repeated base, A >= |x|: Subst, Match x (|x| - 1),
                    // score |x| - 2 -- provided x_drop >= 1, which the reader checks
  SK_UNK_DEL = 3,   // needs a DP; score <= |x| - 1 (x == y[1 .. |x| + 1): one leading deletion)
  SK_UNK = 4,       // needs a DP; score <= max(|x| - 2, 0) (first pair mismatches)
  SK_UNK_EQ = 5     // needs a DP; no bound below |x| (first pair matches: not a maximal seed)
};
struct HitSide {
  uint16_t kind;
  uint16_t eq;  // transcript targets: leading y symbols equal to the genome window's side (0: other x, or none), up to min(A, A_genome, |x| + bw0 + 1)
  uint32_t A;   // symbols of the target available to the extension (clamped to 2^32 - 1)
};
// one transcript target of a hit after lift_mem_to_tx and extend_seed_match (src/txome.rs:82-103, src/aligner.rs:410-426)
struct HitTgt {
  uint32_t tx, ent;    // transcript, entry of the exon grid it was reached through
  int32_t tr;          // the lifted, exactly extended seed: transcript position ...
  uint16_t t_q, t_len; // ... read position, length
  HitSide r, l;
  uint32_t pos;        // position in exon_to_tx.find's yield order (ties go to the earlier target, src/aligner.rs:249)
};
constexpr int HIT_MAX_OPEN = 3;  // distinct targets of one hit that need a DP (more: the wave-per-read kernels take the read)
constexpr int HIT_MAX_VAR = 4;   // targets whose window ends depend on the band (see win_*)
enum : uint8_t { HF_COMPLEX = 1, HF_KNOWN = 2 };
struct HitSum {
  uint64_t hr;        // the hit: text position ...
  uint32_t ref_id;    // ... its contig copy ...
  uint16_t q, len;    // ... read position, length
  HitSide gr, gl;     // genome window: right and left extension (A: symbols up to the contig's ends)
  // Window bytes of the transcript targets (THM_CNT_WINDOW_BYTES counts [tr0 - (L + bw), tr0 + len0 + L + bw + 1) cut to
  // the transcript, with the seed as lifted, before its exact extension).  A side of a window is never cut (counted in
  // win_nl / win_nr: L + bw resp. L + bw + 1 symbols), always cut (its size is in win_fixed), or cut for some bands
  // (win_var: the symbols available, left sides first: win_nvl of them, then win_nvr right sides).
  uint32_t win_fixed;  // sum over the targets of len0 + the always-cut sides
  uint8_t win_nl, win_nr, win_nvl, win_nvr;
  uint16_t win_var[HIT_MAX_VAR];
  uint8_t n_tgt;       // targets evaluated (up to and including the first that is exact): two extend() calls each
  uint8_t n_open;
  uint8_t flags;       // HF_COMPLEX: beyond this path's capacities (why in `why`); HF_KNOWN: `known` is set
  uint8_t why;
  HitTgt known;        // the best target whose two extensions are known in closed form (first of the best, in yield order)
  HitTgt open[HIT_MAX_OPEN];  // the targets that need a DP, in yield order, without repetitions of an earlier one
  uint32_t pad_;
};
static_assert(sizeof(HitSide) == 8 && sizeof(HitTgt) == 36 && sizeof(HitSum) == 200, "hit summary layout");
// (read, q, len, hr) of every hit of the reads this path takes, at the hit's slot = its read's cand_off + ordinal
struct HitHdr {
  uint64_t hr;
  uint32_t read;  // 0xFFFFFFFF: not a hit of this path
  uint16_t q, len;
};
template <class C>
struct HitParamsT {
  DeviceIndexT<C> ix;
  ReadBatch reads;
  thm_align_opts opts;
  const SmemT<C>* smems;
  const ReadRecT<C>* read_recs;
  uint32_t max_read_len, max_hits;  // the reads of this path: fast class, fewer than max_hits hits
  HitHdr* hdr;                      // [slots]
  HitSum* sums;                     // [slots]
  const uint64_t* total_hits;       // slots in use (the last element of the scan of the hit counts)
  uint64_t slot_cap;
  const int* fault_seed;
};
hipError_t launch_hit_expand(const HitParamsT<uint32_t>& p, hipStream_t s);
hipError_t launch_hit_expand(const HitParamsT<uint64_t>& p, hipStream_t s);
hipError_t launch_hit_summaries(const HitParamsT<uint32_t>& p, int n_blocks, hipStream_t s);
hipError_t launch_hit_summaries(const HitParamsT<uint64_t>& p, int n_blocks, hipStream_t s);

// ---- extension problems as the unit of wavefront work (kernels_tpr.hip) ----
// One SwgExtend::extend call that needs a DP: written by the control kernel (thread per read), computed by the DP
// kernel (wave per record), read back by the control kernel in the next round.
constexpr int DP_MAX_EDITS = 8;  // ops other than Match one result may carry (more: the wave-per-read kernels take the read)
struct DpRec {
  // request: x[t] = x0[t * dir], the read as the extension walks it; y[t] = y0[t * dir], the target (text or transcript
  // sequence).  After the DP the same 16 bytes hold the result's ops that are not Match: n_edits entries of
  // (index << 2 | op kind), index counting from the end cell back to the seed (the traceback's order), ascending.
  const uint8_t* x0;
  const uint8_t* y0;
  uint64_t pad0_;
  uint16_t xlen, ylen, bw, xd;
  int8_t dir;         // +1: right extension, -1: left extension
  uint8_t cls;        // 0: at most DPT_SLOTS band slots hold cells (thread-per-problem kernel); else the band slots per lane the
                      // wave-per-problem kernel needs, ceil(min(2 bw + 1, xlen + 1) / 64) = 1..4
  uint16_t n_edits;   // result: ops other than Match (0xFFFF: more than DP_MAX_EDITS)
  uint32_t read;      // (diagnostics)
  // result
  int32_t score;
  uint16_t xend, yend, nops;
  uint16_t done;      // 0: requested, 1: computed
  uint32_t cells, cols;  // DP work of the result (THM_CNT_DP_CELLS / _COLS when a finished read used it)
  uint32_t pad2_;
};
static_assert(sizeof(DpRec) == 64, "DpRec layout");
constexpr int DPT_SLOTS = 16;      // band slots of the thread-per-problem DP kernel (registers of one thread)
constexpr int DP_NQ = 5;           // request queues: class 0 (narrow) and band classes 1..4
constexpr int TPR_MAX_ROUNDS = 8;  // rounds of requests one read may take (then: the wave-per-read kernel)
// Reads with this many seed hits and more stay with the workgroup-per-read / wave-per-read kernels: the control kernel
// replays a read's hits in one thread, and a thread with thousands of hits would be the tail of its launch.
constexpr unsigned TPR_MAX_HITS = 32;
// The records of a read, by round: round k asked for the extension problems of the hits from first_hit[k] on (one
// hit, or all the remaining ones), in the order the replay meets them, under the band and X-drop in force there:
// records base[k] .. base[k] + cnt[k].
struct ReadMemo {
  uint32_t base[TPR_MAX_ROUNDS];
  uint16_t cnt[TPR_MAX_ROUNDS];
  uint16_t first_hit[TPR_MAX_ROUNDS];
  uint8_t n_rounds;
  uint8_t pad_[7];
};
static_assert(sizeof(ReadMemo) == 72, "ReadMemo layout");
template <class C>
struct TprParamsT {
  ReadRecT<C>* recs_rw;   // = ExtendParamsT::read_recs, writable: a finished read gets len = 0xFFFFFFFF (skipped by every later kernel)
  ReadMemo* memos;        // [n_reads]
  DpRec* recs;
  uint64_t rec_cap;
  unsigned long long* rec_cursor;
  uint32_t* q_list;            // [DP_NQ][q_stride] record indices by class, appended to over the rounds
  uint64_t q_stride;
  unsigned long long* q_cur;   // [DP_NQ] ends of the lists
  const uint32_t* act_in;      // reads of this round (round 0: all reads, no list)
  const unsigned long long* n_act_in;
  uint32_t* act_out;           // reads that wait for results: the next round's list
  unsigned long long* n_act_out;
  unsigned long long* bail;    // reads left to the wave-per-read kernel: appended to its list (ExtendParamsT::heavy)
  unsigned long long* bail_count;
  unsigned long long* team;    // ... or, with TEAM_MIN_HITS hits and more, to the workgroup-per-read kernel's (null: none runs)
  unsigned long long* team_count;
  uint32_t max_hits;           // reads with this many hits and more are not this path's (TPR_MAX_HITS)
  uint32_t dpt_cols;           // columns the thread-per-problem DP kernel's trace holds (DpParams::tcols)
  const HitSum* sums;          // hit summaries (kernels_hit.hip), by hit slot
  uint32_t round;
  uint32_t last_round;         // 1: no DP launch follows; a read that still needs results goes to the wave-per-read kernel
  unsigned long long* stats;   // 16 words (may be null): [0] reads left to the wave-per-read kernel, [1..15] why
};
struct DpParams {
  DpRec* recs;
  const uint32_t* q_list;
  uint64_t q_stride;
  const unsigned long long* q_cur;   // [DP_NQ]
  const unsigned long long* q_done;  // [DP_NQ] ends of the lists as of the previous round
  unsigned int* work;                // work counters of this launch, one per band class, 64 bytes apart (zeroed before the run)
  int* fault;
  uint32_t x_cap, y_cap;             // per-wave LDS bytes for x and y (multiples of 16, 64 bytes of slack each)
  unsigned long long* trace_scratch; // problems of more than 64 band slots: [waves of the launch][trace_per_wave] u64
  uint64_t trace_per_wave;
  uint32_t tcols;                    // thread-per-problem kernel: columns its LDS trace holds (a class-0 problem has at most
                                     // max(L + DPT_SLOTS / 2 + 1, DPT_SLOTS + bw) columns)
};
size_t extend_dpt_lds_bytes(uint32_t tcols);  // per workgroup
hipError_t launch_extend_dpt(const DpParams& p, int n_blocks, hipStream_t s);  // class 0: one problem per THREAD
size_t extend_dp_lds_bytes(uint32_t x_cap, uint32_t y_cap);  // per workgroup (4 waves)
size_t extend_dp_trace_bytes(uint32_t y_cap, int cpl_max);   // per wave
hipError_t launch_extend_ctl(const ExtendParamsT<uint32_t>& p, const TprParamsT<uint32_t>& tp, int n_blocks, hipStream_t s);
hipError_t launch_extend_ctl(const ExtendParamsT<uint64_t>& p, const TprParamsT<uint64_t>& tp, int n_blocks, hipStream_t s);
hipError_t launch_extend_dp(const DpParams& p, int cpl_max, int n_blocks, hipStream_t s);  // all band classes up to cpl_max in one launch
// the reads of the control kernel (fast class, fewer than max_hits hits) by descending hit count -> out[0 .. *n_out)
hipError_t launch_tpr_order(const ReadRecT<uint32_t>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins, uint32_t* out,
                            unsigned long long* n_out, const int* fault_seed, hipStream_t s);
hipError_t launch_tpr_order(const ReadRecT<uint64_t>* recs, uint64_t n, uint32_t max_len, uint32_t max_hits, unsigned long long* bins, uint32_t* out,
                            unsigned long long* n_out, const int* fault_seed, hipStream_t s);
constexpr int TPR_CTL_BLOCKS_PER_CU = 4, TPR_DP_BLOCKS_PER_CU = 8;
// layout of the small control block of a run (u64 words; zeroed before the run)
enum { TPRC_REC_CUR = 0, TPRC_Q_CUR = 2 /* [DP_NQ] */, TPRC_Q_DONE = 7 /* [DP_NQ] */, TPRC_N_ACT = 12 /* [TPR_MAX_ROUNDS + 2] */, TPRC_STATS = 24 /* [16] */, TPRC_BAIL_CNT = 23,
       TPRC_BINS = 40 /* [128]: reads per hit count, cursors (tpr_order_kernel) */,
       TPRC_WORK_BYTES = 2048 /* u32 work counters, 64 bytes apart: [round][class] */ };
constexpr size_t TPRC_BYTES = TPRC_WORK_BYTES + (size_t)(TPR_MAX_ROUNDS + 1) * 4 * 64;

struct CompactParams {
  uint64_t n_reads;
  const uint64_t* read_cand_off;
  const Cand* cands;
  const uint32_t* order;
  const uint8_t* cand_ops;
  const uint32_t* read_n_alns;
  const uint64_t* read_aln_off;  // exclusive scan of read_n_alns (as u64), [n_reads+1]
  const uint64_t* read_ops_off;  // exclusive scan of read_op_bytes, [n_reads+1]
  const uint64_t* read_offsets;  // read lengths
  thm_aln* alns;
  uint8_t* ops;
  const int* fault;  // [0] seed stage, [1] extend stage: a faulted attempt is replayed by the host, nothing is compacted
  uint64_t alns_cap, ops_cap;  // entries in alns[], bytes in ops[]
  // reads with many alignments (kernels_extend.hip, "final layout"): [0] reads listed, [1] descriptors written (zero
  // when the launch starts); heavy_list / heavy_desc hold heavy_cap entries each; rel one word per alignment
  unsigned long long* heavy_cnt;
  uint64_t* heavy_list;
  uint64_t* heavy_desc;
  uint32_t* rel;
  uint64_t heavy_cap;
};
hipError_t launch_compact(const CompactParams& p, hipStream_t s);
// counters[k] += sum over rows of wave_counters[row][k]
hipError_t launch_counters_reduce(const unsigned long long* wave_counters, uint32_t n_rows, unsigned long long* counters, hipStream_t s);

hipError_t launch_calib_gather(const uint8_t* table, uint64_t span, uint64_t n_threads, int pattern, unsigned long long* sink,
                               hipStream_t s);
hipError_t launch_widen_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t s);

}  // namespace thm
#endif
