// launch.h -- kernel parameter blocks and host-callable launch wrappers.
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
  uint32_t x_cap, y_cap;  // per-wave LDS bytes for x and y (multiples of 16)
};
size_t swg_batch_lds_bytes(const SwgBatchParams& p, int cpl);
hipError_t launch_swg_batch(const SwgBatchParams& p, int cpl, int n_blocks, hipStream_t s);
hipError_t launch_wave_prims(const int* in, int* out, hipStream_t s);

// ---- read-level pipeline ----
struct ReadBatch {
  const uint8_t* bases;     // upper-cased, sanitised reads (bytes outside ACGTN -> 0), 128 B zero padding, device
  const uint64_t* offsets;  // [n_reads+1]
  uint64_t n_reads;
};

struct SeedParams {
  DeviceIndex ix;
  ReadBatch reads;
  uint32_t min_seed_len;
  uint32_t max_read_len;       // LDS sizing
  uint32_t pos_per_read;       // probe slots per read: max_read_len - min_seed_len + 1 (>= 1) rounded up to 8; the row stride
  uint16_t* ms_end;            // [n_reads * pos_per_read] end of the longest match from this position (0: < k)
  uint32_t* ms_lo;             // its suffix-array interval
  uint32_t* ms_hi;
  unsigned long long* work_reads;   // [n_reads] reads that need more than the probe at position 0
  unsigned long long* work_cells;   // [n_reads * cells per read] (read << 16 | cell) of grid cells to probe
  unsigned long long* work_counts;  // [4] list lengths ([2]: heavy reads, [3]: select overflow), zeroed before launch
  const unsigned long long* sel_list;   // reads seed_select_kernel goes through, and how many (set by launch_seed)
  const unsigned long long* sel_count;
  unsigned long long* sel_list_out;     // seed_select_thread_kernel: reads with more SMEMs than its list holds
  unsigned long long* sel_count_out;
  Smem* smems;                 // pool
  uint64_t smem_cap;           // pool capacity (entries)
  unsigned long long* cursor;  // bump allocator head (entries), zeroed before launch
  uint64_t* read_smem_off;     // [n_reads] first entry of the read's run in the pool
  uint32_t* read_smem_cnt;     // [n_reads]
  uint64_t* read_hits;         // [n_reads] total occurrences = sum(hi-lo)
  unsigned long long* counters;
  unsigned int* queue;
  int* fault;  // 1 = smem pool overflow
};
size_t seed_lds_bytes(uint32_t max_read_len);
hipError_t launch_seed(const SeedParams& p, int n_blocks, hipStream_t s);
// list the reads with >= HEAVY_HITS hits (after the seed stage): heavy[0 .. *count)
hipError_t launch_plan_heavy(const uint64_t* read_hits, uint64_t n_reads, unsigned long long* heavy, unsigned long long* count,
                             hipStream_t s);
hipError_t launch_sanitize(const uint8_t* in, uint8_t* out, uint64_t n, uint64_t n_padded, hipStream_t s);

// expand SMEMs into Mem lists (thm_smems_batch)
struct ExpandParams {
  DeviceIndex ix;
  uint64_t n_reads;
  const Smem* smems;
  const uint64_t* read_smem_off;
  const uint32_t* read_smem_cnt;
  const uint64_t* read_mem_off;  // exclusive prefix sum of read_hits, [n_reads+1]
  thm_mem* mems;
};
hipError_t launch_expand(const ExpandParams& p, hipStream_t s);

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
constexpr size_t QUEUE_BYTES = (EXT_NQ + 1) * EXT_QSTRIDE * 4;  // + the counter of the heavy-read list
constexpr unsigned HEAVY_HITS = 8;  // reads with at least this many seed hits are scheduled first

struct ExtendParams {
  DeviceIndex ix;
  ReadBatch reads;
  thm_align_opts opts;
  const Smem* smems;
  const uint64_t* read_smem_off;
  const uint32_t* read_smem_cnt;
  const uint64_t* read_cand_off;  // exclusive prefix sum of read_hits: the read's slice of cands[]
  const unsigned long long* heavy;        // reads with >= HEAVY_HITS hits (plan_heavy_kernel)
  const unsigned long long* heavy_count;  // [1]
  Cand* cands;
  uint64_t cand_cap;  // entries in cands[] (order[] holds twice as many u32)
  uint32_t* order;  // [total hits] per-read scratch for the final ordering (indices into the read's slice)
  uint8_t* cand_ops;
  uint64_t cand_ops_cap;
  unsigned long long* ops_cursor;
  uint32_t* read_n_alns;      // [n_reads] final alignment count
  uint64_t* read_op_bytes;    // [n_reads] serialised op bytes of the final alignments
  unsigned long long* counters;
  unsigned int* queue;
  int* fault;  // bit 0: op pool overflow, bit 1: internal inconsistency, bit 2: out-of-contract (lift failure)
  const int* fault_seed;  // set by the seed kernels on an SMEM pool overflow: the SMEM runs are incomplete, nothing may be read
  uint32_t max_read_len;
  uint32_t max_bw;
  unsigned long long* trace_scratch;  // [waves in the grid * extend_trace_scratch_bytes / 8] (unused when cpl == 1)
  unsigned long long* prof;  // 16 slots of shader clocks per section (THM_PROF builds), else unused
};
size_t extend_lds_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl);
size_t extend_trace_scratch_bytes(uint32_t max_read_len, uint32_t max_bw, int cpl);
hipError_t launch_extend(const ExtendParams& p, int cpl, int n_blocks, hipStream_t s);

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
};
hipError_t launch_compact(const CompactParams& p, hipStream_t s);

hipError_t launch_widen_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t s);

}  // namespace thm
#endif
