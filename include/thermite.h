/*
 * thermite.h -- C ABI of the MI355X-native seed-and-extend hot path.
 *
 * This is the drop-in boundary.  Every entry point is
 * `extern "C"`, takes plain pointers / sizes / POD structs and returns an
 * int32 status; nothing unwinds across it.  It is what a Rust `extern "C"`
 * block inside the reference's `src/aligner.rs` would bind (INTEGRATION.md
 * shows that block).  Each declaration cites the reference interface it
 * replaces as `file:line` relative to the reference repository root.
 *
 * The reference aligns one read per call (src/aligner.rs:123); a GPU wants
 * batches, so the ABI is batched: `n_reads` reads per call, concatenated bases
 * plus an offsets array.  A batch of one read reproduces the reference's call
 * shape.
 *
 * All coordinates follow the reference: "concatenated" coordinates index the
 * text T = for each contig: UPPER(seq) '$' UPPER(revcomp(seq)) '$'
 * (src/index.rs:67-101); chromosome coordinates are relative to the forward
 * strand of one contig (src/aligner.rs:429-449).
 */
#ifndef THERMITE_AMD_THERMITE_H
#define THERMITE_AMD_THERMITE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define THM_OK 0
#define THM_ERR_INVALID_ARG (-1)     /* null pointer, bad size, inconsistent table */
#define THM_ERR_NO_DEVICE (-2)       /* no HIP device / device id out of range    */
#define THM_ERR_HIP (-3)             /* a HIP runtime call failed                 */
#define THM_ERR_UNSUPPORTED (-4)     /* input exceeds a documented build limit    */
#define THM_ERR_OUT_OF_CONTRACT (-5) /* input that panics in the reference        */
#define THM_ERR_OOM (-6)
#define THM_ERR_INTERNAL (-7)

/* --------------------------------------------------------- alignment ops */
/* bio::alignment::AlignmentOperation as used by src/swg.rs:2, src/txome.rs:3.
 * Serialised op stream: one byte per Match/Subst/Del/Ins; Xclip/Yclip are the
 * kind byte followed by a little-endian u32 length.  Round-trips exactly to
 * the reference's Vec<AlignmentOperation>. */
enum {
  THM_OP_MATCH = 0,
  THM_OP_SUBST = 1,
  THM_OP_DEL = 2, /* consumes y (reference) */
  THM_OP_INS = 3, /* consumes x (query)     */
  THM_OP_XCLIP = 4,
  THM_OP_YCLIP = 5 /* intron, src/txome.rs:138 */
};

/* AlnType, src/txome.rs:64-69 */
enum { THM_ALN_EXONIC = 0, THM_ALN_INTRONIC = 1, THM_ALN_INTERGENIC = 2 };

#define THM_NO_IDX 0xFFFFFFFFu

/* ----------------------------------------------------------------- structs */

/* AlignOpts, src/aligner.rs:452-464 (same five fields, same meaning). */
typedef struct thm_align_opts {
  uint64_t min_seed_len;         /* -k, default 20 (src/main.rs:115-117) */
  float min_aln_score_percent;   /* -s, default 0.66                     */
  int32_t min_aln_score;         /* default 30                           */
  uint64_t multimap_score_range; /* default 1                            */
  int32_t intron_mode;           /* bool                                 */
  int32_t reserved;
} thm_align_opts;

/* Ref, src/index.rs:391-399.  `seq` is not carried: the text holds both
 * strands, and T[start_idx..end_idx-1) of a reverse-strand Ref equals the
 * revcomp copy that Index::seq_slice (src/index.rs:304-323) materialises. */
typedef struct thm_ref {
  uint64_t start_idx;
  uint64_t end_idx; /* includes the '$' */
  uint64_t len;     /* excludes it      */
  uint32_t name_id; /* index into the caller's contig-name table */
  uint8_t strand;   /* 1 = forward */
  uint8_t pad_[3];
} thm_ref;

/* Exon, src/txome.rs:36-41 (concatenated coordinates). */
typedef struct thm_exon {
  uint64_t start;
  uint64_t end;
  uint32_t tx_idx;
  uint32_t pad_;
} thm_exon;

/* Tx, src/txome.rs:18-26.  exons[exon_begin .. exon_begin+n_exons) are in
 * transcript order (already reversed for '-' strand, src/index.rs:192-195);
 * tx_seq[seq_off .. seq_off+seq_len) is Tx::seq. */
typedef struct thm_tx {
  uint64_t exon_begin;
  uint64_t seq_off;
  uint64_t seq_len;
  uint32_t n_exons;
  uint32_t gene_idx;
  uint8_t strand;
  uint8_t pad_[7];
} thm_tx;

/* one entry of Txome::gene_intervals, src/index.rs:134,159-162,208-213 */
typedef struct thm_span {
  uint64_t start;
  uint64_t end;
} thm_span;

/* Mem, src/index.rs:383-388 */
typedef struct thm_mem {
  uint64_t ref_idx;
  uint32_t query_idx;
  uint32_t len;
} thm_mem;

/* One GenomeAlignment (src/txome.rs:54-61) = gx_aln (bio Alignment) +
 * aln_type + ref + strand + primary.  For THM_ALN_EXONIC the tx_* fields hold
 * AlnType::Exonic::tx_aln and `tx_or_gene_idx` is tx_idx; for INTRONIC it is
 * gene_idx; for INTERGENIC THM_NO_IDX.  Op streams live in the batch's op
 * pool at [ops_off, ops_off+ops_len). */
typedef struct thm_aln {
  uint64_t ystart; /* chromosome coords, forward strand (src/aligner.rs:429-449) */
  uint64_t yend;
  uint64_t ylen;
  uint64_t ops_off;
  uint64_t tx_ystart; /* transcript coords */
  uint64_t tx_yend;
  uint64_t tx_ylen;
  uint64_t tx_ops_off;
  int32_t score;
  uint32_t ref_id; /* index into thm_ref[] */
  uint32_t xstart;
  uint32_t xend;
  uint32_t xlen;
  uint32_t ops_len; /* bytes */
  uint32_t tx_or_gene_idx;
  int32_t tx_score;
  uint32_t tx_xstart;
  uint32_t tx_xend;
  uint32_t tx_ops_len; /* bytes */
  uint8_t strand;      /* Ref::strand of the hit */
  uint8_t primary;
  uint8_t aln_type;
  uint8_t pad_;
} thm_aln; /* 112 bytes */

/* Host view of one aligned batch, in input order: alignments of read r are
 * alns[read_aln_off[r] .. read_aln_off[r+1]) in the order align_read returns
 * them (src/aligner.rs:183-189). */
typedef struct thm_batch_view {
  uint64_t n_reads;
  uint64_t n_alns;
  uint64_t n_op_bytes;
  const uint64_t* read_aln_off; /* [n_reads+1] */
  const thm_aln* alns;          /* [n_alns]    */
  const uint8_t* ops;           /* [n_op_bytes] */
  /* Per-read status.  The reference aligns a read of any length with a band of any width
   * (src/swg.rs:17-26, src/aligner.rs:137-141) and panics on a few inconsistent inputs; one such read must not
   * fail the other reads of its batch, so those outcomes are reported per read: n_failed_reads counts the
   * reads whose status is not THM_OK (they have no alignments), read_status is NULL when there are none,
   * else [n_reads] of THM_OK, THM_ERR_UNSUPPORTED (read longer than 65535 bases, or a band whose DP trace
   * exceeds the device-memory budget) or THM_ERR_OUT_OF_CONTRACT (a condition that panics in the reference:
   * lift_mem_to_tx / lift_tx_to_gx, src/txome.rs:102,154). */
  uint64_t n_failed_reads;
  const int32_t* read_status;
} thm_batch_view;

/* Result of thm_smems_batch: mems of read r in Index::all_smems order. */
typedef struct thm_mems_view {
  uint64_t n_reads;
  uint64_t n_mems;
  const uint64_t* read_mem_off; /* [n_reads+1] */
  const thm_mem* mems;
} thm_mems_view;

/* Result of one SwgExtend::extend call (src/swg.rs:156-166).
 * xstart = ystart = 0 always. */
typedef struct thm_swg_aln {
  uint64_t ops_off;
  uint32_t ops_len;
  int32_t score;
  uint32_t xend;
  uint32_t yend;
} thm_swg_aln;

typedef struct thm_swg_view {
  uint64_t n;
  uint64_t n_op_bytes;
  const thm_swg_aln* alns;
  const uint8_t* ops;
} thm_swg_view;

/* Counters summed over a run; this is the vector the multi-GPU path
 * all-reduces (one ncclAllReduce of THM_N_COUNTERS u64). */
enum {
  THM_CNT_READS = 0,
  THM_CNT_ALIGNED = 1, /* reads with >= 1 alignment */
  THM_CNT_UNMAPPED = 2,
  THM_CNT_ALNS = 3,
  THM_CNT_EXONIC = 4,
  THM_CNT_INTRONIC = 5,
  THM_CNT_INTERGENIC = 6,
  THM_CNT_SMEMS = 7,
  THM_CNT_HITS = 8,      /* seed occurrences = align_seed_hit calls */
  THM_CNT_SWG_CALLS = 9, /* SwgExtend::extend calls                */
  THM_CNT_DP_CELLS = 10,
  THM_CNT_DP_COLS = 11,
  THM_CNT_OP_BYTES = 12,
  THM_CNT_WINDOW_BYTES = 13, /* window bytes of every target extended: the genome window (src/aligner.rs:212-215) and, per
                                transcript, [seed - (L + bw), seed end + L + bw + 1) -- the term of SURVEY.md 8(d)'s
                                algorithmic bytes; counted whether or not the kernel had to stage the window */
  THM_N_COUNTERS = 16
};

/* per-stage device time of the last thm_batch_run, from HIP events recorded
 * on the aligner's stream */
enum { THM_T_SEED = 0, THM_T_PLAN = 1, THM_T_EXTEND = 2, THM_T_COMPACT = 3, THM_T_TOTAL = 4, THM_N_TIMINGS = 8 };

typedef struct thm_index thm_index;
typedef struct thm_aligner thm_aligner;

/* ------------------------------------------------------------------- index */

/* Replaces the in-memory product of Index::create_from_files
 * (src/index.rs:52-223): the caller supplies the concatenated text and the
 * annotation tables already in concatenated coordinates; the library builds
 * its own search structures (suffix array, k-mer prefix table, flattened
 * interval trees) -- it cannot read the reference's bincode .tai
 * (src/main.rs:37-43).  `sa` may be NULL (then the library builds it) or a
 * valid suffix array of `text` (checked).  Interval-tree insertion order
 * follows src/index.rs:164-191 (per transcript, genomic exon order) and
 * src/index.rs:208-213 (gene index order).  `name_rank[name_id]` is the rank of
 * that contig's name in byte-wise string order (filter_overlapping sorts by
 * ref_name, src/aligner.rs:322-327); NULL means name_id order.  Immutable and
 * shareable across aligners and threads, like Arc<Index> in src/wrapper.rs:22. */
int32_t thm_index_create_in_memory(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs,
                                   const thm_tx* txs, uint32_t n_txs, const thm_exon* exons, uint64_t n_exons,
                                   const uint8_t* tx_seq, uint64_t n_tx_seq, const thm_span* genes,
                                   uint32_t n_genes, const uint32_t* name_rank, uint32_t n_names,
                                   const uint32_t* sa, thm_index** out);
/* The same with a coordinate-width choice.  The reference keeps every text position and suffix-array rank in a
 * usize (src/index.rs:364-388: divsufsort64 -> Vec<usize>; Mem{ref_idx: usize}); a GRCh38-2020-A text has
 * about 6.2 G symbols.  The library holds the index either with 32-bit positions and ranks (text below 2^31
 * symbols; the default, half the bytes) or with 64-bit ones (any text; chosen automatically from `n`, or forced
 * by THM_INDEX_WIDE in `flags` or THM_FORCE_WIDE=1 in the environment, which runs the wide code path on small
 * texts).  Results do not depend on the width.  `sa` may be NULL or a suffix array with `sa_elem_bytes` = 4 or
 * 8 bytes per entry (checked, converted to the index's width). */
#define THM_INDEX_WIDE 1u
int32_t thm_index_create_in_memory_ex(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs,
                                      const thm_tx* txs, uint32_t n_txs, const thm_exon* exons, uint64_t n_exons,
                                      const uint8_t* tx_seq, uint64_t n_tx_seq, const thm_span* genes,
                                      uint32_t n_genes, const uint32_t* name_rank, uint32_t n_names, const void* sa,
                                      uint32_t sa_elem_bytes, uint32_t flags, thm_index** out);
void thm_index_free(thm_index*);
/* number of text symbols; bytes per text position / rank inside the index (4 or 8); suffix-array pointer
 * (host copy, n entries) in the index's width: the other accessor returns NULL */
uint64_t thm_index_text_len(const thm_index*);
uint32_t thm_index_coord_bytes(const thm_index*);
const uint32_t* thm_index_suffix_array(const thm_index*);
const uint64_t* thm_index_suffix_array64(const thm_index*);
/* Index::idx_to_ref, src/index.rs:287-290: returns ref index, writes offset */
int32_t thm_index_idx_to_ref(const thm_index*, uint64_t idx, uint64_t* offset);

/* host-side index construction helper (offline; src/index.rs:103-105 uses
 * libdivsufsort): suffix array of `text` by induced sorting */
int32_t thm_build_suffix_array(const uint8_t* text, uint64_t n, uint32_t* sa_out);
/* the same with 64-bit entries, for texts of 2^31 symbols and more (src/index.rs:104: divsufsort64) */
int32_t thm_build_suffix_array64(const uint8_t* text, uint64_t n, uint64_t* sa_out);
/* the same array built on the current HIP device (prefix doubling over a radix sort: seconds where the host builder
 * takes minutes); sa_out holds n entries of elem_bytes = 4 or 8 bytes.  THM_ERR_NO_DEVICE / THM_ERR_OOM /
 * THM_ERR_UNSUPPORTED (2^32 - 2 symbols and more with 4-byte entries): use the host builders.  thm_index_create_* without a supplied
 * suffix array tries this first for texts of 4 Mi symbols and more (THM_SA_HOST=1 in the environment: never). */
int32_t thm_build_suffix_array_gpu(const uint8_t* text, uint64_t n, void* sa_out, uint32_t elem_bytes);

/* ----------------------------------------------------------------- aligner */

/* One aligner per host thread / GPU: owns a HIP stream and device scratch
 * (the analogue of one ThermiteAligner clone, src/wrapper.rs:20-27).  The
 * index is uploaded to `device_id` on first use and shared by aligners on
 * that device. */
int32_t thm_aligner_create(const thm_index*, const thm_align_opts*, int32_t device_id, thm_aligner** out);
void thm_aligner_free(thm_aligner*);
const char* thm_last_error(const thm_aligner*); /* NULL: the calling thread's last error from a call without an aligner */
int32_t thm_aligner_set_opts(thm_aligner*, const thm_align_opts*);
/* the aligner's hipStream_t, as void* (for events / ExternalStream) */
void* thm_aligner_stream(thm_aligner*);
/* the index the aligner was created over (ThermiteAligner::index, src/wrapper.rs:22) */
const thm_index* thm_aligner_index(const thm_aligner*);

/* aligner::align_read (src/aligner.rs:123-190) for a batch.  `bases` holds the
 * reads back to back, read r = bases[offsets[r] .. offsets[r+1]).  Host
 * buffers in, host view out.  Results land in two pinned host buffer sets
 * used alternately: a view stays valid until the second-next fetch on this
 * aligner (or thm_aligner_free), so a writer can render batch k while batch
 * k+1 is uploaded, run and fetched. */
int32_t thm_align_batch(thm_aligner*, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                        thm_batch_view* out);

/* The same call split so that inputs can be resident in HBM before a timed
 * region: upload (H2D), run (kernels only, asynchronous on the aligner's
 * stream), fetch (sync + D2H + view). */
int32_t thm_batch_upload(thm_aligner*, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads);
int32_t thm_batch_run(thm_aligner*);
int32_t thm_batch_sync(thm_aligner*);
int32_t thm_batch_fetch(thm_aligner*, thm_batch_view* out);

/* Index::all_smems (src/index.rs:228-255) for a batch: the seed-level parity
 * surface. */
int32_t thm_smems_batch(thm_aligner*, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                        uint64_t min_seed_len, thm_mems_view* out);

/* SwgExtend::new + SwgExtend::extend (src/swg.rs:17-26, 31-167) for a batch
 * of independent problems: the operator-level parity surface pinned by the
 * reference's known-answer tests (src/swg.rs:249-317).  Scoring is the
 * aligner's (gap_open -1, gap_extend -1, match 1, mismatch -1,
 * src/aligner.rs:140).  band_width[i] > max_band_width panics in the reference
 * (src/swg.rs:32) and x_drop[i] < band_width[i] is undefined there (SURVEY
 * Appendix A.5): both return THM_ERR_OUT_OF_CONTRACT. */
int32_t thm_swg_extend_batch(thm_aligner*, const uint8_t* x_bases, const uint64_t* x_off, const uint8_t* y_bases,
                             const uint64_t* y_off, const uint32_t* band_width, const int32_t* x_drop,
                             uint32_t max_band_width, uint64_t n, thm_swg_view* out);

/* counters accumulated since the last reset (host copy) */
int32_t thm_counters_get(thm_aligner*, uint64_t out[THM_N_COUNTERS]);
int32_t thm_counters_reset(thm_aligner*);
/* device pointer to the same THM_N_COUNTERS u64 (for an RCCL all-reduce) */
void* thm_counters_device_ptr(thm_aligner*);
/* The one collective of the multi-GPU path (one process per GPU, reads sharded,
 * index replicated): in-place sum over all ranks of the counter vector, RCCL
 * all-reduce on the aligner's stream, synchronous on return; afterwards
 * thm_counters_get returns the job-wide totals on every rank.  The communicator
 * is an RCCL communicator: rank 0 calls thm_comm_unique_id and hands the 128
 * bytes to the other ranks by any means (the reference has no multi-process
 * mode; a torch.distributed host can equally all-reduce thm_counters_device_ptr). */
#define THM_COMM_ID_BYTES 128
typedef struct thm_comm thm_comm;
int32_t thm_comm_unique_id(uint8_t out[THM_COMM_ID_BYTES]);
int32_t thm_comm_create(const uint8_t id[THM_COMM_ID_BYTES], int32_t nranks, int32_t rank, int32_t device_id,
                        thm_comm** out);
void thm_comm_free(thm_comm*);
int32_t thm_counters_allreduce(thm_aligner*, thm_comm*);
/* milliseconds per stage of the last thm_batch_run (after thm_batch_sync) */
int32_t thm_timings_get(thm_aligner*, float out[THM_N_TIMINGS]);

/* library / build info */
const char* thm_version(void);
int32_t thm_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* THERMITE_AMD_THERMITE_H */
