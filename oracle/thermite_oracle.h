/*
 * thermite_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C ABI of the CPU restatement of thermite's seed-and-extend path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (thermite_amd/) never links or calls it.
 *
 * Parity status (see oracle/README.md):
 *   - SWG extension, left/right splice, lifting, overlap filter: PINNED by the
 *     reference's own known-answer tests (tests/test_oracle_kats.py).
 *   - Seed set / seed order / interval-tree iteration order: restated from the
 *     published algorithm of the `bio` crate 0.37.1 (Cargo.lock:92-95), which
 *     is not present under /root/reference -- PARITY UNPINNED.
 *
 * Result layouts are byte-identical to include/thermite.h (thm_aln, thm_mem,
 * thm_swg_aln) so tests can compare arrays directly.
 */
#ifndef THERMITE_ORACLE_H
#define THERMITE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_swg orc_swg;
typedef struct orc_index orc_index;
typedef struct orc_result orc_result;

/* layouts mirrored from include/thermite.h */
typedef struct orc_ref {
  uint64_t start_idx, end_idx, len;
  uint32_t name_id;
  uint8_t strand;
  uint8_t pad_[3];
} orc_ref;
typedef struct orc_exon {
  uint64_t start, end;
  uint32_t tx_idx;
  uint32_t pad_;
} orc_exon;
typedef struct orc_tx {
  uint64_t exon_begin, seq_off, seq_len;
  uint32_t n_exons, gene_idx;
  uint8_t strand;
  uint8_t pad_[7];
} orc_tx;
typedef struct orc_span {
  uint64_t start, end;
} orc_span;
typedef struct orc_mem {
  uint64_t ref_idx;
  uint32_t query_idx;
  uint32_t len;
} orc_mem;
typedef struct orc_opts {
  uint64_t min_seed_len;
  float min_aln_score_percent;
  int32_t min_aln_score;
  uint64_t multimap_score_range;
  int32_t intron_mode;
  int32_t reserved;
} orc_opts;
typedef struct orc_aln {
  uint64_t ystart, yend, ylen, ops_off;
  uint64_t tx_ystart, tx_yend, tx_ylen, tx_ops_off;
  int32_t score;
  uint32_t ref_id, xstart, xend, xlen, ops_len, tx_or_gene_idx;
  int32_t tx_score;
  uint32_t tx_xstart, tx_xend, tx_ops_len;
  uint8_t strand, primary, aln_type, pad_;
} orc_aln;
typedef struct orc_swg_aln {
  uint64_t ops_off;
  uint32_t ops_len;
  int32_t score;
  uint32_t xend, yend;
} orc_swg_aln;

/* ---- SwgExtend (src/swg.rs) ---- */
orc_swg* orc_swg_new(uint64_t max_band_width, int32_t gap_open, int32_t gap_extend, int32_t match, int32_t mismatch);
void orc_swg_free(orc_swg*);
/* returns 0, or -5 if band_width > max (the reference asserts, src/swg.rs:32).
 * ops are serialised (1 byte M/S/D/I; kind + u32 for clips) into ops_buf;
 * *ops_len receives the byte count (call fails with -1 if cap is too small). */
int32_t orc_swg_extend(orc_swg*, const uint8_t* x, uint64_t xlen, const uint8_t* y, uint64_t ylen, uint64_t band_width,
                       int32_t x_drop, int32_t* score, uint64_t* xend, uint64_t* yend, uint8_t* ops_buf,
                       uint64_t ops_cap, uint64_t* ops_len);
/* number of times a phase-1 X-drop break fired since creation */
uint64_t orc_swg_phase1_breaks(const orc_swg*);
uint64_t orc_swg_cells(const orc_swg*);

/* batch form with the layout of thm_swg_extend_batch; a fresh SwgExtend with
 * max_band_width per problem.  Result accessors below. */
orc_result* orc_swg_extend_batch(const uint8_t* x_bases, const uint64_t* x_off, const uint8_t* y_bases,
                                 const uint64_t* y_off, const uint32_t* band_width, const int32_t* x_drop,
                                 uint32_t max_band_width, uint64_t n);

/* ---- extend_left_right (src/aligner.rs:352-407) ---- */
int32_t orc_extend_left_right(orc_swg*, const uint8_t* ref_seq, uint64_t ref_len, uint64_t hit_ref_idx,
                              uint64_t hit_query_idx, uint64_t hit_len, const uint8_t* read, uint64_t read_len,
                              uint64_t band_width, int32_t x_drop, int32_t* score, uint64_t* ystart, uint64_t* xstart,
                              uint64_t* yend, uint64_t* xend, uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_len);

/* ---- extend_seed_match (src/aligner.rs:410-426); mem updated in place ---- */
void orc_extend_seed_match(const uint8_t* ref_seq, uint64_t ref_len, orc_mem* hit, const uint8_t* read,
                           uint64_t read_len);

/* ---- lifting (src/txome.rs:77-160) ---- */
int32_t orc_intersect(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1);
/* returns 0 or -5 when the reference would hit unreachable!() */
int32_t orc_lift_mem_to_tx(const orc_mem* mem, const orc_exon* exons, uint64_t n_exons, orc_mem* out);
/* ops in/out serialised; returns 0, or -5 on the reference's assert_eq!(i, yend) */
int32_t orc_lift_tx_to_gx(const uint8_t* ops, uint64_t ops_len, uint64_t ystart, uint64_t yend, const orc_exon* exons,
                          uint64_t n_exons, uint64_t* out_ystart, uint64_t* out_yend, uint8_t* out_ops,
                          uint64_t out_cap, uint64_t* out_len);

/* ---- filter_overlapping (src/aligner.rs:317-349) on (name_rank, strand,
 * ystart, yend, score) tuples; writes kept input indices, returns count ---- */
uint64_t orc_filter_overlapping(const uint32_t* name_rank, const uint8_t* strand, const uint64_t* ystart,
                                const uint64_t* yend, const int32_t* score, uint64_t n, uint64_t* kept_idx);

/* ---- index ---- */
/* naive suffix array (std::sort on suffixes): small texts only */
void orc_suffix_array_naive(const uint8_t* text, uint64_t n, uint32_t* sa);
/* O(n) validity check of a suffix array; 1 = valid */
int32_t orc_suffix_array_verify(const uint8_t* text, uint64_t n, const uint32_t* sa);

/* Builds the FMD index (BWT, Less, sampled Occ, sampled SA: src/index.rs:103-111)
 * and the two AVL interval trees (src/index.rs:135,182-183,208-213).  `sa` must be
 * a valid suffix array of text (verified; NULL => naive construction).
 * ref names are compared through name_rank = rank of the contig name in byte
 * order, supplied per name_id. */
orc_index* orc_index_create(const uint8_t* text, uint64_t n, const orc_ref* refs, uint32_t n_refs, const orc_tx* txs,
                            uint32_t n_txs, const orc_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                            uint64_t n_tx_seq, const orc_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                            uint32_t n_names, const uint32_t* sa, uint32_t sa_sampling_rate,
                            uint32_t occ_sampling_rate);
/* The same with a 64-bit suffix array: any text length, like the reference's Vec<usize> (src/index.rs:103-111,
 * :364-388).  flags bit 0: take the suffix array unverified (texts of billions of symbols); bit 1: do not keep the
 * plain suffix array (the matching-statistics cross-check is not available then). */
orc_index* orc_index_create64(const uint8_t* text, uint64_t n, const orc_ref* refs, uint32_t n_refs, const orc_tx* txs,
                              uint32_t n_txs, const orc_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                              uint64_t n_tx_seq, const orc_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                              uint32_t n_names, const uint64_t* sa, uint32_t sa_rate, uint32_t occ_rate, uint32_t flags);
int32_t orc_suffix_array_verify64(const uint8_t* text, uint64_t n, const uint64_t* sa);
void orc_index_free(orc_index*);

/* Index::all_smems (src/index.rs:228-255) through the FMD index */
orc_result* orc_all_smems_batch(const orc_index*, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                                uint64_t min_seed_len);
/* same seed list from the implementation-independent definition (matching
 * statistics on the plain suffix array) -- used to cross-check the FMD walk */
orc_result* orc_all_smems_batch_ms(const orc_index*, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                                   uint64_t min_seed_len);

/* IntervalTree::find order (bio, recalled): values in yield order */
uint64_t orc_exon_tree_find(const orc_index*, uint64_t start, uint64_t end, uint32_t* out, uint64_t cap);
uint64_t orc_gene_tree_find(const orc_index*, uint64_t start, uint64_t end, uint32_t* out, uint64_t cap);

/* aligner::align_read (src/aligner.rs:123-190) over a batch, n_threads host
 * threads (reads sharded contiguously; 1 = the reference's sequential loop) */
orc_result* orc_align_batch(const orc_index*, const orc_opts*, const uint8_t* bases, const uint64_t* offsets,
                            uint64_t n_reads, uint32_t n_threads);

/* ---- result accessors ---- */
void orc_result_free(orc_result*);
uint64_t orc_result_n(const orc_result*);             /* reads / problems   */
uint64_t orc_result_n_items(const orc_result*);       /* alns / mems / swg  */
uint64_t orc_result_n_op_bytes(const orc_result*);
const uint64_t* orc_result_offsets(const orc_result*); /* [n+1] */
const orc_aln* orc_result_alns(const orc_result*);
const orc_mem* orc_result_mems(const orc_result*);
const orc_swg_aln* orc_result_swg(const orc_result*);
const uint8_t* orc_result_ops(const orc_result*);
const uint64_t* orc_result_counters(const orc_result*); /* [16], include/thermite.h THM_CNT_* */

#ifdef __cplusplus
}
#endif
#endif
