/*
 * thermite_io.h -- the callers and data formats either side of the hot path
 * (SURVEY.md section 8f, ranks 1-4): reference ingestion + index file, FASTQ
 * batcher, SAM/PAF writer and the whole-file driver.  Same rules as
 * thermite.h: `extern "C"`, plain pointers and sizes, int32 status codes,
 * nothing unwinds.  Each declaration cites the reference interface it stands
 * for (file:line relative to the reference repository root).
 *
 * None of this is on the per-read GPU path: it is host code (C++) that feeds
 * thm_batch_upload / _run / _sync / _fetch and renders their results.
 */
#ifndef THERMITE_AMD_THERMITE_IO_H
#define THERMITE_AMD_THERMITE_IO_H

#include "thermite.h"

#ifdef __cplusplus
extern "C" {
#endif

#define THM_ERR_IO (-8)     /* file cannot be opened / read / written            */
#define THM_ERR_FORMAT (-9) /* malformed FASTA / GTF / FASTQ / index file        */

/* ------------------------------------------------ reference ingestion */

/* Index::create_from_files, src/index.rs:52-223: FASTA (optionally .gz) +
 * GTF -> index.  Text layout src/index.rs:67-101; transcript / exon / gene
 * lifting src/index.rs:126-213.  GTF row semantics follow the `transcriptome`
 * crate (Cargo.lock:1272-1274; not in the reference checkout: parity unpinned,
 * the Python restatement thermite_amd/refdata.py is the cross-check). */
int32_t thm_index_create_from_files(const char* fasta_path, const char* gtf_path, thm_index** out);

/* Names for an index built from in-memory tables; the writer needs
 * Ref::name src/index.rs:392, Tx::id src/txome.rs:19, Gene::{id,name}
 * src/txome.rs:29-33.  contig_names is indexed by thm_ref::name_id. */
int32_t thm_index_set_names(thm_index* ix, const char* const* contig_names, uint32_t n_contigs,
                            const char* const* tx_ids, uint32_t n_txs, const char* const* gene_ids,
                            const char* const* gene_names, uint32_t n_genes);

/* `thermite index -o`, src/main.rs:37-43, and ThermiteAligner::new(index_path),
 * src/wrapper.rs:31-37.  Own little-endian container ("THMIDX01"): the
 * reference's .tai is a bincode dump of bio's FMD index and cannot be read
 * without Rust.  Holds the tables and the suffix array; the k-mer table and
 * the interval grids are rebuilt at load. */
int32_t thm_index_save(const thm_index* ix, const char* path);
int32_t thm_index_load(const char* path, thm_index** out);

/* read-only view of the tables an index holds (pointers live as long as it) */
typedef struct thm_tables_view {
  uint64_t n_text;
  const uint8_t* text;
  uint32_t n_refs;
  const thm_ref* refs;
  uint32_t n_txs;
  const thm_tx* txs;
  uint64_t n_exons;
  const thm_exon* exons;
  uint64_t n_tx_seq;
  const uint8_t* tx_seq;
  uint32_t n_genes;
  const thm_span* genes;
  const uint32_t* name_rank; /* per ref */
  uint32_t n_contigs;        /* 0 when no names were supplied */
} thm_tables_view;
int32_t thm_index_tables(const thm_index* ix, thm_tables_view* out);
const char* thm_index_contig_name(const thm_index* ix, uint32_t name_id); /* NULL when unknown */
const char* thm_index_tx_id(const thm_index* ix, uint32_t tx_idx);
const char* thm_index_gene_id(const thm_index* ix, uint32_t gene_idx);
const char* thm_index_gene_name(const thm_index* ix, uint32_t gene_idx);

/* ------------------------------------------------------- FASTQ batcher */

/* needletail::parse_fastx_file + the record loop of align_reads_from_file,
 * src/aligner.rs:51-56, in batches.  Plain or gzip FASTQ (FASTA records are
 * accepted too and get empty qualities). */
typedef struct thm_fastq thm_fastq;

typedef struct thm_read_batch {
  uint64_t n_reads;
  uint64_t n_bases;
  const uint8_t* bases;     /* record.seq() back to back                  */
  const uint64_t* offsets;  /* [n_reads + 1]                              */
  const uint8_t* quals;     /* record.qual(), same offsets; may be NULL   */
  const uint8_t* names;     /* record.id() (full header line) back to back */
  const uint64_t* name_off; /* [n_reads + 1]                              */
} thm_read_batch;

int32_t thm_fastq_open(const char* path, thm_fastq** out);
/* up to max_reads records; out->n_reads == 0 at end of file.  The view is
 * valid until the next call on the same reader. */
int32_t thm_fastq_next_batch(thm_fastq* r, uint64_t max_reads, thm_read_batch* out);
void thm_fastq_close(thm_fastq* r);

/* -------------------------------------------------------- SAM / PAF writer */

/* OutputFormat, src/aln_writer.rs:16-21 */
enum { THM_FMT_PAF = 0, THM_FMT_SAM = 1, THM_FMT_BAM = 2 };

typedef struct thm_text {
  const uint8_t* data;
  uint64_t len;
} thm_text;

typedef struct thm_writer thm_writer;
/* n_threads formatting threads (0 = hardware concurrency, at most 16; explicit values up to 32) */
int32_t thm_writer_create(const thm_index* ix, int32_t format, uint32_t n_threads, thm_writer** out);
void thm_writer_free(thm_writer* w);
/* build_sam_header, src/aln_writer.rs:256-276 (empty for PAF; for BAM the
 * BGZF-compressed magic + header text + reference list of bam::Writer::write_header
 * / write_reference_sequences, src/aligner.rs:41-46) */
int32_t thm_writer_header(thm_writer* w, thm_text* out);
/* what closes the file: the BGZF end-of-file block for BAM, nothing otherwise */
int32_t thm_writer_trailer(thm_writer* w, thm_text* out);
/* aln_to_sam_record / unmapped_sam_record / PafEntry, src/aln_writer.rs:47-253,
 * applied in the order of the writer loop src/aligner.rs:58-115: the records
 * of `reads` rendered from `result` (= what thm_align_batch / thm_batch_fetch
 * returned for exactly these reads).  For BAM the result is a run of complete
 * BGZF blocks (records re-encoded in binary, bam::Writer::write_sam_record,
 * src/aligner.rs:69-76,98-108).  Text valid until the next call on `w`. */
int32_t thm_writer_format_batch(thm_writer* w, const thm_read_batch* reads, const thm_batch_view* result, thm_text* out);

/* ------------------------------------------------------ whole-file driver */

typedef struct thm_run_stats {
  uint64_t n_reads;
  uint64_t n_aligned_reads;
  uint64_t n_records;
  uint64_t n_batches;
  uint64_t n_output_bytes;
  double parse_s;  /* summed over batches; stages overlap, so the sum exceeds wall_s */
  double gpu_s;    /* upload + run + sync + fetch                                    */
  double format_s;
  double write_s;
  double wall_s;
} thm_run_stats;

/* align_reads_from_file, src/aligner.rs:22-120: every record of every FASTQ in
 * order -> output_path ("-" = stdout).  Three overlapped stages (parse | GPU |
 * format+write) over batches of `batch_reads` reads (0 = 250 000). */
int32_t thm_align_files(thm_aligner* a, const char* const* fastq_paths, uint32_t n_paths, const char* output_path,
                        int32_t format, uint64_t batch_reads, uint32_t n_threads, thm_run_stats* stats);
/* The same over several aligners -- one per GPU of the node, all over one index (the shape of ThermiteAligner: Clone +
 * Send over Arc<Index>, src/wrapper.rs:20-27): batches are dealt to the aligners in input order, each aligner is
 * driven by its own host thread, parsing runs on several threads, and the records still leave in input order
 * (src/aligner.rs:54-115).  thm_align_files is this call with one aligner. */
int32_t thm_align_files_multi(thm_aligner* const* aligners, uint32_t n_aligners, const char* const* fastq_paths, uint32_t n_paths,
                              const char* output_path, int32_t format, uint64_t batch_reads, uint32_t n_threads,
                              thm_run_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
