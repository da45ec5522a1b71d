// io_internal.h -- shared by the host-side I/O sources (not part of the ABI).
#ifndef THERMITE_IO_INTERNAL_H
#define THERMITE_IO_INTERNAL_H
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/thermite_io.h"

namespace thm {

// one batch of records, owned
struct HostBatch {
  std::vector<uint8_t> bases, quals, names;  // raw storage: the first nb / nq / nn bytes are in use
  size_t nb = 0, nq = 0, nn = 0;
  std::vector<uint64_t> offsets, name_off;
  bool has_quals = true;  // false when a record came without qualities (FASTA input)
  uint64_t n_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
  void clear() {
    nb = nq = nn = 0;
    offsets.assign(1, 0);
    name_off.assign(1, 0);
    has_quals = true;
  }
  thm_read_batch view() const {
    thm_read_batch v;
    v.n_reads = n_reads();
    v.n_bases = nb;
    v.bases = bases.data();
    v.offsets = offsets.data();
    v.quals = has_quals ? quals.data() : nullptr;
    v.names = names.data();
    v.name_off = name_off.data();
    return v;
  }
};

// up to max_reads further records of the reader into `b` (cleared first)
int fastq_fill(thm_fastq* r, uint64_t max_reads, HostBatch& b);

// Parallel parsing (the whole-file driver): the reader thread only cuts the byte stream into blocks of whole
// 4-line FASTQ records, parser threads turn blocks into batches.
//   fastq_is_plain_fastq   the input starts with '@' (FASTA and anything else take the sequential parser)
//   fastq_next_raw_block   up to max_reads records as raw bytes (`raw` reused; n_lines = 4 x records, except for a
//                          truncated last record, which the block parser reports); 0 lines at the end of the input
//                          (last_block: nothing follows this block)
//   fastq_parse_block      strict 4-line records (CR LF tolerated, empty lines only at the very end of the input,
//                          i.e. at the end of the block for which last_block is set)
bool fastq_is_plain_fastq(thm_fastq* r);
int fastq_next_raw_block(thm_fastq* r, uint64_t max_reads, std::vector<char>& raw, size_t& raw_len, uint64_t& n_lines,
                         uint64_t& first_line, bool& last_block);
int fastq_parse_block(const char* p, size_t n, const std::string& path, uint64_t first_line, bool last_block, HostBatch& b,
                      std::string& err);
const std::string& fastq_path(const thm_fastq* r);

// gzip input (io_inflate.cpp): a table-driven inflate with CRC-32 / ISIZE checks per member
uint32_t crc32_fast(uint32_t crc, const uint8_t* p, size_t n);
class GzInflater {
 public:
  GzInflater();
  ~GzInflater();
  GzInflater(const GzInflater&) = delete;
  GzInflater& operator=(const GzInflater&) = delete;
  // Takes the descriptor over.  n_threads > 1 and a regular file of some size: the file is mapped and inflated by
  // that many worker threads (io_inflate.cpp, "several threads on one gzip file"); the calling thread only collects.
  void open(int fd, const std::string& path, unsigned n_threads = 1);
  // Up to `cap` further bytes of the inflated stream into dst (cap >= 1024; the call stops a few hundred bytes short
  // of cap rather than in the middle of a match).  dst[-history, 0) must hold the `history` bytes that came before
  // (min(bytes so far, 32768) of them: matches reach back there).  0: end of the input; -1: error().
  long read(uint8_t* dst, size_t cap, size_t history);
  const std::string& error() const;

 private:
  struct Impl;
  struct Par;
  Impl* p_;
  Par* par_ = nullptr;
  bool open_parallel(int fd, const std::string& path, unsigned n_threads);
  long par_read(uint8_t* dst, size_t cap);
  static long serial_read(Impl& s, uint8_t* dst, size_t cap, size_t history);
};

// DEFLATE for the BAM writer's BGZF blocks (io_deflate.cpp): one complete raw DEFLATE stream (a single final block)
// for at most 65280 input bytes; out must hold deflate_block_bound(n) bytes; returns the bytes written
struct DeflateScratch {
  uint32_t tok[0xff00 + 8];
  uint16_t head[1 << 13];
};
size_t deflate_block_bound(size_t n);
size_t deflate_block(const uint8_t* in, size_t n, uint8_t* out, DeflateScratch& sc);

// thm_writer_format_batch without the final concatenation: the text of the batch is
// chunks[0] ++ chunks[1] ++ ... (one chunk per formatting thread, valid until the next call on `w`)
int writer_format_chunks(thm_writer* w, const thm_read_batch* reads, const thm_batch_view* res,
                         std::vector<const std::string*>& chunks);

}  // namespace thm
#endif
