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

// thm_writer_format_batch without the final concatenation: the text of the batch is
// chunks[0] ++ chunks[1] ++ ... (one chunk per formatting thread, valid until the next call on `w`)
int writer_format_chunks(thm_writer* w, const thm_read_batch* reads, const thm_batch_view* res,
                         std::vector<const std::string*>& chunks);

}  // namespace thm
#endif
