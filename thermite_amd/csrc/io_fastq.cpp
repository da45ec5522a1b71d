// io_fastq.cpp -- FASTQ / FASTA record batcher (include/thermite_io.h; SURVEY.md
// section 8f rank 4).  Stands for needletail::parse_fastx_file + the record loop
// of align_reads_from_file (reference src/aligner.rs:51-56): record.id() is the
// whole header line without its '@' / '>', record.seq() the bases as written
// (case kept: the aligner upper-cases on the device, the writer echoes the
// original), record.qual() the quality line.  Plain or gzip input (zlib).
#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

struct thm_fastq {
  gzFile f = nullptr;
  std::string path;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  uint64_t lineno = 0;
  std::string pending;  // a FASTA header read ahead while collecting sequence lines
  bool have_pending = false;
  thm::HostBatch own;  // storage behind thm_fastq_next_batch's view
  std::string line, seq;

  bool next_line(std::string& out) {
    out.clear();
    for (;;) {
      if (pos == end) {
        if (eof) return !out.empty();
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) {
          eof = true;
          return !out.empty();
        }
        pos = 0;
        end = (size_t)n;
      }
      const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
      if (nl) {
        out.append((const char*)(buf.data() + pos), nl);
        pos = (size_t)(nl - buf.data()) + 1;
        lineno++;
        while (!out.empty() && out.back() == '\r') out.pop_back();
        return true;
      }
      out.append((const char*)(buf.data() + pos), (const char*)(buf.data() + end));
      pos = end;
    }
  }
};

namespace {
int fail(int code, const std::string& msg) {
  thm::set_global_error(msg);
  return code;
}
}  // namespace

namespace thm {

int fastq_fill(thm_fastq* r, uint64_t max_reads, HostBatch& b) {
  b.clear();
  uint64_t n = 0;
  std::string& line = r->line;
  while (n < max_reads) {
    bool got;
    if (r->have_pending) {
      line.swap(r->pending);
      r->have_pending = false;
      got = true;
    } else {
      got = r->next_line(line);
    }
    if (!got) break;
    if (line.empty()) continue;
    const std::string where = r->path + ":" + std::to_string(r->lineno);
    if (line[0] == '@') {
      b.names.insert(b.names.end(), line.begin() + 1, line.end());
      b.name_off.push_back(b.names.size());
      std::string& s = r->seq;
      if (!r->next_line(s)) return fail(THM_ERR_FORMAT, "truncated FASTQ record at " + where);
      b.bases.insert(b.bases.end(), s.begin(), s.end());
      const size_t slen = s.size();
      if (!r->next_line(line) || line.empty() || line[0] != '+')
        return fail(THM_ERR_FORMAT, "FASTQ record without a '+' line at " + where);
      if (!r->next_line(line) && slen != 0) return fail(THM_ERR_FORMAT, "truncated FASTQ record at " + where);
      if (line.size() != slen) return fail(THM_ERR_FORMAT, "FASTQ quality length differs from sequence length at " + where);
      b.quals.insert(b.quals.end(), line.begin(), line.end());
      b.offsets.push_back(b.bases.size());
      n++;
    } else if (line[0] == '>') {
      b.names.insert(b.names.end(), line.begin() + 1, line.end());
      b.name_off.push_back(b.names.size());
      // sequence lines up to the next header
      while (r->next_line(line)) {
        if (!line.empty() && line[0] == '>') {
          r->pending.swap(line);
          r->have_pending = true;
          break;
        }
        b.bases.insert(b.bases.end(), line.begin(), line.end());
      }
      b.has_quals = false;  // record.qual() is None for FASTA
      b.quals.resize(b.bases.size(), (uint8_t)'!');
      b.offsets.push_back(b.bases.size());
      n++;
    } else {
      return fail(THM_ERR_FORMAT, "expected '@' or '>' at " + where);
    }
  }
  return THM_OK;
}

}  // namespace thm

extern "C" {

int32_t thm_fastq_open(const char* path, thm_fastq** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!path) return THM_ERR_INVALID_ARG;
  gzFile f = gzopen(path, "rb");
  if (!f) return fail(THM_ERR_IO, std::string("cannot open ") + path);
  gzbuffer(f, 1 << 20);
  thm_fastq* r = new thm_fastq();
  r->f = f;
  r->path = path;
  r->buf.resize(4 << 20);
  *out = r;
  return THM_OK;
}

void thm_fastq_close(thm_fastq* r) {
  if (!r) return;
  if (r->f) gzclose(r->f);
  delete r;
}

int32_t thm_fastq_next_batch(thm_fastq* r, uint64_t max_reads, thm_read_batch* out) {
  if (!r || !out) return THM_ERR_INVALID_ARG;
  const int rc = thm::fastq_fill(r, max_reads, r->own);
  if (rc != THM_OK) return rc;
  *out = r->own.view();
  return THM_OK;
}

}  // extern "C"
