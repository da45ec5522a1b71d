// io_fastq.cpp -- FASTQ / FASTA record batcher (include/thermite_io.h; SURVEY.md
// section 8f rank 4).  Stands for needletail::parse_fastx_file + the record loop
// of align_reads_from_file (reference src/aligner.rs:51-56): record.id() is the
// whole header line without its '@' / '>', record.seq() the bases as written
// (case kept: the aligner upper-cases on the device, the writer echoes the
// original), record.qual() the quality line.  Plain or gzip input (zlib).
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <string>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

struct thm_fastq {
  gzFile f = nullptr;  // gzip input
  int fd = -1;         // plain input: read(2) straight into the line buffer
  std::string path;
  std::vector<char> buf;  // lines are handed out as views into this buffer (no per-line copies)
  size_t pos = 0, end = 0;
  bool eof = false;
  bool io_err = false;  // a read error (not end of file): reported as THM_ERR_IO, never as a short file
  std::string io_msg;
  uint64_t lineno = 0;
  std::string pending;  // a FASTA header read ahead while collecting sequence lines
  bool have_pending = false;
  thm::HostBatch own;  // storage behind thm_fastq_next_batch's view

  // Next line without its terminator ('\n' or "\r\n") as a view valid until the next call.
  bool next_line(const char*& p, size_t& len) {
    for (;;) {
      const char* nl = (pos < end) ? (const char*)memchr(buf.data() + pos, '\n', end - pos) : nullptr;
      if (nl) {
        p = buf.data() + pos;
        len = (size_t)(nl - p);
        pos += len + 1;
        lineno++;
        while (len && p[len - 1] == '\r') len--;
        return true;
      }
      if (eof) {
        if (pos == end) return false;
        p = buf.data() + pos;
        len = end - pos;
        pos = end;
        lineno++;
        while (len && p[len - 1] == '\r') len--;
        return true;
      }
      // keep the partial line, refill behind it
      if (pos > 0) {
        memmove(buf.data(), buf.data() + pos, end - pos);
        end -= pos;
        pos = 0;
      }
      if (end == buf.size()) buf.resize(buf.size() * 2);
      const size_t room = std::min<size_t>(buf.size() - end, 1u << 30);
      const long n = f ? (long)gzread(f, buf.data() + end, (unsigned)room) : (long)read(fd, buf.data() + end, room);
      if (n > 0) end += (size_t)n;
      if (f) {
        // a corrupt stream returns -1; a truncated one a short count and then 0 with Z_BUF_ERROR
        if (n < (long)room) {
          int zerr = Z_OK;
          const char* zmsg = gzerror(f, &zerr);
          if (n < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) {
            io_err = true;
            io_msg = "gzip read error in " + path + ": " + (zmsg && *zmsg ? zmsg : "corrupt or truncated stream");
            eof = true;
          } else if (n == 0) {
            eof = true;
          }
        }
      } else if (n < 0) {
        if (errno == EINTR) continue;
        io_err = true;
        io_msg = "read error in " + path + ": " + strerror(errno);
        eof = true;
      } else if (n == 0) {
        eof = true;
      }
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

static int fastq_fill_raw(thm_fastq* r, uint64_t max_reads, HostBatch& b, size_t& nb, size_t& nq, size_t& nn);

// append without the per-call bookkeeping of vector::insert: grow geometrically, then memcpy
static inline void put(std::vector<uint8_t>& v, size_t& used, const char* p, size_t len) {
  if (used + len > v.size()) v.resize(std::max(v.size() * 2, used + len + 4096));
  memcpy(v.data() + used, p, len);
  used += len;
}

int fastq_fill(thm_fastq* r, uint64_t max_reads, HostBatch& b) {
  b.clear();
  const int rc = fastq_fill_raw(r, max_reads, b, b.nb, b.nq, b.nn);
  // a read error ends the input early; whatever the parser made of the stump, the error is what is reported
  // (needletail returns the error to align_reads_from_file, reference src/aligner.rs:52-55)
  if (r->io_err) return fail(THM_ERR_IO, r->io_msg);
  return rc;
}

static int fastq_fill_raw(thm_fastq* r, uint64_t max_reads, HostBatch& b, size_t& nb, size_t& nq, size_t& nn) {
  uint64_t n = 0;
  const char* p = nullptr;
  size_t len = 0;
  while (n < max_reads) {
    if (r->have_pending) {
      p = r->pending.data();
      len = r->pending.size();
      r->have_pending = false;
    } else if (!r->next_line(p, len)) {
      break;
    }
    if (len == 0) continue;
    if (p[0] == '@') {
      put(b.names, nn, p + 1, len - 1);
      b.name_off.push_back(nn);
      const uint64_t at = r->lineno;
      auto where = [&] { return r->path + ":" + std::to_string(at); };
      if (!r->next_line(p, len)) return fail(THM_ERR_FORMAT, "truncated FASTQ record at " + where());
      put(b.bases, nb, p, len);
      const size_t slen = len;
      if (!r->next_line(p, len) || len == 0 || p[0] != '+')
        return fail(THM_ERR_FORMAT, "FASTQ record without a '+' line at " + where());
      if (!r->next_line(p, len)) {
        if (slen != 0) return fail(THM_ERR_FORMAT, "truncated FASTQ record at " + where());
        len = 0;
      }
      if (len != slen) return fail(THM_ERR_FORMAT, "FASTQ quality length differs from sequence length at " + where());
      put(b.quals, nq, p, len);
      b.offsets.push_back(nb);
      n++;
    } else if (p[0] == '>') {
      put(b.names, nn, p + 1, len - 1);
      b.name_off.push_back(nn);
      // sequence lines up to the next header
      while (r->next_line(p, len)) {
        if (len && p[0] == '>') {
          r->pending.assign(p, len);
          r->have_pending = true;
          break;
        }
        put(b.bases, nb, p, len);
      }
      b.has_quals = false;  // record.qual() is None for FASTA
      if (nb > b.quals.size()) b.quals.resize(std::max(b.quals.size() * 2, nb + 4096));
      memset(b.quals.data() + nq, '!', nb - nq);
      nq = nb;
      b.offsets.push_back(nb);
      n++;
    } else {
      return fail(THM_ERR_FORMAT, "expected '@' or '>' at " + r->path + ":" + std::to_string(r->lineno));
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
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(THM_ERR_IO, std::string("cannot open ") + path);
  unsigned char magic[2] = {0, 0};
  const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  gzFile f = nullptr;
  if (gz) {
    f = gzdopen(fd, "rb");
    if (!f) {
      close(fd);
      return fail(THM_ERR_IO, std::string("cannot open ") + path);
    }
    gzbuffer(f, 1 << 20);
  }
  thm_fastq* r = new thm_fastq();
  r->f = f;
  r->fd = gz ? -1 : fd;
  r->path = path;
  r->buf.resize(8 << 20);
  *out = r;
  return THM_OK;
}

void thm_fastq_close(thm_fastq* r) {
  if (!r) return;
  if (r->f) gzclose(r->f);
  if (r->fd >= 0) close(r->fd);
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
