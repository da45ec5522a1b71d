// io_fastq.cpp -- FASTQ / FASTA record batcher (include/thermite_io.h; SURVEY.md
// section 8f rank 4).  Stands for needletail::parse_fastx_file + the record loop
// of align_reads_from_file (reference src/aligner.rs:51-56): record.id() is the
// whole header line without its '@' / '>', record.seq() the bases as written
// (case kept: the aligner upper-cases on the device, the writer echoes the
// original), record.qual() the quality line.  Plain or gzip input (io_inflate.cpp).
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

struct thm_fastq {
  thm::GzInflater* z = nullptr;  // gzip input (io_inflate.cpp)
  int fd = -1;                   // plain input: read(2) straight into the line buffer
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

  // More bytes behind [pos, end); false when none came (end of the input, or an error: io_err).  The unread tail
  // moves to the front of the buffer -- for gzip input together with the 32 KiB before `end`, which the matches of
  // the bytes to come may reach back into.
  bool fill_more() {
    if (eof) return false;
    constexpr size_t WINDOW = 32768;
    const size_t keep_from = z ? std::min(pos, end > WINDOW ? end - WINDOW : (size_t)0) : pos;
    if (keep_from > 0) {
      memmove(buf.data(), buf.data() + keep_from, end - keep_from);
      pos -= keep_from;
      end -= keep_from;
    }
    if (buf.size() - end < (1u << 16)) buf.resize(buf.size() * 2);
    const size_t room = std::min<size_t>(buf.size() - end, 1u << 30);
    long n;
    if (z) {
      // a corrupt or truncated stream is an error of the inflater, never a short count
      n = z->read((uint8_t*)buf.data() + end, room, std::min(end, WINDOW));
      if (n < 0) {
        io_err = true;
        io_msg = z->error();
        eof = true;
      } else if (n == 0) {
        eof = true;
      } else if (!z->error().empty()) {  // the bytes inflated before an error come first; the error is already known
        io_err = true;
        io_msg = z->error();
        eof = true;
      }
    } else {
      do n = (long)read(fd, buf.data() + end, room);
      while (n < 0 && errno == EINTR);
      if (n < 0) {
        io_err = true;
        io_msg = "read error in " + path + ": " + strerror(errno);
        eof = true;
      } else if (n == 0) {
        eof = true;
      }
    }
    if (n > 0) end += (size_t)n;
    return n > 0;
  }

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
      fill_more();  // keeps the partial line, reads behind it
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
    if (len == 0) {
      // empty lines are tolerated at the very end of the input only (the block parser of the parallel driver applies
      // the same rule, so that both public paths accept the same files)
      bool more = false;
      while (r->next_line(p, len))
        if (len) {
          more = true;
          break;
        }
      if (!more) break;
      return fail(THM_ERR_FORMAT, "empty line inside the input before " + r->path + ":" + std::to_string(r->lineno));
    }
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

// ---- block cutting and block parsing for the parallel driver ----
const std::string& fastq_path(const thm_fastq* r) { return r->path; }

static bool refill(thm_fastq* r) { return r->fill_more(); }  // more bytes behind [pos, end); false at the end of the input

bool fastq_is_plain_fastq(thm_fastq* r) {
  while (r->pos == r->end && refill(r)) {
  }
  return r->pos < r->end && r->buf[r->pos] == '@' && !r->have_pending;
}

static inline size_t count_newlines(const char* p, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; i++) c += p[i] == '\n';  // vectorised by the compiler
  return c;
}

int fastq_next_raw_block(thm_fastq* r, uint64_t max_reads, std::vector<char>& raw, size_t& raw_len, uint64_t& n_lines,
                         uint64_t& first_line, bool& last_block) {
  const uint64_t want = max_reads * 4;
  raw_len = 0;
  n_lines = 0;
  first_line = r->lineno + 1;
  for (;;) {
    if (r->pos == r->end && !refill(r)) break;
    const char* p = r->buf.data() + r->pos;
    size_t avail = r->end - r->pos;
    size_t take = avail;
    const size_t nl = count_newlines(p, avail);
    uint64_t got = nl;
    if (n_lines + nl >= want) {  // the block ends inside this span: find its last newline
      uint64_t need = want - n_lines;
      const char* q = p;
      while (need) {
        q = (const char*)memchr(q, '\n', (size_t)(p + avail - q)) + 1;
        need--;
      }
      take = (size_t)(q - p);
      got = want - n_lines;
    }
    if (raw_len + take > raw.size()) raw.resize(std::max(raw.size() * 2, raw_len + take + 4096));
    memcpy(raw.data() + raw_len, p, take);
    raw_len += take;
    r->pos += take;
    n_lines += got;
    if (n_lines >= want) break;
  }
  if (raw_len && raw[raw_len - 1] != '\n') n_lines++;  // a last line without its newline
  r->lineno += n_lines;
  while (r->pos == r->end && refill(r)) {
  }
  last_block = r->pos == r->end;  // nothing behind this block
  if (r->io_err) return fail(THM_ERR_IO, r->io_msg);
  return THM_OK;
}

int fastq_parse_block(const char* p, size_t n, const std::string& path, uint64_t first_line, bool last_block, HostBatch& b,
                      std::string& err) {
  b.clear();
  const char* end = p + n;
  uint64_t line = first_line;
  auto next = [&](const char*& s, size_t& len) -> bool {
    if (p >= end) return false;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    s = p;
    len = nl ? (size_t)(nl - p) : (size_t)(end - p);
    p = nl ? nl + 1 : end;
    line++;
    while (len && s[len - 1] == '\r') len--;
    return true;
  };
  // One pass of appends into storage reserved once.  Every byte copied below comes out of the block, so n bytes
  // bound each array whatever the records look like (a well-formed block needs about n / 2, but a malformed one --
  // one sequence line of most of the block, quality missing -- is copied before it is found out).
  if (b.bases.size() < n + 4096) b.bases.resize(n + 4096);
  if (b.quals.size() < n + 4096) b.quals.resize(n + 4096);
  if (b.names.size() < n + 4096) b.names.resize(n + 4096);
  size_t nb = 0, nq = 0, nn = 0;
  const char* s;
  size_t len;
  while (next(s, len)) {
    const uint64_t at = line - 1;
    auto where = [&] { return path + ":" + std::to_string(at); };
    if (len == 0) {  // empty lines are tolerated at the very end of the INPUT only, not at the end of any block
      const char* t = p;
      while (t < end && (*t == '\n' || *t == '\r')) t++;
      if (t == end && last_block) break;
      err = "expected '@' at " + where();
      return THM_ERR_FORMAT;
    }
    if (s[0] != '@') {
      err = "expected '@' at " + where();
      return THM_ERR_FORMAT;
    }
    memcpy(b.names.data() + nn, s + 1, len - 1);
    nn += len - 1;
    b.name_off.push_back(nn);
    if (!next(s, len)) {
      err = "truncated FASTQ record at " + where();
      return THM_ERR_FORMAT;
    }
    memcpy(b.bases.data() + nb, s, len);
    const size_t slen = len;
    if (!next(s, len) || len == 0 || s[0] != '+') {
      err = "FASTQ record without a '+' line at " + where();
      return THM_ERR_FORMAT;
    }
    if (!next(s, len)) {
      if (slen != 0) {
        err = "truncated FASTQ record at " + where();
        return THM_ERR_FORMAT;
      }
      len = 0;
    }
    if (len != slen) {
      err = "FASTQ quality length differs from sequence length at " + where();
      return THM_ERR_FORMAT;
    }
    memcpy(b.quals.data() + nq, s, len);
    nb += slen;
    nq += len;
    b.offsets.push_back(nb);
  }
  b.nb = nb;
  b.nq = nq;
  b.nn = nn;
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
  thm_fastq* r = new thm_fastq();
  r->path = path;
  if (gz) {
    // worker threads of the chunk-parallel decoder (a file of some size): a quarter of the hardware threads, 2 to 8,
    // unless THM_INFLATE_THREADS says otherwise (1: the serial decoder on the caller's thread)
    unsigned n_threads = std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 4));
    if (const char* e = getenv("THM_INFLATE_THREADS")) n_threads = (unsigned)std::max(1, std::min(64, atoi(e)));
    r->z = new thm::GzInflater();
    r->z->open(fd, r->path, n_threads);
  } else {
    r->fd = fd;
  }
  r->buf.resize(8 << 20);
  *out = r;
  return THM_OK;
}

void thm_fastq_close(thm_fastq* r) {
  if (!r) return;
  delete r->z;
  if (r->fd >= 0) close(r->fd);
  delete r;
}

// test hook: the parallel driver's block cutter + block parser over a whole file; the batches are concatenated into the
// reader's own storage so that the result can be compared with the sequential parser's (thm_fastq_next_batch)
int32_t thm_debug_fastq_blocks(thm_fastq* r, uint64_t max_reads_per_block, thm_read_batch* out) {
  if (!r || !out) return THM_ERR_INVALID_ARG;
  thm::HostBatch& all = r->own;
  all.clear();
  if (!thm::fastq_is_plain_fastq(r)) return r->io_err ? fail(THM_ERR_IO, r->io_msg) : fail(THM_ERR_FORMAT, "not a plain FASTQ input");
  std::vector<char> raw;
  thm::HostBatch b;
  for (;;) {
    size_t raw_len = 0;
    uint64_t n_lines = 0, first_line = 0;
    bool last_block = false;
    int rc = thm::fastq_next_raw_block(r, max_reads_per_block, raw, raw_len, n_lines, first_line, last_block);
    if (rc != THM_OK) return rc;
    if (n_lines == 0) break;
    std::string err;
    rc = thm::fastq_parse_block(raw.data(), raw_len, r->path, first_line, last_block, b, err);
    if (rc != THM_OK) return fail(rc, err);
    const thm_read_batch v = b.view();
    for (uint64_t i = 0; i < v.n_reads; i++) {
      const size_t L = (size_t)(v.offsets[i + 1] - v.offsets[i]), N = (size_t)(v.name_off[i + 1] - v.name_off[i]);
      thm::put(all.bases, all.nb, (const char*)v.bases + v.offsets[i], L);
      thm::put(all.quals, all.nq, (const char*)v.quals + v.offsets[i], L);
      thm::put(all.names, all.nn, (const char*)v.names + v.name_off[i], N);
      all.offsets.push_back(all.nb);
      all.name_off.push_back(all.nn);
    }
  }
  *out = all.view();
  return THM_OK;
}

int32_t thm_fastq_next_batch(thm_fastq* r, uint64_t max_reads, thm_read_batch* out) {
  if (!r || !out) return THM_ERR_INVALID_ARG;
  const int rc = thm::fastq_fill(r, max_reads, r->own);
  if (rc != THM_OK) return rc;
  *out = r->own.view();
  return THM_OK;
}

}  // extern "C"
