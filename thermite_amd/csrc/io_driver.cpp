// io_driver.cpp -- whole-file driver (include/thermite_io.h): the loop of
// align_reads_from_file, reference src/aligner.rs:22-120, as three overlapped
// stages over batches of reads:
//     parse (FASTQ -> HostBatch)  |  GPU (upload, run, sync, fetch)  |  format + write
// connected by bounded queues of a few reusable slots, so that the host work
// either side of the hot path hides behind the GPU (or the other way round:
// at tens of millions of reads per second on the GPU the host side is the
// bottleneck).  Records leave in input order.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

namespace {

using Clock = std::chrono::steady_clock;
double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Slot {
  thm::HostBatch reads;
  thm_batch_view res;  // into the aligner's pinned result buffers (two sets, used alternately)
  bool aligned = false;
  bool last = false;  // sentinel: no more batches
};

// blocking queue of slot pointers
struct Queue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Slot*> q;
  void push(Slot* s) {
    {
      std::lock_guard<std::mutex> g(mu);
      q.push_back(s);
    }
    cv.notify_one();
  }
  Slot* pop() {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return !q.empty(); });
    Slot* s = q.front();
    q.pop_front();
    return s;
  }
};

struct Shared {
  std::mutex mu;
  int rc = THM_OK;
  std::string msg;
  void set(int code, const std::string& m) {
    std::lock_guard<std::mutex> g(mu);
    if (rc == THM_OK) {
      rc = code;
      msg = m;
    }
  }
  bool failed() {
    std::lock_guard<std::mutex> g(mu);
    return rc != THM_OK;
  }
};

}  // namespace

extern "C" int32_t thm_align_files(thm_aligner* a, const char* const* fastq_paths, uint32_t n_paths, const char* output_path,
                                   int32_t format, uint64_t batch_reads, uint32_t n_threads, thm_run_stats* stats) {
  if (!a || !fastq_paths || n_paths == 0 || !output_path) return THM_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < n_paths; i++)
    if (!fastq_paths[i]) return THM_ERR_INVALID_ARG;
  if (batch_reads == 0) batch_reads = 250000;
  const thm_index* ix = thm_aligner_index(a);
  thm_writer* w = nullptr;
  int rc = thm_writer_create(ix, format, n_threads, &w);
  if (rc != THM_OK) return rc;
  const bool to_stdout = strcmp(output_path, "-") == 0;
  if (to_stdout) fflush(stdout);
  const int fo = to_stdout ? 1 : open(output_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fo < 0) {
    thm_writer_free(w);
    thm::set_global_error(std::string("cannot create ") + output_path);
    return THM_ERR_IO;
  }
  // a regular file takes the chunks of a batch at their final offsets from several threads
  struct stat sb;
  const bool positional = !to_stdout && fstat(fo, &sb) == 0 && S_ISREG(sb.st_mode);
  uint64_t file_off = 0;
  auto write_all = [&](const char* p, size_t n, uint64_t off) -> bool {
    while (n) {
      const ssize_t k = positional ? pwrite(fo, p, n, (off_t)off) : write(fo, p, n);
      if (k <= 0) return false;
      p += k;
      n -= (size_t)k;
      off += (uint64_t)k;
    }
    return true;
  };

  thm_run_stats st;
  memset(&st, 0, sizeof st);
  const auto t_start = Clock::now();
  Shared sh;
  constexpr int N_SLOTS = 4;
  // the aligner keeps the results of the last two fetches: batch j may be fetched only
  // once batch j-2 has left the writer
  std::mutex done_mu;
  std::condition_variable done_cv;
  uint64_t n_written = 0;
  Slot slots[N_SLOTS];
  Queue q_free, q_parsed, q_aligned;
  for (auto& s : slots) q_free.push(&s);

  // ---- stage 1: parse ----
  std::thread parser([&] {
    for (uint32_t pi = 0; pi < n_paths && !sh.failed(); pi++) {
      thm_fastq* r = nullptr;
      int prc = thm_fastq_open(fastq_paths[pi], &r);
      if (prc != THM_OK) {
        sh.set(prc, thm_last_error(nullptr));
        break;
      }
      for (;;) {
        Slot* s = q_free.pop();
        s->last = false;
        const auto t0 = Clock::now();
        prc = thm::fastq_fill(r, batch_reads, s->reads);
        st.parse_s += secs(t0, Clock::now());
        if (prc != THM_OK || sh.failed() || s->reads.n_reads() == 0) {
          if (prc != THM_OK) sh.set(prc, thm_last_error(nullptr));
          q_free.push(s);
          break;
        }
        q_parsed.push(s);
      }
      thm_fastq_close(r);
    }
    Slot* s = q_free.pop();
    s->last = true;
    q_parsed.push(s);
  });

  // ---- stage 3: format + write ----
  std::thread writer([&] {
    thm_text t;
    std::vector<const std::string*> chunks;
    if (thm_writer_header(w, &t) == THM_OK && t.len) {
      if (!write_all((const char*)t.data, t.len, file_off)) sh.set(THM_ERR_IO, "short write");
      file_off += t.len;
      st.n_output_bytes += t.len;
    }
    for (;;) {
      Slot* s = q_aligned.pop();
      if (s->last) {
        if (!sh.failed() && thm_writer_trailer(w, &t) == THM_OK && t.len) {
          if (!write_all((const char*)t.data, t.len, file_off)) sh.set(THM_ERR_IO, "short write");
          file_off += t.len;
          st.n_output_bytes += t.len;
        }
        q_free.push(s);
        break;
      }
      if (!sh.failed() && s->aligned) {
        const auto t0 = Clock::now();
        const thm_read_batch rb = s->reads.view();
        const thm_batch_view& v = s->res;
        const int wrc = thm::writer_format_chunks(w, &rb, &v, chunks);
        const auto t1 = Clock::now();
        st.format_s += secs(t0, t1);
        if (wrc != THM_OK) {
          sh.set(wrc, thm_last_error(nullptr));
        } else {
          std::vector<uint64_t> at(chunks.size());
          for (size_t c = 0; c < chunks.size(); c++) {
            at[c] = file_off;
            file_off += chunks[c]->size();
            st.n_output_bytes += chunks[c]->size();
          }
          std::vector<char> good(chunks.size(), 1);
          auto put = [&](size_t c) { good[c] = write_all(chunks[c]->data(), chunks[c]->size(), at[c]) ? 1 : 0; };
          if (positional && chunks.size() > 1) {
            std::vector<std::thread> th;
            for (size_t c = 1; c < chunks.size(); c++) th.emplace_back(put, c);
            put(0);
            for (auto& x : th) x.join();
          } else {
            for (size_t c = 0; c < chunks.size(); c++) put(c);
          }
          for (char g : good)
            if (!g) sh.set(THM_ERR_IO, "short write");
          st.write_s += secs(t1, Clock::now());
          for (uint64_t r = 0; r < v.n_reads; r++) {
            const uint64_t k = v.read_aln_off[r + 1] - v.read_aln_off[r];
            st.n_aligned_reads += k != 0;
            st.n_records += k ? k : (format == THM_FMT_PAF ? 0 : 1);
          }
        }
      }
      {
        std::lock_guard<std::mutex> g(done_mu);
        n_written++;
      }
      done_cv.notify_all();
      q_free.push(s);
    }
  });

  // ---- stage 2: GPU (this thread; the aligner handle is single-threaded) ----
  uint64_t n_pushed = 0;  // batches handed to the writer so far
  for (;;) {
    Slot* s = q_parsed.pop();
    if (s->last) {
      q_aligned.push(s);
      break;
    }
    s->aligned = false;
    if (!sh.failed()) {
      const auto t0 = Clock::now();
      const thm_read_batch rb = s->reads.view();
      int grc = thm_batch_upload(a, rb.bases, rb.offsets, rb.n_reads);
      if (grc == THM_OK) grc = thm_batch_run(a);
      auto t1 = Clock::now();
      if (grc == THM_OK) {
        std::unique_lock<std::mutex> g(done_mu);
        done_cv.wait(g, [&] { return n_written + 1 >= n_pushed; });
      }
      const auto t2 = Clock::now();
      if (grc == THM_OK) grc = thm_batch_fetch(a, &s->res);
      if (grc != THM_OK) {
        sh.set(grc, thm_last_error(a));
      } else {
        s->aligned = true;
        st.n_reads += s->res.n_reads;
        st.n_batches += 1;
      }
      st.gpu_s += secs(t0, t1) + secs(t2, Clock::now());
    }
    n_pushed++;
    q_aligned.push(s);
  }
  parser.join();
  writer.join();
  bool ok = true;
  if (!to_stdout) ok = close(fo) == 0;
  if (!ok) sh.set(THM_ERR_IO, std::string("error closing ") + output_path);
  thm_writer_free(w);
  st.wall_s = secs(t_start, Clock::now());
  if (stats) *stats = st;
  if (sh.rc != THM_OK) thm::set_global_error(sh.msg);
  return sh.rc;
}
