// io_driver.cpp -- whole-file driver (include/thermite_io.h): the loop of
// align_reads_from_file, reference src/aligner.rs:22-120, as three overlapped
// stages over batches of reads:
//     parse (FASTQ -> HostBatch)  |  GPU (upload, run, sync, fetch)  |  format + write
// connected by bounded queues of a few reusable slots, so that the host work
// either side of the hot path hides behind the GPU (or the other way round:
// at tens of millions of reads per second on the GPU the host side is the
// bottleneck).  Records leave in input order.
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
  std::vector<uint64_t> aln_off;
  std::vector<thm_aln> alns;
  std::vector<uint8_t> ops;
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
  if (batch_reads == 0) batch_reads = 500000;
  const thm_index* ix = thm_aligner_index(a);
  thm_writer* w = nullptr;
  int rc = thm_writer_create(ix, format, n_threads, &w);
  if (rc != THM_OK) return rc;
  const bool to_stdout = strcmp(output_path, "-") == 0;
  FILE* fo = to_stdout ? stdout : fopen(output_path, "wb");
  if (!fo) {
    thm_writer_free(w);
    thm::set_global_error(std::string("cannot create ") + output_path);
    return THM_ERR_IO;
  }
  std::vector<char> obuf(4 << 20);
  if (!to_stdout) setvbuf(fo, obuf.data(), _IOFBF, obuf.size());

  thm_run_stats st;
  memset(&st, 0, sizeof st);
  const auto t_start = Clock::now();
  Shared sh;
  constexpr int N_SLOTS = 3;
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
    if (thm_writer_header(w, &t) == THM_OK && t.len) {
      if (fwrite(t.data, 1, t.len, fo) != t.len) sh.set(THM_ERR_IO, "short write");
      st.n_output_bytes += t.len;
    }
    for (;;) {
      Slot* s = q_aligned.pop();
      if (s->last) {
        q_free.push(s);
        break;
      }
      if (!sh.failed()) {
        const auto t0 = Clock::now();
        const thm_read_batch rb = s->reads.view();
        thm_batch_view v;
        v.n_reads = rb.n_reads;
        v.n_alns = s->alns.size();
        v.n_op_bytes = s->ops.size();
        v.read_aln_off = s->aln_off.data();
        v.alns = s->alns.data();
        v.ops = s->ops.data();
        const int wrc = thm_writer_format_batch(w, &rb, &v, &t);
        const auto t1 = Clock::now();
        st.format_s += secs(t0, t1);
        if (wrc != THM_OK) {
          sh.set(wrc, thm_last_error(nullptr));
        } else {
          if (t.len && fwrite(t.data, 1, t.len, fo) != t.len) sh.set(THM_ERR_IO, "short write");
          st.n_output_bytes += t.len;
          st.write_s += secs(t1, Clock::now());
          for (uint64_t r = 0; r < v.n_reads; r++) {
            const uint64_t k = v.read_aln_off[r + 1] - v.read_aln_off[r];
            st.n_aligned_reads += k != 0;
            st.n_records += k ? k : (format == THM_FMT_SAM ? 1 : 0);
          }
        }
      }
      q_free.push(s);
    }
  });

  // ---- stage 2: GPU (this thread; the aligner handle is single-threaded) ----
  for (;;) {
    Slot* s = q_parsed.pop();
    if (s->last) {
      q_aligned.push(s);
      break;
    }
    if (!sh.failed()) {
      const auto t0 = Clock::now();
      thm_batch_view v;
      const thm_read_batch rb = s->reads.view();
      int grc = thm_align_batch(a, rb.bases, rb.offsets, rb.n_reads, &v);
      if (grc != THM_OK) {
        sh.set(grc, thm_last_error(a));
      } else {
        s->aln_off.assign(v.read_aln_off, v.read_aln_off + v.n_reads + 1);
        s->alns.assign(v.alns, v.alns + v.n_alns);
        s->ops.assign(v.ops, v.ops + v.n_op_bytes);
        st.n_reads += v.n_reads;
        st.n_batches += 1;
      }
      st.gpu_s += secs(t0, Clock::now());
    }
    q_aligned.push(s);
  }
  parser.join();
  writer.join();
  bool ok = true;
  if (!to_stdout)
    ok = fclose(fo) == 0;
  else
    fflush(fo);
  if (!ok) sh.set(THM_ERR_IO, std::string("error closing ") + output_path);
  thm_writer_free(w);
  st.wall_s = secs(t_start, Clock::now());
  if (stats) *stats = st;
  if (sh.rc != THM_OK) thm::set_global_error(sh.msg);
  return sh.rc;
}
