// io_driver.cpp -- whole-file driver (include/thermite_io.h): the loop of
// align_reads_from_file, reference src/aligner.rs:22-120, as overlapped stages over
// batches of reads:
//     inflate one thread per gzip input file, a few files at a time: gzip bytes -> blocks of whole FASTQ
//             records, ahead of the file's turn within a memory budget (the records still leave in input order)
//     cut     one thread: mapped file bytes / the inflaters' blocks -> slots, in input order
//     parse   a few threads: block -> batch (names, bases, qualities)
//     GPU     one thread per aligner (= per GPU): upload, run, sync, fetch
//     write   one thread: batches in input order -> formatting threads -> pwrite
// connected by queues over a fixed set of reusable slots, so that the host work either
// side of the hot path hides behind the GPUs (or the other way round: at tens of millions
// of reads per second per GPU the host side is the bottleneck).
// Records leave in input order whatever the number of GPUs.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
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
  uint64_t seq = 0;  // position of the batch in the input
  std::vector<char> raw;      // block bytes (gzip input: inflated here) ...
  const char* raw_ptr = nullptr;  // ... or a range of the memory-mapped input file (plain input: no copy)
  size_t raw_len = 0;
  uint64_t first_line = 0;
  bool last_block = false;  // nothing of this input file follows the block (blank lines are tolerated at its end)
  std::string path;
  bool parsed = false;  // `reads` already holds the batch (sequential parser: FASTA, odd inputs)
  thm::HostBatch reads;
  thm_batch_view res;  // into the pinned result buffers of the aligner that ran it (two sets per aligner, used alternately)
  bool aligned = false;
};

// A gzip FASTQ file on a thread of its own: inflate, cut into blocks of whole records, queue.  The reference reads its
// query files one after the other (src/aligner.rs:51); so does the cutter -- but an inflater decodes only about a
// gigabyte a second, several times less than the stages behind it take, so the inflaters of the next files start
// early and run ahead of their turn until their share of the budget is queued.  (Nothing of this reorders the
// output: the cutter takes the files' blocks strictly in input order.)
struct Ahead {
  struct Block {
    std::vector<char> raw;
    size_t raw_len = 0;
    uint64_t n_lines = 0, first_line = 0;
    bool last_block = false;
    int rc = THM_OK;
    std::string err;
  };
  std::string path;
  uint64_t batch_reads = 0;
  size_t budget = 0;  // bytes this file may hold queued (one more block is cut once the queue is below it)
  thm_fastq* r = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Block> ready;
  std::vector<std::vector<char>> spare;
  size_t queued = 0;
  int kind = -1;  // -1 not known yet; 0 not 4-line FASTQ (the cutter runs the sequential parser on `r`); 1 blocks; 2 cannot open
  int open_rc = THM_OK;
  std::string open_err;
  bool cancel = false;
  double busy_s = 0;

  void run() {
    const auto t0 = Clock::now();
    int rc = thm_fastq_open(path.c_str(), &r);
    bool fast = false;
    if (rc == THM_OK) fast = thm::fastq_is_plain_fastq(r);
    busy_s += secs(t0, Clock::now());
    {
      std::lock_guard<std::mutex> g(mu);
      open_rc = rc;
      if (rc != THM_OK) open_err = thm_last_error(nullptr);
      kind = rc != THM_OK ? 2 : (fast ? 1 : 0);
    }
    cv.notify_all();
    if (!fast) return;
    for (;;) {
      Block b;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return queued < budget || cancel; });
        if (cancel) return;
        if (!spare.empty()) {
          b.raw.swap(spare.back());
          spare.pop_back();
        }
      }
      const auto t1 = Clock::now();
      try {
        b.rc = thm::fastq_next_raw_block(r, batch_reads, b.raw, b.raw_len, b.n_lines, b.first_line, b.last_block);
        if (b.rc != THM_OK) b.err = thm_last_error(nullptr);
      } catch (const std::bad_alloc&) {
        b.rc = THM_ERR_OOM;
        b.err = "out of host memory inflating " + path;
      }
      busy_s += secs(t1, Clock::now());
      const bool last = b.rc != THM_OK || b.n_lines == 0;
      {
        std::lock_guard<std::mutex> g(mu);
        queued += b.raw_len;
        ready.push_back(std::move(b));
      }
      cv.notify_all();
      if (last) return;
    }
  }
  void start() {
    th = std::thread([this] { run(); });
  }
  int wait_kind() {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return kind >= 0; });
    return kind;
  }
  // the next block of the file (the last one holds no line, or an error); `old` goes back for reuse
  void next(Block& out, std::vector<char>& old) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return !ready.empty(); });
    out = std::move(ready.front());
    ready.pop_front();
    queued -= out.raw_len;
    if (old.capacity()) spare.emplace_back(std::move(old));
    g.unlock();
    cv.notify_all();
  }
  void stop() {
    {
      std::lock_guard<std::mutex> g(mu);
      cancel = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
    if (r) thm_fastq_close(r);
    r = nullptr;
  }
};

// blocking queue of slot pointers; nullptr = end of stream
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
  // Runs one step of a pipeline thread.  An exception (bad_alloc from a growing buffer, mostly) must not leave the
  // thread body: std::terminate would take the host process down, and a thread that simply stopped would leave the
  // others waiting on its queue.  The step's failure becomes the run's status and the thread goes on with its
  // hand-over protocol.
  template <class F>
  int step(F&& f) {
    try {
      return f();
    } catch (const std::bad_alloc&) {
      set(THM_ERR_OOM, "out of host memory in the file driver");
      return THM_ERR_OOM;
    } catch (const std::exception& e) {
      set(THM_ERR_INTERNAL, std::string("file driver: ") + e.what());
      return THM_ERR_INTERNAL;
    } catch (...) {
      set(THM_ERR_INTERNAL, "file driver: unknown exception");
      return THM_ERR_INTERNAL;
    }
  }
};

}  // namespace

extern "C" int32_t thm_align_files_multi(thm_aligner* const* aligners, uint32_t n_aligners, const char* const* fastq_paths,
                                         uint32_t n_paths, const char* output_path, int32_t format, uint64_t batch_reads,
                                         uint32_t n_threads, thm_run_stats* stats) {
  if (!aligners || n_aligners == 0 || !fastq_paths || n_paths == 0 || !output_path) return THM_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < n_paths; i++)
    if (!fastq_paths[i]) return THM_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < n_aligners; i++)
    if (!aligners[i]) return THM_ERR_INVALID_ARG;
  if (batch_reads == 0) batch_reads = 250000;
  const thm_index* ix = thm_aligner_index(aligners[0]);
  for (uint32_t i = 1; i < n_aligners; i++)
    if (thm_aligner_index(aligners[i]) != ix) {
      thm::set_global_error("thm_align_files_multi: the aligners must share one index");
      return THM_ERR_INVALID_ARG;
    }
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  if (n_threads == 0) n_threads = std::min(hw, 16u);
  // parser threads: a block parses at roughly 10 M records/s per thread; the rest of the budget formats
  const unsigned n_parsers = std::max(1u, std::min(4u, n_threads / 4));
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
  std::mutex st_mu;  // parse_s / gpu_s are summed by several threads
  const auto t_start = Clock::now();
  Shared sh;
  // every aligner keeps the results of its last two fetches: two batches per GPU may be between fetch and write,
  // one more is running; the parsers and the cutter work ahead of that
  const unsigned n_slots = 3 * n_aligners + n_parsers + 2;
  std::vector<Slot> slots(n_slots);
  Queue q_free, q_raw;
  for (auto& s : slots) q_free.push(&s);
  // parsed batches are handed to the GPU threads strictly in input order (parsers may finish out of order; an aligner
  // that ran ahead of a batch it would later have to wait for could block the writer for good)
  std::mutex par_mu;
  std::condition_variable par_cv;
  std::map<uint64_t, Slot*> parsed;
  uint64_t next_align = 0;
  bool parse_done = false;
  // finished batches, by sequence number, for the writer
  std::mutex done_mu;
  std::condition_variable done_cv;
  std::map<uint64_t, Slot*> done;
  uint64_t n_batches_cut = 0;  // set when the cutter has seen the end of the input
  bool cut_done = false;
  uint64_t n_written = 0;  // batches the writer has finished (guarded by done_mu)

  std::vector<std::pair<const char*, size_t>> maps;
  // ---- stage 1: cut the input into blocks of whole records ----
  std::thread cutter([&] {
    uint64_t seq = 0;
    // gzip inputs get an inflater thread each (Ahead), those of the next few files running ahead of their turn
    std::vector<std::unique_ptr<Ahead>> ahead(n_paths);
    std::vector<char> is_gz(n_paths, 0);
    unsigned n_gz = 0;
    for (uint32_t pi = 0; pi < n_paths; pi++) {
      unsigned char magic[2] = {0, 0};
      const int fd = open(fastq_paths[pi], O_RDONLY);
      if (fd >= 0) {
        is_gz[pi] = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        close(fd);
      }
      n_gz += is_gz[pi];
    }
    const unsigned ahead_files = std::max(1u, std::min(n_gz, std::max(2u, n_threads / 4)));
    size_t ahead_bytes = (size_t)768 << 20;  // all inflaters together; THM_INFLATE_AHEAD_MB overrides
    if (const char* e = getenv("THM_INFLATE_AHEAD_MB")) ahead_bytes = (size_t)std::max(1L, atol(e)) << 20;
    auto top_up = [&](uint32_t from) {  // the inflaters of the next `ahead_files` gzip files at or behind `from`
      unsigned running = 0;
      for (uint32_t k = from; k < n_paths && running < ahead_files; k++) {
        if (!is_gz[k]) continue;
        running++;
        if (ahead[k]) continue;
        ahead[k].reset(new Ahead());
        ahead[k]->path = fastq_paths[k];
        ahead[k]->batch_reads = batch_reads;
        ahead[k]->budget = ahead_bytes / ahead_files;
        ahead[k]->start();
      }
    };
    for (uint32_t pi = 0; pi < n_paths && !sh.failed(); pi++) {
      thm_fastq* r = nullptr;
      Ahead* ah = nullptr;
      int prc = THM_OK;
      bool fast = false;
      if (sh.step([&] { top_up(pi); return THM_OK; }) != THM_OK) break;
      if (is_gz[pi]) {
        ah = ahead[pi].get();
        const int kind = ah->wait_kind();
        if (kind == 2) {
          sh.set(ah->open_rc, ah->open_err);
          break;
        }
        fast = kind == 1;
        r = ah->r;  // (the sequential parser's, when the file is not 4-line FASTQ: the inflater thread has left)
      } else {
        prc = thm_fastq_open(fastq_paths[pi], &r);
        if (prc != THM_OK) {
          sh.set(prc, thm_last_error(nullptr));
          break;
        }
        fast = thm::fastq_is_plain_fastq(r);
      }
      // a plain (not gzip) FASTQ file is mapped and cut in place: a block is a range of the mapping
      const char* map = nullptr;
      size_t map_len = 0, map_pos = 0;
      uint64_t map_line = 1;
      if (fast) {
        unsigned char magic[2] = {0, 0};
        const int fd = open(fastq_paths[pi], O_RDONLY);
        struct stat ms;
        if (fd >= 0 && fstat(fd, &ms) == 0 && S_ISREG(ms.st_mode) && ms.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
          void* m = mmap(nullptr, (size_t)ms.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
          if (m != MAP_FAILED) {
            map = (const char*)m;
            map_len = (size_t)ms.st_size;
            (void)madvise(m, map_len, MADV_SEQUENTIAL);
          }
        }
        if (fd >= 0) close(fd);
      }
      auto count_nl = [](const char* p, size_t n) {
        size_t c = 0;
        for (size_t i = 0; i < n; i++) c += p[i] == '\n';
        return c;
      };
      for (;;) {
        Slot* s = q_free.pop();
        s->seq = seq;
        s->parsed = false;
        s->aligned = false;
        s->path = fastq_paths[pi];
        const auto t0 = Clock::now();
        uint64_t n_lines = 0;
        s->raw_ptr = nullptr;
        if (map) {
          // the position behind the (4 x batch_reads)-th newline from map_pos, found a span at a time
          const uint64_t want = batch_reads * 4;
          size_t p0 = map_pos, p = map_pos;
          while (p < map_len && n_lines < want) {
            const size_t span = std::min<size_t>(map_len - p, 1u << 20);
            const size_t nl = count_nl(map + p, span);
            if (n_lines + nl < want) {
              n_lines += nl;
              p += span;
              continue;
            }
            const char* q = map + p;
            while (n_lines < want) {
              q = (const char*)memchr(q, '\n', (size_t)(map + p + span - q)) + 1;
              n_lines++;
            }
            p = (size_t)(q - map);
          }
          if (p == map_len && p > p0 && map[p - 1] != '\n') n_lines++;
          s->raw_ptr = map + p0;
          s->raw_len = p - p0;
          s->last_block = p == map_len;
          s->first_line = map_line;
          map_line += n_lines;
          map_pos = p;
          prc = THM_OK;
        } else if (fast && ah) {
          Ahead::Block b;
          ah->next(b, s->raw);
          s->raw.swap(b.raw);
          s->raw_len = b.raw_len;
          s->first_line = b.first_line;
          s->last_block = b.last_block;
          n_lines = b.n_lines;
          prc = b.rc;
          if (prc != THM_OK) thm::set_global_error(b.err);
          s->raw_ptr = s->raw.data();
        } else if (fast) {
          prc = sh.step([&] { return thm::fastq_next_raw_block(r, batch_reads, s->raw, s->raw_len, n_lines, s->first_line, s->last_block); });
          s->raw_ptr = s->raw.data();
        } else {  // FASTA and anything that is not plain 4-line FASTQ: the sequential parser
          prc = sh.step([&] { return thm::fastq_fill(r, batch_reads, s->reads); });
          s->parsed = true;
          n_lines = s->reads.n_reads();
        }
        if (!(fast && ah)) {  // (an inflater thread counts its own time)
          std::lock_guard<std::mutex> g(st_mu);
          st.parse_s += secs(t0, Clock::now());
        }
        if (prc != THM_OK || sh.failed() || n_lines == 0) {
          if (prc != THM_OK) sh.set(prc, thm_last_error(nullptr));
          q_free.push(s);
          break;
        }
        seq++;
        q_raw.push(s);
      }
      if (ah) {
        ah->stop();  // (closes the reader)
        std::lock_guard<std::mutex> g(st_mu);
        st.parse_s += ah->busy_s;
      } else {
        thm_fastq_close(r);
      }
      if (map) maps.emplace_back(map, map_len);  // unmapped when every batch has been written
    }
    for (auto& a : ahead)
      if (a) a->stop();  // (inflaters started ahead of a run that failed)
    {
      std::lock_guard<std::mutex> g(done_mu);
      n_batches_cut = seq;
      cut_done = true;
    }
    done_cv.notify_all();
    for (unsigned k = 0; k < n_parsers; k++) q_raw.push(nullptr);
  });

  // ---- stage 2: parse blocks (several threads) ----
  std::vector<std::thread> parsers;
  std::mutex parsers_mu;
  unsigned parsers_left = n_parsers;
  for (unsigned k = 0; k < n_parsers; k++)
    parsers.emplace_back([&] {
      for (;;) {
        Slot* s = q_raw.pop();
        if (!s) break;
        if (!s->parsed && !sh.failed()) {
          const auto t0 = Clock::now();
          std::string err;
          const int prc = sh.step([&] { return thm::fastq_parse_block(s->raw_ptr, s->raw_len, s->path, s->first_line, s->last_block, s->reads, err); });
          {
            std::lock_guard<std::mutex> g(st_mu);
            st.parse_s += secs(t0, Clock::now());
          }
          if (prc != THM_OK) sh.set(prc, err);
        }
        {
          std::lock_guard<std::mutex> g(par_mu);
          parsed[s->seq] = s;
        }
        par_cv.notify_all();
      }
      std::lock_guard<std::mutex> g(parsers_mu);
      if (--parsers_left == 0) {
        {
          std::lock_guard<std::mutex> g2(par_mu);
          parse_done = true;
        }
        par_cv.notify_all();
      }
    });

  // ---- stage 3: one thread per aligner ----
  std::vector<std::thread> gpus;
  for (uint32_t gi = 0; gi < n_aligners; gi++)
    gpus.emplace_back([&, gi] {
      thm_aligner* a = aligners[gi];
      std::deque<uint64_t> fetched;  // sequence numbers of this aligner's fetched batches, oldest first
      for (;;) {
        Slot* s = nullptr;
        {
          std::unique_lock<std::mutex> g(par_mu);
          par_cv.wait(g, [&] { return parsed.count(next_align) || (parse_done && parsed.empty()); });
          auto it = parsed.find(next_align);
          if (it == parsed.end()) break;
          s = it->second;
          parsed.erase(it);
          next_align++;
        }
        par_cv.notify_all();
        if (!sh.failed() && s->reads.n_reads() != 0) {  // (a blank tail cut into a block of its own holds no read)
          const auto t0 = Clock::now();
          const thm_read_batch rb = s->reads.view();
          int grc = sh.step([&] {
            int g2 = thm_batch_upload(a, rb.bases, rb.offsets, rb.n_reads);
            if (g2 == THM_OK) g2 = thm_batch_run(a);
            if (g2 == THM_OK) g2 = thm_batch_sync(a);
            return g2;
          });
          const auto t1 = Clock::now();
          // the fetch below reuses the buffers of this aligner's second-last fetch: that batch must have been written
          if (grc == THM_OK && fetched.size() >= 2) {
            const uint64_t must = fetched[fetched.size() - 2];
            std::unique_lock<std::mutex> g(done_mu);
            done_cv.wait(g, [&] { return n_written > must || sh.failed(); });
          }
          const auto t2 = Clock::now();
          if (grc == THM_OK) grc = sh.step([&] { return thm_batch_fetch(a, &s->res); });
          if (grc == THM_OK && s->res.n_failed_reads) {
            // the reference aligns every read or panics; a read this build cannot take fails the run, by name
            uint64_t bad = 0;
            while (bad < s->res.n_reads && s->res.read_status[bad] == THM_OK) bad++;
            const thm_read_batch v = s->reads.view();
            std::string name((const char*)v.names + v.name_off[bad], (size_t)(v.name_off[bad + 1] - v.name_off[bad]));
            sh.set(s->res.read_status[bad], "read " + name + (s->res.read_status[bad] == THM_ERR_UNSUPPORTED
                                                                   ? ": longer than 65535 bases, or its DP trace exceeds the device-memory budget"
                                                                   : ": hits a condition that panics in the reference (lift_mem_to_tx / lift_tx_to_gx)"));
            grc = s->res.read_status[bad];
          }
          if (grc != THM_OK) {
            sh.set(grc, thm_last_error(a));
          } else {
            s->aligned = true;
            fetched.push_back(s->seq);
            if (fetched.size() > 4) fetched.pop_front();
          }
          std::lock_guard<std::mutex> g(st_mu);
          st.gpu_s += secs(t0, t1) + secs(t2, Clock::now());
          if (grc == THM_OK) {
            st.n_reads += s->res.n_reads;
            st.n_batches += 1;
          }
        }
        {
          std::lock_guard<std::mutex> g(done_mu);
          done[s->seq] = s;
        }
        done_cv.notify_all();
      }
    });

  // ---- stage 4: format (this thread, on the formatting threads of the writer) and write (one more thread), in input
  // order; two writer objects are used alternately, so that batch k + 1 is formatted while batch k is written ----
  {
    // two writer objects, used alternately: the chunks of batch k are views into writer k & 1 and are still being
    // written while batch k + 1 is formatted, so one object cannot stand in for both
    thm_writer* w2 = nullptr;
    if (thm_writer_create(ix, format, n_threads, &w2) != THM_OK) {
      w2 = nullptr;
      sh.set(THM_ERR_OOM, "cannot create the second writer object");
    }
    thm_writer* ws[2] = {w, w2 ? w2 : w};  // (after a failure nothing is formatted any more: sh.failed())
    struct WriteJob {
      Slot* s = nullptr;
      uint64_t seq = 0;
      std::vector<const std::string*> chunks;
      std::vector<uint64_t> at;
      bool stop = false;
    };
    std::mutex wj_mu;
    std::condition_variable wj_cv;
    std::deque<WriteJob*> wj_q;
    WriteJob jobs[2];
    bool job_busy[2] = {false, false};
    thm_text t;
    if (thm_writer_header(w, &t) == THM_OK && t.len) {
      if (!write_all((const char*)t.data, t.len, file_off)) sh.set(THM_ERR_IO, "short write");
      file_off += t.len;
      st.n_output_bytes += t.len;
    }
    auto release = [&](Slot* s, uint64_t seq) {
      {
        std::lock_guard<std::mutex> g(done_mu);
        n_written = seq + 1;
      }
      done_cv.notify_all();
      q_free.push(s);
    };
    std::thread wthread([&] {
      for (;;) {
        WriteJob* j;
        {
          std::unique_lock<std::mutex> g(wj_mu);
          wj_cv.wait(g, [&] { return !wj_q.empty(); });
          j = wj_q.front();
          wj_q.pop_front();
        }
        if (j->stop) break;
        const auto t1 = Clock::now();
        std::vector<char> good(j->chunks.size(), 1);
        auto put = [&](size_t c) { good[c] = write_all(j->chunks[c]->data(), j->chunks[c]->size(), j->at[c]) ? 1 : 0; };
        if (positional && j->chunks.size() > 1) {
          std::vector<std::thread> th;
          for (size_t c = 1; c < j->chunks.size(); c++) th.emplace_back(put, c);
          put(0);
          for (auto& x : th) x.join();
        } else {
          for (size_t c = 0; c < j->chunks.size(); c++) put(c);
        }
        for (char g : good)
          if (!g) sh.set(THM_ERR_IO, "short write");
        st.write_s += secs(t1, Clock::now());
        release(j->s, j->seq);
        {
          std::lock_guard<std::mutex> g(wj_mu);
          job_busy[j - jobs] = false;
        }
        wj_cv.notify_all();
      }
    });
    for (uint64_t next = 0;; next++) {
      Slot* s = nullptr;
      {
        std::unique_lock<std::mutex> g(done_mu);
        done_cv.wait(g, [&] { return done.count(next) || (cut_done && next >= n_batches_cut); });
        auto it = done.find(next);
        if (it == done.end()) break;  // the input is exhausted
        s = it->second;
        done.erase(it);
      }
      const int k = (int)(next & 1);
      {  // the writer object (and job) of this parity: its previous batch must have left
        std::unique_lock<std::mutex> g(wj_mu);
        wj_cv.wait(g, [&] { return !job_busy[k]; });
      }
      bool queued = false;
      if (!sh.failed() && s->aligned) {
        const auto t0 = Clock::now();
        const thm_read_batch rb = s->reads.view();
        const thm_batch_view& v = s->res;
        WriteJob& j = jobs[k];
        const int wrc = sh.step([&] { return thm::writer_format_chunks(ws[k], &rb, &v, j.chunks); });
        st.format_s += secs(t0, Clock::now());
        if (wrc != THM_OK) {
          sh.set(wrc, thm_last_error(nullptr));
        } else {
          j.at.resize(j.chunks.size());
          for (size_t c = 0; c < j.chunks.size(); c++) {
            j.at[c] = file_off;
            file_off += j.chunks[c]->size();
            st.n_output_bytes += j.chunks[c]->size();
          }
          for (uint64_t r = 0; r < v.n_reads; r++) {
            const uint64_t kk = v.read_aln_off[r + 1] - v.read_aln_off[r];
            st.n_aligned_reads += kk != 0;
            st.n_records += kk ? kk : (format == THM_FMT_PAF ? 0 : 1);
          }
          j.s = s;
          j.seq = next;
          {
            std::lock_guard<std::mutex> g(wj_mu);
            job_busy[k] = true;
            wj_q.push_back(&j);
          }
          wj_cv.notify_all();
          queued = true;
        }
      }
      if (!queued) {
        // nothing to write for this batch: it still leaves in order behind the writes that are queued
        std::unique_lock<std::mutex> g(wj_mu);
        wj_cv.wait(g, [&] { return !job_busy[0] && !job_busy[1]; });
        g.unlock();
        release(s, next);
      }
    }
    {
      std::unique_lock<std::mutex> g(wj_mu);
      wj_cv.wait(g, [&] { return !job_busy[0] && !job_busy[1]; });
    }
    WriteJob stop;
    stop.stop = true;
    {
      std::lock_guard<std::mutex> g(wj_mu);
      wj_q.push_back(&stop);
    }
    wj_cv.notify_all();
    wthread.join();
    if (!sh.failed() && thm_writer_trailer(w, &t) == THM_OK && t.len) {
      if (!write_all((const char*)t.data, t.len, file_off)) sh.set(THM_ERR_IO, "short write");
      file_off += t.len;
      st.n_output_bytes += t.len;
    }
    if (w2) thm_writer_free(w2);
  }
  cutter.join();
  for (auto& m : maps) munmap((void*)m.first, m.second);
  for (auto& x : parsers) x.join();
  for (auto& x : gpus) x.join();
  bool ok = true;
  if (!to_stdout) ok = close(fo) == 0;
  if (!ok) sh.set(THM_ERR_IO, std::string("error closing ") + output_path);
  thm_writer_free(w);
  st.wall_s = secs(t_start, Clock::now());
  if (stats) *stats = st;
  if (sh.rc != THM_OK) thm::set_global_error(sh.msg);
  return sh.rc;
}

extern "C" int32_t thm_align_files(thm_aligner* a, const char* const* fastq_paths, uint32_t n_paths, const char* output_path,
                                   int32_t format, uint64_t batch_reads, uint32_t n_threads, thm_run_stats* stats) {
  if (!a) return THM_ERR_INVALID_ARG;
  thm_aligner* one[1] = {a};
  return thm_align_files_multi(one, 1, fastq_paths, n_paths, output_path, format, batch_reads, n_threads, stats);
}
