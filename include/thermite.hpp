// thermite.hpp -- thin C++ mirror of the reference's aligner API over the C ABI
// (include/thermite.h).  Header-only; links against libthermite_amd.so.
//
// Names and argument meaning follow the reference so that call sites read the
// same:
//     thermite::AlignOpts      <->  aligner::AlignOpts        src/aligner.rs:452-464
//     thermite::Index          <->  index::Index              src/index.rs:39-44  (Arc-shared, immutable)
//     thermite::Aligner        <->  wrapper::ThermiteAligner  src/wrapper.rs:20-27 (one per thread / GPU)
//     Aligner::align_read      <->  aligner::align_read       src/aligner.rs:123
//     Aligner::align_reads     <->  the loop of align_reads_from_file, src/aligner.rs:51-56, batched
//     Aligner::all_smems       <->  Index::all_smems          src/index.rs:228
//     Aligner::swg_extend      <->  SwgExtend::extend         src/swg.rs:31
//     thermite::ThermiteAligner<->  wrapper::ThermiteAligner  src/wrapper.rs:20-123 (index file in, SAM records out)
//     thermite::align_reads_from_file <-> aligner::align_reads_from_file  src/aligner.rs:22-120
// Errors that are panics in the reference are exceptions here (never across the C ABI).
#ifndef THERMITE_AMD_THERMITE_HPP
#define THERMITE_AMD_THERMITE_HPP

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "thermite.h"
#include "thermite_io.h"

namespace thermite {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// reference defaults: src/main.rs:115-140, src/wrapper.rs:40-46
struct AlignOpts {
  std::size_t min_seed_len = 20;
  float min_aln_score_percent = 0.66f;
  std::int32_t min_aln_score = 30;
  std::size_t multimap_score_range = 1;
  bool intron_mode = false;
  thm_align_opts c() const {
    thm_align_opts o{};
    o.min_seed_len = min_seed_len;
    o.min_aln_score_percent = min_aln_score_percent;
    o.min_aln_score = min_aln_score;
    o.multimap_score_range = multimap_score_range;
    o.intron_mode = intron_mode ? 1 : 0;
    return o;
  }
};

enum class Op : std::uint8_t { Match = 0, Subst = 1, Del = 2, Ins = 3, Xclip = 4, Yclip = 5 };
struct AlignmentOperation {
  Op op;
  std::uint32_t len;  // 1 except for clips
};

// decode a serialised op stream into the reference's Vec<AlignmentOperation>
inline std::vector<AlignmentOperation> decode_ops(const std::uint8_t* p, std::size_t n) {
  std::vector<AlignmentOperation> out;
  for (std::size_t i = 0; i < n;) {
    const std::uint8_t k = p[i++];
    if (k >= THM_OP_XCLIP) {
      std::uint32_t v = 0;
      for (int b = 0; b < 4; b++) v |= (std::uint32_t)p[i + b] << (8 * b);
      i += 4;
      out.push_back({(Op)k, v});
    } else {
      out.push_back({(Op)k, 1});
    }
  }
  return out;
}

class Index {
 public:
  // tables in concatenated coordinates, see thm_index_create_in_memory
  Index(const std::vector<std::uint8_t>& text, const std::vector<thm_ref>& refs, const std::vector<thm_tx>& txs,
        const std::vector<thm_exon>& exons, const std::vector<std::uint8_t>& tx_seq, const std::vector<thm_span>& genes,
        const std::vector<std::uint32_t>& name_rank) {
    thm_index* h = nullptr;
    const int rc = thm_index_create_in_memory(text.data(), text.size(), refs.data(), (std::uint32_t)refs.size(), txs.data(),
                                              (std::uint32_t)txs.size(), exons.data(), exons.size(), tx_seq.data(),
                                              tx_seq.size(), genes.data(), (std::uint32_t)genes.size(), name_rank.data(),
                                              (std::uint32_t)name_rank.size(), nullptr, &h);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    h_.reset(h, thm_index_free);
  }
  // Index::create_from_files, src/index.rs:52-223
  static Index from_files(const std::string& fasta_path, const std::string& gtf_path) {
    thm_index* h = nullptr;
    const int rc = thm_index_create_from_files(fasta_path.c_str(), gtf_path.c_str(), &h);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    return Index(h);
  }
  // the index file written by save() (the reference's .tai role, src/main.rs:37-43)
  static Index load(const std::string& path) {
    thm_index* h = nullptr;
    const int rc = thm_index_load(path.c_str(), &h);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    return Index(h);
  }
  void save(const std::string& path) const {
    const int rc = thm_index_save(h_.get(), path.c_str());
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
  }
  const thm_index* get() const { return h_.get(); }

 private:
  explicit Index(thm_index* h) : h_(h, thm_index_free) {}
  std::shared_ptr<thm_index> h_;  // Arc<Index>, src/wrapper.rs:22
};

struct GenomeAlignment {  // src/txome.rs:54-61
  thm_aln rec;
  std::vector<AlignmentOperation> operations;     // gx_aln.operations
  std::vector<AlignmentOperation> tx_operations;  // AlnType::Exonic::tx_aln.operations
};

class Aligner {
 public:
  Aligner(const Index& index, const AlignOpts& opts, int device = 0) : index_(index) {
    thm_align_opts o = opts.c();
    thm_aligner* h = nullptr;
    const int rc = thm_aligner_create(index.get(), &o, device, &h);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    h_.reset(h, thm_aligner_free);
  }
  // aligner::align_read for a batch; result[r] = alignments of read r in the reference's order
  std::vector<std::vector<GenomeAlignment>> align_reads(const std::vector<std::string>& reads) {
    std::vector<std::uint8_t> bases;
    std::vector<std::uint64_t> off{0};
    for (const auto& r : reads) {
      bases.insert(bases.end(), r.begin(), r.end());
      off.push_back(bases.size());
    }
    thm_batch_view v;
    check(thm_align_batch(h_.get(), bases.data(), off.data(), reads.size(), &v));
    std::vector<std::vector<GenomeAlignment>> out(reads.size());
    for (std::uint64_t r = 0; r < v.n_reads; r++)
      for (std::uint64_t a = v.read_aln_off[r]; a < v.read_aln_off[r + 1]; a++) {
        GenomeAlignment g;
        g.rec = v.alns[a];
        g.operations = decode_ops(v.ops + g.rec.ops_off, g.rec.ops_len);
        if (g.rec.aln_type == THM_ALN_EXONIC) g.tx_operations = decode_ops(v.ops + g.rec.tx_ops_off, g.rec.tx_ops_len);
        out[r].push_back(std::move(g));
      }
    return out;
  }
  std::vector<GenomeAlignment> align_read(const std::string& read) { return align_reads({read})[0]; }

  std::vector<thm_mem> all_smems(const std::string& query, std::size_t min_seed_len) {
    const std::uint64_t off[2] = {0, query.size()};
    thm_mems_view v;
    check(thm_smems_batch(h_.get(), (const std::uint8_t*)query.data(), off, 1, min_seed_len, &v));
    return std::vector<thm_mem>(v.mems, v.mems + v.n_mems);
  }

  struct SwgAlignment {
    std::int32_t score;
    std::size_t xend, yend;
    std::vector<AlignmentOperation> operations;
  };
  SwgAlignment swg_extend(const std::string& x, const std::string& y, std::uint32_t band_width, std::int32_t x_drop,
                          std::uint32_t max_band_width) {
    const std::uint64_t xo[2] = {0, x.size()}, yo[2] = {0, y.size()};
    thm_swg_view v;
    check(thm_swg_extend_batch(h_.get(), (const std::uint8_t*)x.data(), xo, (const std::uint8_t*)y.data(), yo, &band_width,
                               &x_drop, max_band_width, 1, &v));
    return SwgAlignment{v.alns[0].score, v.alns[0].xend, v.alns[0].yend, decode_ops(v.ops + v.alns[0].ops_off, v.alns[0].ops_len)};
  }
  thm_aligner* get() { return h_.get(); }
  void set_opts(const AlignOpts& opts) {
    thm_align_opts o = opts.c();
    check(thm_aligner_set_opts(h_.get(), &o));
  }

 private:
  void check(int rc) {
    if (rc != THM_OK) throw Error(rc, thm_last_error(h_.get()));
  }
  Index index_;
  std::shared_ptr<thm_aligner> h_;
};

enum class OutputFormat { Paf = THM_FMT_PAF, Sam = THM_FMT_SAM, Bam = THM_FMT_BAM };  // src/aln_writer.rs:16-21

// aligner::align_reads_from_file, src/aligner.rs:22-120
inline thm_run_stats align_reads_from_file(Aligner& aligner, const std::vector<std::string>& query_paths,
                                           const std::string& output_path, OutputFormat output_fmt,
                                           std::uint64_t batch_reads = 0, unsigned n_threads = 0) {
  std::vector<const char*> p;
  for (const auto& q : query_paths) p.push_back(q.c_str());
  thm_run_stats st;
  const int rc = thm_align_files(aligner.get(), p.data(), (std::uint32_t)p.size(), output_path.c_str(), (int)output_fmt,
                                 batch_reads, n_threads, &st);
  if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
  return st;
}

// wrapper::ThermiteAligner, src/wrapper.rs:20-123: index file in, one read per call, SAM records out.
// align_read returns the records as SAM text lines without the TX / GX / GN / RE tags: the reference converts the same
// text to rust_htslib Records and removes those four tags (src/wrapper.rs:126-141, cellranger adds its own);
// align_read_with_tags keeps them (the lines `thermite align` writes, src/aln_writer.rs:170-219).
class ThermiteAligner {
 public:
  explicit ThermiteAligner(const std::string& index_path, int device = 0)
      : index_(Index::load(index_path)), aligner_(index_, AlignOpts{}, device) {  // default settings, src/wrapper.rs:40-46
    thm_writer* w = nullptr;
    const int rc = thm_writer_create(index_.get(), THM_FMT_SAM, 1, &w);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    writer_.reset(w, thm_writer_free);
    thm_text t;
    thm_writer_header(w, &t);
    header_.assign((const char*)t.data, t.len);
  }
  std::vector<std::string> align_read(const std::string& name, const std::string& read, const std::string& qual) {
    std::vector<std::string> out = align_read_with_tags(name, read, qual);
    for (std::string& rec : out) {  // record.remove_aux(b"TX" | b"GX" | b"GN" | b"RE"), src/wrapper.rs:136-139
      std::string kept;
      std::size_t pos = 0;
      int field = 0;
      while (pos <= rec.size()) {
        std::size_t tab = rec.find('\t', pos);
        if (tab == std::string::npos) tab = rec.size();
        const bool drop = field >= 11 && tab - pos >= 3 && rec[pos + 2] == ':' &&
                          ((rec[pos] == 'T' && rec[pos + 1] == 'X') || (rec[pos] == 'G' && rec[pos + 1] == 'X') ||
                           (rec[pos] == 'G' && rec[pos + 1] == 'N') || (rec[pos] == 'R' && rec[pos + 1] == 'E'));
        if (!drop) {
          if (field) kept.push_back('\t');
          kept.append(rec, pos, tab - pos);
        }
        field++;
        pos = tab + 1;
      }
      rec.swap(kept);
    }
    return out;
  }
  std::vector<std::string> align_read_with_tags(const std::string& name, const std::string& read, const std::string& qual) {
    aligner_.set_opts(opts_);
    const std::uint64_t off[2] = {0, read.size()}, noff[2] = {0, name.size()};
    thm_batch_view v;
    int rc = thm_align_batch(aligner_.get(), (const std::uint8_t*)read.data(), off, 1, &v);
    if (rc != THM_OK) throw Error(rc, thm_last_error(aligner_.get()));
    thm_read_batch rb;
    rb.n_reads = 1;
    rb.n_bases = read.size();
    rb.bases = (const std::uint8_t*)read.data();
    rb.offsets = off;
    rb.quals = qual.size() == read.size() ? (const std::uint8_t*)qual.data() : nullptr;
    rb.names = (const std::uint8_t*)name.data();
    rb.name_off = noff;
    thm_text t;
    rc = thm_writer_format_batch(writer_.get(), &rb, &v, &t);
    if (rc != THM_OK) throw Error(rc, thm_last_error(nullptr));
    std::vector<std::string> out;
    const char* p = (const char*)t.data;
    const char* e = p + t.len;
    while (p < e) {
      const char* nl = p;
      while (nl < e && *nl != '\n') nl++;
      out.emplace_back(p, nl);  // omit the newline, src/wrapper.rs:132-133
      p = nl + 1;
    }
    return out;
  }
  // the size of the index file, src/wrapper.rs:104-108
  static std::size_t est_mem(const std::string& index_path) {
    FILE* f = fopen(index_path.c_str(), "rb");
    if (!f) throw Error(THM_ERR_IO, "Failed to open " + index_path);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fclose(f);
    return (std::size_t)n;
  }
  const AlignOpts& opts() const { return opts_; }
  AlignOpts& opts_mut() { return opts_; }
  const std::string& header_view() const { return header_; }  // SAM header text (a HeaderView in the reference)

 private:
  Index index_;
  Aligner aligner_;
  AlignOpts opts_;
  std::shared_ptr<thm_writer> writer_;
  std::string header_;
};

}  // namespace thermite
#endif
