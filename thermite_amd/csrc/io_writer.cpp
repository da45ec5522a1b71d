// io_writer.cpp -- SAM / PAF rendering of aligned batches (include/thermite_io.h;
// SURVEY.md section 8f rank 3).  Restates reference src/aln_writer.rs:
//   PafEntry::new / Display                 :47-109   (12 columns + a trailing tab)
//   aln_to_sam_record                       :118-238  (flags, MAPQ, tags AS NH HI nM [TX GX GN] RE)
//   unmapped_sam_record                     :241-253
//   build_sam_header                        :256-276
//   to_noodles_cigar                        :279-323  (Match and Subst -> M, Xclip -> S, Yclip -> N)
//   multimapq                               :332-340
//   format_read_name / format_maybe_empty   :344-358
// and the record order of the writer loop, src/aligner.rs:58-115.  The text
// layout of a SAM line is noodles-sam 0.1.0's Display (Cargo.lock:773-776; not in
// the reference checkout: restated from the SAM specification, parity unpinned).
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

struct thm_writer {
  const thm_index* ix = nullptr;
  int format = THM_FMT_SAM;
  unsigned n_threads = 1;
  std::string header;
  std::vector<std::string> chunk;  // per formatting thread
  std::string out;
  std::string err;
};

namespace {

int fail(int code, const std::string& msg) {
  thm::set_global_error(msg);
  return code;
}

// bio::alphabets::dna::complement: ACGT + IUPAC codes in both cases, everything else unchanged
struct CompTable {
  uint8_t t[256];
  CompTable() {
    for (int i = 0; i < 256; i++) t[i] = (uint8_t)i;
    const char* a = "AGCTYRWSKMDVHBN";
    const char* b = "TCGARYWSMKHBDVN";
    for (int i = 0; a[i]; i++) {
      t[(uint8_t)a[i]] = (uint8_t)b[i];
      t[(uint8_t)a[i] + 32] = (uint8_t)(b[i] + 32);
    }
  }
};
const CompTable COMP;

inline void put_u64(std::string& s, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) s.push_back(tmp[--n]);
}
inline void put_i64(std::string& s, int64_t v) {
  if (v < 0) {
    s.push_back('-');
    put_u64(s, (uint64_t)(-(v + 1)) + 1);
  } else {
    put_u64(s, (uint64_t)v);
  }
}

// multimapq, src/aln_writer.rs:332-340: round(-10 log10(1 - 1/n)) for n = 2, 3, 4 is 3, 2, 1
inline unsigned multimapq(uint64_t n) {
  if (n <= 1) return 255;
  if (n >= 5) return 0;
  return n == 2 ? 3 : (n == 3 ? 2 : 1);
}

struct OpCounts {
  uint64_t n_match = 0, n_subst = 0, n_not_yclip = 0;
};

// to_noodles_cigar over a serialised op stream; appends "*" for an empty list.
// Consecutive equal ops form a run; two clips are equal only if their lengths are,
// and a run of clips is written with the clip's own length (:285-296, :309).
bool put_cigar(std::string& s, const uint8_t* p, size_t n, OpCounts* cnt) {
  if (n == 0) {
    s.push_back('*');
    return true;
  }
  static const char KIND[6] = {'M', 'M', 'D', 'I', 'S', 'N'};
  int prev_k = -1;
  uint32_t prev_clip = 0;
  uint64_t run = 0;
  auto flush = [&]() {
    if (prev_k < 0) return;
    put_u64(s, prev_k >= THM_OP_XCLIP ? (uint64_t)prev_clip : run);
    s.push_back(KIND[prev_k]);
  };
  for (size_t i = 0; i < n;) {
    int k = p[i++];
    uint32_t clip = 0;
    if (k > THM_OP_YCLIP) return false;
    if (k >= THM_OP_XCLIP) {
      if (i + 4 > n) return false;
      clip = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) | ((uint32_t)p[i + 3] << 24);
      i += 4;
    }
    if (cnt) {
      cnt->n_match += k == THM_OP_MATCH;
      cnt->n_subst += k == THM_OP_SUBST;
      cnt->n_not_yclip += k != THM_OP_YCLIP;
    }
    if (k == THM_OP_SUBST) k = THM_OP_MATCH;
    if (k == prev_k && (k < THM_OP_XCLIP || clip == prev_clip)) {
      run++;
    } else {
      flush();
      prev_k = k;
      prev_clip = clip;
      run = 1;
    }
  }
  flush();
  return true;
}

inline void put_maybe_empty(std::string& s, const uint8_t* p, size_t n) {
  if (n == 0)
    s.push_back('*');
  else
    s.append((const char*)p, n);
}

struct Ctx {
  const thm_index* ix;
  const thm_read_batch* reads;
  const thm_batch_view* res;
  int format;
};

bool format_range(const Ctx& c, uint64_t r0, uint64_t r1, std::string& s, std::string& err) {
  const thm_index* ix = c.ix;
  for (uint64_t r = r0; r < r1; r++) {
    const uint8_t* name = c.reads->names + c.reads->name_off[r];
    const size_t name_len = (size_t)(c.reads->name_off[r + 1] - c.reads->name_off[r]);
    const uint8_t* seq = c.reads->bases + c.reads->offsets[r];
    const size_t L = (size_t)(c.reads->offsets[r + 1] - c.reads->offsets[r]);
    const uint8_t* qual = c.reads->quals ? c.reads->quals + c.reads->offsets[r] : nullptr;
    const size_t QL = qual ? L : 0;
    // format_read_name: up to the first space (:344-349)
    size_t qn = name_len;
    if (const void* sp = memchr(name, ' ', name_len)) qn = (size_t)((const uint8_t*)sp - name);
    const uint64_t a0 = c.res->read_aln_off[r], a1 = c.res->read_aln_off[r + 1];
    const uint64_t multimap = a1 - a0;
    if (multimap == 0) {
      if (c.format == THM_FMT_SAM) {  // unmapped_sam_record; PAF writes nothing (src/aligner.rs:58-81)
        s.append((const char*)name, qn);
        s.append("\t4\t*\t0\t255\t*\t*\t0\t0\t");
        put_maybe_empty(s, seq, L);
        s.push_back('\t');
        put_maybe_empty(s, qual, QL);
        s.push_back('\n');
      }
      continue;
    }
    for (uint64_t a = a0; a < a1; a++) {
      const thm_aln& al = c.res->alns[a];
      if (al.ref_id >= ix->refs.size() || al.ops_off + al.ops_len > c.res->n_op_bytes) {
        err = "alignment record out of range";
        return false;
      }
      const thm_ref& ref = ix->refs[al.ref_id];
      const std::string& rname = ix->contig_names[ref.name_id];
      const uint8_t* ops = c.res->ops + al.ops_off;
      OpCounts cnt;
      if (c.format == THM_FMT_PAF) {
        std::string dummy;
        if (!put_cigar(dummy, ops, al.ops_len, &cnt)) {
          err = "malformed op stream";
          return false;
        }
        s.append((const char*)name, name_len);  // PAF carries the whole id (:95)
        s.push_back('\t');
        put_u64(s, L);
        s.push_back('\t');
        put_u64(s, al.xstart);
        s.push_back('\t');
        put_u64(s, al.xend);
        s.push_back('\t');
        s.push_back(al.strand ? '+' : '-');
        s.push_back('\t');
        s += rname;
        s.push_back('\t');
        put_u64(s, al.ylen);
        s.push_back('\t');
        put_u64(s, al.ystart);
        s.push_back('\t');
        put_u64(s, al.yend);
        s.push_back('\t');
        put_u64(s, cnt.n_match);
        s.push_back('\t');
        put_u64(s, cnt.n_not_yclip);
        s.push_back('\t');
        put_u64(s, multimapq(multimap));
        s.append("\t\n");
        continue;
      }
      // ---- SAM ----
      s.append((const char*)name, qn);
      s.push_back('\t');
      put_u64(s, (al.strand ? 0u : 16u) | (al.primary ? 0u : 256u));
      s.push_back('\t');
      s += rname;
      s.push_back('\t');
      put_u64(s, al.ystart + 1);  // 1-based (:231-234)
      s.push_back('\t');
      put_u64(s, multimapq(multimap));
      s.push_back('\t');
      if (!put_cigar(s, ops, al.ops_len, &cnt)) {
        err = "malformed op stream";
        return false;
      }
      s.append("\t*\t0\t0\t");
      if (L == 0) {
        s.push_back('*');
      } else if (al.strand) {
        s.append((const char*)seq, L);
      } else {
        const size_t at = s.size();
        s.resize(at + L);
        for (size_t i = 0; i < L; i++) s[at + i] = (char)COMP.t[seq[L - 1 - i]];
      }
      s.push_back('\t');
      if (QL == 0) {
        s.push_back('*');
      } else if (al.strand) {
        s.append((const char*)qual, QL);
      } else {
        const size_t at = s.size();
        s.resize(at + QL);
        for (size_t i = 0; i < QL; i++) s[at + i] = (char)qual[QL - 1 - i];
      }
      s.append("\tAS:i:");
      put_i64(s, al.score);
      s.append("\tNH:i:");
      put_u64(s, multimap);
      s.append("\tHI:i:");
      put_u64(s, a - a0 + 1);
      s.append("\tnM:i:");
      put_u64(s, cnt.n_subst);
      if (al.aln_type == THM_ALN_EXONIC) {
        const uint32_t t = al.tx_or_gene_idx;
        if (t >= ix->txs.size() || al.tx_ops_off + al.tx_ops_len > c.res->n_op_bytes) {
          err = "transcript alignment out of range";
          return false;
        }
        const uint32_t g = ix->txs[t].gene_idx;
        s.append("\tTX:Z:");
        s += ix->tx_ids[t];
        s.append(",+");
        put_u64(s, al.tx_ystart);
        s.push_back(',');
        if (!put_cigar(s, c.res->ops + al.tx_ops_off, al.tx_ops_len, nullptr)) {
          err = "malformed transcript op stream";
          return false;
        }
        s.append("\tGX:Z:");
        s += ix->gene_ids[g];
        s.append("\tGN:Z:");
        s += ix->gene_names[g];
        s.append("\tRE:A:E");
      } else if (al.aln_type == THM_ALN_INTRONIC) {
        const uint32_t g = al.tx_or_gene_idx;
        if (g >= ix->gene_ids.size()) {
          err = "gene index out of range";
          return false;
        }
        s.append("\tGX:Z:");
        s += ix->gene_ids[g];
        s.append("\tGN:Z:");
        s += ix->gene_names[g];
        s.append("\tRE:A:N");
      } else {
        s.append("\tRE:A:I");
      }
      s.push_back('\n');
    }
  }
  return true;
}

}  // namespace

namespace thm {

int writer_format_chunks(thm_writer* w, const thm_read_batch* reads, const thm_batch_view* res,
                         std::vector<const std::string*>& chunks) {
  chunks.clear();
  if (!w || !reads || !res) return THM_ERR_INVALID_ARG;
  if (reads->n_reads != res->n_reads) return fail(THM_ERR_INVALID_ARG, "thm_writer_format_batch: reads and results differ in n_reads");
  if (reads->n_reads && (!reads->offsets || !reads->name_off || !reads->names || !res->read_aln_off)) return THM_ERR_INVALID_ARG;
  const uint64_t n = reads->n_reads;
  const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(w->n_threads, (n + 4095) / 4096));
  w->chunk.resize(std::max<size_t>(w->chunk.size(), T));
  Ctx c{w->ix, reads, res, w->format};
  std::vector<std::string> errs(T);
  std::vector<char> ok(T, 1);
  auto work = [&](unsigned t) {
    std::string& s = w->chunk[t];
    s.clear();
    const uint64_t r0 = n * t / T, r1 = n * (t + 1) / T;
    ok[t] = format_range(c, r0, r1, s, errs[t]) ? 1 : 0;
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < T; t++)
    if (!ok[t]) return fail(THM_ERR_INTERNAL, "thm_writer_format_batch: " + errs[t]);
  for (unsigned t = 0; t < T; t++) chunks.push_back(&w->chunk[t]);
  return THM_OK;
}

}  // namespace thm

extern "C" {

int32_t thm_writer_create(const thm_index* ix, int32_t format, uint32_t n_threads, thm_writer** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!ix) return THM_ERR_INVALID_ARG;
  if (format == THM_FMT_BAM) return fail(THM_ERR_UNSUPPORTED, "BAM output is not built; write SAM and convert");
  if (format != THM_FMT_SAM && format != THM_FMT_PAF) return THM_ERR_INVALID_ARG;
  if (ix->contig_names.empty() || ix->tx_ids.size() != ix->txs.size() || ix->gene_ids.size() != ix->genes.size())
    return fail(THM_ERR_INVALID_ARG, "the writer needs contig / transcript / gene names: thm_index_set_names or thm_index_create_from_files");
  thm_writer* w = new thm_writer();
  w->ix = ix;
  w->format = format;
  unsigned t = n_threads ? n_threads : std::min(16u, std::thread::hardware_concurrency());
  w->n_threads = std::max(1u, std::min(t, 32u));
  if (format == THM_FMT_SAM) {
    // build_sam_header (:256-276): the reference sequences are collected into a map keyed by name,
    // so the forward and reverse Ref of a contig share one @SQ line (first-seen order)
    std::vector<char> seen(ix->contig_names.size(), 0);
    for (const thm_ref& r : ix->refs) {
      if (seen[r.name_id]) continue;
      seen[r.name_id] = 1;
      w->header += "@SQ\tSN:" + ix->contig_names[r.name_id] + "\tLN:" + std::to_string(r.len) + "\n";
    }
    w->header += "@PG\tID:thermite\n";
  }
  *out = w;
  return THM_OK;
}

void thm_writer_free(thm_writer* w) { delete w; }

int32_t thm_writer_header(thm_writer* w, thm_text* out) {
  if (!w || !out) return THM_ERR_INVALID_ARG;
  out->data = (const uint8_t*)w->header.data();
  out->len = w->header.size();
  return THM_OK;
}

int32_t thm_writer_format_batch(thm_writer* w, const thm_read_batch* reads, const thm_batch_view* res, thm_text* out) {
  if (!out) return THM_ERR_INVALID_ARG;
  std::vector<const std::string*> chunks;
  const int rc = thm::writer_format_chunks(w, reads, res, chunks);
  if (rc != THM_OK) return rc;
  size_t total = 0;
  for (const std::string* c : chunks) total += c->size();
  w->out.clear();
  w->out.reserve(total);
  for (const std::string* c : chunks) w->out += *c;
  out->data = (const uint8_t*)w->out.data();
  out->len = w->out.size();
  return THM_OK;
}

}  // extern "C"
