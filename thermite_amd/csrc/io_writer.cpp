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
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "io_internal.h"
#include "thermite_internal.h"

struct thm_writer {
  const thm_index* ix = nullptr;
  int format = THM_FMT_SAM;
  unsigned n_threads = 1;
  std::string header;                // SAM text header; for BAM: the BGZF-compressed BAM header
  std::string trailer;               // BAM: the BGZF end-of-file block
  std::vector<int32_t> sq_of_name;   // contig name_id -> index of its @SQ line (BAM refID)
  std::vector<std::string> chunk;    // per formatting thread
  std::vector<std::string> raw;      // BAM: uncompressed records per formatting thread
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
    // fast path: a stretch of plain matches (most of every op stream), eight bytes at a time
    if (p[i] == THM_OP_MATCH) {
      size_t j = i;
      uint64_t w8;
      while (j + 8 <= n && (memcpy(&w8, p + j, 8), w8 == 0)) j += 8;
      while (j < n && p[j] == THM_OP_MATCH) j++;
      const uint64_t m = (uint64_t)(j - i);
      if (cnt) {
        cnt->n_match += m;
        cnt->n_not_yclip += m;
      }
      if (prev_k == THM_OP_MATCH) {
        run += m;
      } else {
        flush();
        prev_k = THM_OP_MATCH;
        prev_clip = 0;
        run = m;
      }
      i = j;
      continue;
    }
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

// the counts of put_cigar without the text (PAF prints only them)
bool count_ops(const uint8_t* p, size_t n, OpCounts& cnt) {
  for (size_t i = 0; i < n;) {
    if (p[i] == THM_OP_MATCH) {
      size_t j = i;
      uint64_t w8;
      while (j + 8 <= n && (memcpy(&w8, p + j, 8), w8 == 0)) j += 8;
      while (j < n && p[j] == THM_OP_MATCH) j++;
      cnt.n_match += (uint64_t)(j - i);
      cnt.n_not_yclip += (uint64_t)(j - i);
      i = j;
      continue;
    }
    const int k = p[i++];
    if (k > THM_OP_YCLIP) return false;
    if (k >= THM_OP_XCLIP) {
      if (i + 4 > n) return false;
      i += 4;
    }
    cnt.n_subst += k == THM_OP_SUBST;
    cnt.n_not_yclip += k != THM_OP_YCLIP;
  }
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
        if (!count_ops(ops, al.ops_len, cnt)) {
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


// ---------------------------------------------------------------- BAM
// The reference writes BAM through noodles-bam 0.1.0 (`write_sam_record`, src/aligner.rs:69-76,98-108):
// the SAM record re-encoded in binary inside BGZF blocks.  Restated from the SAM/BAM specification
// (noodles is not in the reference checkout: parity unpinned; compressed bytes depend on the deflate
// implementation, so parity is defined on the decompressed stream).

inline void le32(std::string& s, uint32_t v) {
  char b[4] = {(char)(v & 0xff), (char)((v >> 8) & 0xff), (char)((v >> 16) & 0xff), (char)((v >> 24) & 0xff)};
  s.append(b, 4);
}
inline void le16(std::string& s, uint32_t v) {
  char b[2] = {(char)(v & 0xff), (char)((v >> 8) & 0xff)};
  s.append(b, 2);
}

// UCSC binning scheme, SAM specification section 5.3
inline uint32_t reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

struct SeqCode {
  uint8_t t[256];
  SeqCode() {
    for (int i = 0; i < 256; i++) t[i] = 15;
    const char* a = "=ACMGRSVTWYHKDBN";
    for (int i = 0; a[i]; i++) {
      t[(uint8_t)a[i]] = (uint8_t)i;
      if (a[i] >= 'A' && a[i] <= 'Z') t[(uint8_t)a[i] + 32] = (uint8_t)i;
    }
  }
};
const SeqCode SEQ_CODE;

// integer tag with the smallest type that holds the value (c C s S i I)
inline void bam_int_tag(std::string& s, const char* tag, int64_t v) {
  s.append(tag, 2);
  if (v >= 0) {
    if (v <= 0xff) {
      s.push_back('C');
      s.push_back((char)v);
    } else if (v <= 0xffff) {
      s.push_back('S');
      le16(s, (uint32_t)v);
    } else {
      s.push_back('I');
      le32(s, (uint32_t)v);
    }
  } else {
    if (v >= -128) {
      s.push_back('c');
      s.push_back((char)(int8_t)v);
    } else if (v >= -32768) {
      s.push_back('s');
      le16(s, (uint32_t)(uint16_t)(int16_t)v);
    } else {
      s.push_back('i');
      le32(s, (uint32_t)(int32_t)v);
    }
  }
}
inline void bam_str_tag(std::string& s, const char* tag, const std::string& v) {
  s.append(tag, 2);
  s.push_back('Z');
  s += v;
  s.push_back('\0');
}

// CIGAR of an op stream as BAM words; returns the reference length it consumes
bool bam_cigar(std::vector<uint32_t>& out, const uint8_t* p, size_t n, OpCounts* cnt, uint64_t& ref_len) {
  out.clear();
  ref_len = 0;
  static const uint32_t CODE[6] = {0 /*M*/, 0, 2 /*D*/, 1 /*I*/, 4 /*S*/, 3 /*N*/};
  int prev_k = -1;
  uint32_t prev_clip = 0;
  uint64_t run = 0;
  auto flush = [&]() {
    if (prev_k < 0) return;
    const uint64_t len = prev_k >= THM_OP_XCLIP ? (uint64_t)prev_clip : run;
    out.push_back((uint32_t)(len << 4) | CODE[prev_k]);
    if (prev_k == THM_OP_MATCH || prev_k == THM_OP_DEL || prev_k == THM_OP_YCLIP) ref_len += len;
  };
  for (size_t i = 0; i < n;) {
    // fast path: a stretch of plain matches (most of every op stream), eight bytes at a time
    if (p[i] == THM_OP_MATCH) {
      size_t j = i;
      uint64_t w8;
      while (j + 8 <= n && (memcpy(&w8, p + j, 8), w8 == 0)) j += 8;
      while (j < n && p[j] == THM_OP_MATCH) j++;
      const uint64_t m = (uint64_t)(j - i);
      if (cnt) {
        cnt->n_match += m;
        cnt->n_not_yclip += m;
      }
      if (prev_k == THM_OP_MATCH) {
        run += m;
      } else {
        flush();
        prev_k = THM_OP_MATCH;
        prev_clip = 0;
        run = m;
      }
      i = j;
      continue;
    }
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

void bam_seq_qual(std::string& s, const uint8_t* seq, size_t L, const uint8_t* qual, size_t QL, bool forward) {
  // (written through a pointer into storage appended once: a push_back per byte was a third of the encoder's time)
  const size_t at = s.size(), nb = (L + 1) / 2;
  s.resize(at + nb + L);
  uint8_t* w = reinterpret_cast<uint8_t*>(&s[at]);
  if (forward) {
    size_t i = 0;
    for (; i + 1 < L; i += 2) *w++ = (uint8_t)((SEQ_CODE.t[seq[i]] << 4) | SEQ_CODE.t[seq[i + 1]]);
    if (i < L) *w++ = (uint8_t)(SEQ_CODE.t[seq[i]] << 4);
  } else {
    size_t i = 0;
    for (; i + 1 < L; i += 2) *w++ = (uint8_t)((SEQ_CODE.t[COMP.t[seq[L - 1 - i]]] << 4) | SEQ_CODE.t[COMP.t[seq[L - 2 - i]]]);
    if (i < L) *w++ = (uint8_t)(SEQ_CODE.t[COMP.t[seq[L - 1 - i]]] << 4);
  }
  if (QL == 0) {
    memset(w, 0xff, L);
  } else if (forward) {
    for (size_t i = 0; i < L; i++) w[i] = (uint8_t)(qual[i] - 33);
  } else {
    for (size_t i = 0; i < L; i++) w[i] = (uint8_t)(qual[L - 1 - i] - 33);
  }
}


bool format_range_bam(const Ctx& c, const std::vector<int32_t>& sq_of_name, uint64_t r0, uint64_t r1, std::string& s,
                      std::string& err) {
  const thm_index* ix = c.ix;
  std::vector<uint32_t> cig;
  std::string txz, tags;
  s.reserve(s.size() + (size_t)(r1 - r0) * 320);
  for (uint64_t r = r0; r < r1; r++) {
    const uint8_t* name = c.reads->names + c.reads->name_off[r];
    const size_t name_len = (size_t)(c.reads->name_off[r + 1] - c.reads->name_off[r]);
    const uint8_t* seq = c.reads->bases + c.reads->offsets[r];
    const size_t L = (size_t)(c.reads->offsets[r + 1] - c.reads->offsets[r]);
    const uint8_t* qual = c.reads->quals ? c.reads->quals + c.reads->offsets[r] : nullptr;
    const size_t QL = qual ? L : 0;
    size_t qn = name_len;
    if (const void* sp = memchr(name, ' ', name_len)) qn = (size_t)((const uint8_t*)sp - name);
    if (qn > 254) {
      err = "read name longer than 254 bytes cannot be stored in BAM";
      return false;
    }
    const uint64_t a0 = c.res->read_aln_off[r], a1 = c.res->read_aln_off[r + 1];
    const uint64_t multimap = a1 - a0;
    auto fixed = [&](int32_t ref_id, int32_t pos, uint32_t mapq, uint32_t bin, uint32_t n_cig, uint32_t flag) {
      // the 32 fixed bytes behind block_size in one append (little-endian host, like every other table of this library)
      uint8_t b[32];
      auto p32 = [&](int at, uint32_t v) { memcpy(b + at, &v, 4); };
      auto p16 = [&](int at, uint32_t v) {
        b[at] = (uint8_t)(v & 0xff);
        b[at + 1] = (uint8_t)(v >> 8);
      };
      p32(0, (uint32_t)ref_id);
      p32(4, (uint32_t)pos);
      b[8] = (uint8_t)(qn + 1);
      b[9] = (uint8_t)mapq;
      p16(10, bin);
      p16(12, n_cig);
      p16(14, flag);
      p32(16, (uint32_t)L);
      p32(20, (uint32_t)-1);  // next refID
      p32(24, (uint32_t)-1);  // next pos
      p32(28, 0);             // tlen
      s.append((const char*)b, 32);
      s.append((const char*)name, qn);
      s.push_back('\0');
    };
    auto close_record = [&](size_t at) {
      const uint32_t bs = (uint32_t)(s.size() - at - 4);
      s[at] = (char)(bs & 0xff);
      s[at + 1] = (char)((bs >> 8) & 0xff);
      s[at + 2] = (char)((bs >> 16) & 0xff);
      s[at + 3] = (char)((bs >> 24) & 0xff);
    };
    if (multimap == 0) {
      const size_t at = s.size();
      le32(s, 0);
      fixed(-1, -1, 255, 4680, 0, 4);
      bam_seq_qual(s, seq, L, qual, QL, true);
      close_record(at);
      continue;
    }
    for (uint64_t a = a0; a < a1; a++) {
      const thm_aln& al = c.res->alns[a];
      if (al.ref_id >= ix->refs.size() || al.ops_off + al.ops_len > c.res->n_op_bytes) {
        err = "alignment record out of range";
        return false;
      }
      const thm_ref& ref = ix->refs[al.ref_id];
      OpCounts cnt;
      uint64_t ref_len = 0;
      if (!bam_cigar(cig, c.res->ops + al.ops_off, al.ops_len, &cnt, ref_len) || cig.size() > 0xffff) {
        err = "malformed op stream";
        return false;
      }
      const int64_t pos = (int64_t)al.ystart;
      const size_t at = s.size();
      le32(s, 0);
      fixed(sq_of_name[ref.name_id], (int32_t)pos, multimapq(multimap), reg2bin(pos, pos + (int64_t)std::max<uint64_t>(ref_len, 1)),
            (uint32_t)cig.size(), (al.strand ? 0u : 16u) | (al.primary ? 0u : 256u));
      s.append((const char*)cig.data(), 4 * cig.size());
      bam_seq_qual(s, seq, L, qual, QL, al.strand != 0);
      {  // the four integer tags through a small buffer (7 bytes each at most)
        std::string& t4 = tags;
        t4.clear();
        bam_int_tag(t4, "AS", al.score);
        bam_int_tag(t4, "NH", (int64_t)multimap);
        bam_int_tag(t4, "HI", (int64_t)(a - a0 + 1));
        bam_int_tag(t4, "nM", (int64_t)cnt.n_subst);
        s += t4;
      }
      if (al.aln_type == THM_ALN_EXONIC) {
        const uint32_t t = al.tx_or_gene_idx;
        if (t >= ix->txs.size() || al.tx_ops_off + al.tx_ops_len > c.res->n_op_bytes) {
          err = "transcript alignment out of range";
          return false;
        }
        const uint32_t g = ix->txs[t].gene_idx;
        txz = ix->tx_ids[t];
        txz += ",+";
        put_u64(txz, al.tx_ystart);
        txz.push_back(',');
        if (!put_cigar(txz, c.res->ops + al.tx_ops_off, al.tx_ops_len, nullptr)) {
          err = "malformed transcript op stream";
          return false;
        }
        bam_str_tag(s, "TX", txz);
        bam_str_tag(s, "GX", ix->gene_ids[g]);
        bam_str_tag(s, "GN", ix->gene_names[g]);
        s.append("REAE", 4);
      } else if (al.aln_type == THM_ALN_INTRONIC) {
        const uint32_t g = al.tx_or_gene_idx;
        if (g >= ix->gene_ids.size()) {
          err = "gene index out of range";
          return false;
        }
        bam_str_tag(s, "GX", ix->gene_ids[g]);
        bam_str_tag(s, "GN", ix->gene_names[g]);
        s.append("REAN", 4);
      } else {
        s.append("REAI", 4);
      }
      close_record(at);
    }
  }
  return true;
}

// BGZF: a series of gzip members of at most 64 KiB, each with a BC extra field giving its size.
// Deflate: csrc/io_deflate.cpp unless THM_BAM_LEVEL=0..9 asks for zlib at that level (the reference's writer is
// flate2 at level 6; parity is defined on the inflated stream, the compressed bytes depend on the implementation as it
// is).  At level 6 sixteen threads deflate some 3 M records a second -- a third of what the stages before deliver --
// and the records leave unsorted, for a sorter that rewrites them anyway.
bool bgzf_compress(const char* p, size_t n, std::string& out) {
  constexpr size_t BLOCK = 0xff00;
  static const int level = [] {
    const char* e = getenv("THM_BAM_LEVEL");
    const int v = e && *e ? atoi(e) : -1;
    return v < 0 || v > 9 ? -1 : v;
  }();
  std::vector<unsigned char> buf(std::max<size_t>(compressBound(BLOCK) + 64, thm::deflate_block_bound(BLOCK)));
  std::unique_ptr<thm::DeflateScratch> scratch;
  // one deflate state for all blocks of the call (deflateInit2 allocates and clears ~270 KB: per 64 KiB block that is
  // a tenth of the work), reset between the members
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (level >= 0) {
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  } else {
    scratch.reset(new thm::DeflateScratch());
  }
  bool ok = true;
  for (size_t off = 0; off < n || (n == 0 && off == 0); off += BLOCK) {
    const size_t len = std::min(BLOCK, n - off);
    size_t clen;
    if (level >= 0) {
      if (off && deflateReset(&zs) != Z_OK) {
        ok = false;
        break;
      }
      zs.next_in = (Bytef*)(p + off);
      zs.avail_in = (uInt)len;
      zs.next_out = buf.data();
      zs.avail_out = (uInt)buf.size();
      const int rc = deflate(&zs, Z_FINISH);
      clen = zs.total_out;
      if (rc != Z_STREAM_END) {
        ok = false;
        break;
      }
    } else {
      clen = thm::deflate_block((const uint8_t*)(p + off), len, buf.data(), *scratch);
    }
    const size_t bsize = clen + 25;  // whole block size - 1
    if (bsize > 0xffff) {
      ok = false;
      break;
    }
    const unsigned char hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                   (unsigned char)(bsize & 0xff), (unsigned char)(bsize >> 8)};
    out.append((const char*)hdr, 18);
    out.append((const char*)buf.data(), clen);
    le32(out, thm::crc32_fast(0, (const uint8_t*)(p + off), len));  // (carry-less multiplication: six times zlib's crc32)
    le32(out, (uint32_t)len);
    if (n == 0) break;
  }
  if (level >= 0) deflateEnd(&zs);
  return ok;
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
  if (w->format == THM_FMT_BAM) w->raw.resize(std::max<size_t>(w->raw.size(), T));
  auto work = [&](unsigned t) {
    std::string& s = w->chunk[t];
    s.clear();
    const uint64_t r0 = n * t / T, r1 = n * (t + 1) / T;
    if (w->format == THM_FMT_BAM) {
      std::string& raw = w->raw[t];
      raw.clear();
      ok[t] = format_range_bam(c, w->sq_of_name, r0, r1, raw, errs[t]) ? 1 : 0;
      if (ok[t] && !raw.empty() && !bgzf_compress(raw.data(), raw.size(), s)) {
        ok[t] = 0;
        errs[t] = "BGZF compression failed";
      }
    } else {
      ok[t] = format_range(c, r0, r1, s, errs[t]) ? 1 : 0;
    }
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
  if (format != THM_FMT_SAM && format != THM_FMT_PAF && format != THM_FMT_BAM) return THM_ERR_INVALID_ARG;
  if (ix->contig_names.empty() || ix->tx_ids.size() != ix->txs.size() || ix->gene_ids.size() != ix->genes.size())
    return fail(THM_ERR_INVALID_ARG, "the writer needs contig / transcript / gene names: thm_index_set_names or thm_index_create_from_files");
  thm_writer* w = new thm_writer();
  w->ix = ix;
  w->format = format;
  unsigned t = n_threads ? n_threads : std::min(16u, std::thread::hardware_concurrency());
  w->n_threads = std::max(1u, std::min(t, 32u));
  if (format == THM_FMT_SAM || format == THM_FMT_BAM) {
    // build_sam_header (:256-276): the reference sequences are collected into a map keyed by name,
    // so the forward and reverse Ref of a contig share one @SQ line (first-seen order)
    std::vector<std::pair<std::string, uint64_t>> sq;
    w->sq_of_name.assign(ix->contig_names.size(), -1);
    for (const thm_ref& r : ix->refs) {
      if (w->sq_of_name[r.name_id] >= 0) continue;
      int32_t k = -1;
      for (size_t i = 0; i < sq.size(); i++)
        if (sq[i].first == ix->contig_names[r.name_id]) k = (int32_t)i;  // same name under another id: same @SQ
      if (k < 0) {
        k = (int32_t)sq.size();
        sq.emplace_back(ix->contig_names[r.name_id], r.len);
      }
      w->sq_of_name[r.name_id] = k;
    }
    std::string text;
    for (const auto& q : sq) text += "@SQ\tSN:" + q.first + "\tLN:" + std::to_string(q.second) + "\n";
    text += "@PG\tID:thermite\n";
    if (format == THM_FMT_SAM) {
      w->header = text;
    } else {
      // bam::Writer::write_header + write_reference_sequences, src/aligner.rs:41-46
      std::string raw("BAM\1", 4);
      le32(raw, (uint32_t)text.size());
      raw += text;
      le32(raw, (uint32_t)sq.size());
      for (const auto& q : sq) {
        le32(raw, (uint32_t)q.first.size() + 1);
        raw += q.first;
        raw.push_back('\0');
        le32(raw, (uint32_t)q.second);
      }
      if (!bgzf_compress(raw.data(), raw.size(), w->header) || !bgzf_compress(nullptr, 0, w->trailer)) {
        delete w;
        return fail(THM_ERR_INTERNAL, "BGZF compression failed");
      }
    }
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

int32_t thm_writer_trailer(thm_writer* w, thm_text* out) {
  if (!w || !out) return THM_ERR_INVALID_ARG;
  out->data = (const uint8_t*)w->trailer.data();
  out->len = w->trailer.size();
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
