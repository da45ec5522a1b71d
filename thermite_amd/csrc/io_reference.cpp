// io_reference.cpp -- FASTA + GTF ingestion and the on-disk index container
// (include/thermite_io.h).  Host only.
//
// Content follows Index::create_from_files, reference src/index.rs:52-223:
//   text  = per contig UPPER(seq) '$' UPPER(revcomp(seq)) '$'       (:67-101)
//   refs  = forward then reverse record per contig                   (:78-100)
//   exons / transcripts / gene spans lifted into concatenated coordinates,
//   reverse-strand features mapped into the revcomp copy and their exon order
//   reversed                                                         (:134-213)
// GTF rows are read the way thermite_amd/refdata.py reads them (the restatement
// of the `transcriptome` crate's behaviour; parity unpinned).
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/thermite_io.h"
#include "thermite_internal.h"

namespace {

using thm::set_global_error;

// line reader over a plain or gzip file (zlib reads both)
struct LineReader {
  gzFile f = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  bool open(const char* path) {
    f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    buf.resize(1 << 20);
    return true;
  }
  ~LineReader() {
    if (f) gzclose(f);
  }
  // next line without its terminator ('\n' or "\r\n"); false at end of file
  bool next(std::string& line) {
    line.clear();
    for (;;) {
      if (pos == end) {
        if (eof) return !line.empty();
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) {
          eof = true;
          return !line.empty();
        }
        pos = 0;
        end = (size_t)n;
      }
      const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
      if (nl) {
        line.append((const char*)(buf.data() + pos), nl);
        pos = (size_t)(nl - buf.data()) + 1;
        while (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
      }
      line.append((const char*)(buf.data() + pos), (const char*)(buf.data() + end));
      pos = end;
    }
  }
};

struct Contig {
  std::string name;
  std::string seq;
};

struct GtfTx {
  std::string id, chrom;
  uint32_t gene_idx = 0;
  bool strand = true;
  std::vector<std::pair<uint64_t, uint64_t>> exons;  // 0-based half-open, contig coordinates
};
struct GtfGene {
  std::string id, name;
};

int fail(int code, const std::string& msg) {
  set_global_error(msg);
  return code;
}

// key "value" pairs of a GTF attribute column: every maximal non-blank run that is
// followed by ` "` and a closing quote (what the pattern (\S+) "([^"]*)" finds)
void parse_attrs(const std::string& a, std::map<std::string, std::string>& out) {
  out.clear();
  size_t i = 0;
  const size_t n = a.size();
  auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
  while (i < n) {
    if (is_space(a[i])) {
      i++;
      continue;
    }
    size_t j = i;
    while (j < n && !is_space(a[j])) j++;
    if (j + 1 < n && a[j] == ' ' && a[j + 1] == '"') {
      const size_t q = a.find('"', j + 2);
      if (q != std::string::npos) {
        out[a.substr(i, j - i)] = a.substr(j + 2, q - (j + 2));
        i = q + 1;
        continue;
      }
    }
    i = j;
  }
}

int parse_fasta(const char* path, std::vector<Contig>& contigs) {
  LineReader r;
  if (!r.open(path)) return fail(THM_ERR_IO, std::string("cannot open FASTA ") + path);
  std::string line;
  bool have = false;
  while (r.next(line)) {
    if (!line.empty() && line[0] == '>') {
      contigs.emplace_back();
      // name = first word of the header (src/index.rs:69)
      const size_t sp = line.find(' ');
      contigs.back().name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
      have = true;
    } else if (!line.empty()) {
      if (!have) return fail(THM_ERR_FORMAT, std::string("FASTA does not start with '>': ") + path);
      contigs.back().seq += line;
    }
  }
  if (contigs.empty()) return fail(THM_ERR_FORMAT, std::string("no FASTA records in ") + path);
  return THM_OK;
}

int parse_gtf(const char* path, std::vector<GtfGene>& genes, std::vector<GtfTx>& txs) {
  LineReader r;
  if (!r.open(path)) return fail(THM_ERR_IO, std::string("cannot open GTF ") + path);
  std::unordered_map<std::string, uint32_t> gene_idx, tx_idx;
  std::map<std::string, std::string> a;
  std::string line;
  std::vector<std::string> col;
  auto need = [&](const char* key, std::string& v) -> bool {
    auto it = a.find(key);
    if (it == a.end()) return false;
    v = it->second;
    return true;
  };
  auto add_gene = [&](const std::string& gid) {
    if (gene_idx.find(gid) == gene_idx.end()) {
      gene_idx[gid] = (uint32_t)genes.size();
      std::string nm;
      if (!need("gene_name", nm)) nm = gid;
      genes.push_back({gid, nm});
    }
  };
  uint64_t lineno = 0;
  while (r.next(line)) {
    lineno++;
    if (line.empty() || line[0] == '#') continue;
    if (line.find_first_not_of(" \t") == std::string::npos) continue;
    col.clear();
    size_t s = 0;
    for (;;) {
      const size_t t = line.find('\t', s);
      col.push_back(line.substr(s, t == std::string::npos ? std::string::npos : t - s));
      if (t == std::string::npos) break;
      s = t + 1;
    }
    if (col.size() < 9) continue;
    const std::string& feat = col[2];
    const bool is_gene = feat == "gene", is_tx = feat == "transcript", is_exon = feat == "exon";
    if (!is_gene && !is_tx && !is_exon) continue;
    parse_attrs(col[8], a);
    const std::string where = std::string(path) + ":" + std::to_string(lineno);
    std::string gid, tid;
    if (is_gene) {
      if (!need("gene_id", gid)) return fail(THM_ERR_FORMAT, "gene row without gene_id at " + where);
      add_gene(gid);
    } else if (is_tx) {
      if (!need("gene_id", gid)) return fail(THM_ERR_FORMAT, "transcript row without gene_id at " + where);
      if (!need("transcript_id", tid)) return fail(THM_ERR_FORMAT, "transcript row without transcript_id at " + where);
      add_gene(gid);
      tx_idx[tid] = (uint32_t)txs.size();
      GtfTx t;
      t.id = tid;
      t.chrom = col[0];
      t.gene_idx = gene_idx[gid];
      t.strand = col[6] == "+";
      txs.push_back(std::move(t));
    } else {
      if (!need("transcript_id", tid)) return fail(THM_ERR_FORMAT, "exon row without transcript_id at " + where);
      auto it = tx_idx.find(tid);
      if (it == tx_idx.end()) return fail(THM_ERR_FORMAT, "exon of an undeclared transcript at " + where);
      char* e1 = nullptr;
      char* e2 = nullptr;
      const long long st = strtoll(col[3].c_str(), &e1, 10), en = strtoll(col[4].c_str(), &e2, 10);
      if (*e1 || *e2 || col[3].empty() || col[4].empty() || st < 1 || en < st)
        return fail(THM_ERR_FORMAT, "bad exon coordinates at " + where);
      txs[it->second].exons.emplace_back((uint64_t)st - 1, (uint64_t)en);  // 1-based inclusive -> 0-based half-open
    }
  }
  for (auto& t : txs) std::sort(t.exons.begin(), t.exons.end());
  return THM_OK;
}

uint8_t comp(uint8_t c) {
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return c;  // N
  }
}

// ---- index container ----
constexpr char MAGIC[8] = {'T', 'H', 'M', 'I', 'D', 'X', '0', '2'};

struct FileHeader {
  char magic[8];
  uint64_t n_text, n_refs, n_txs, n_exons, n_tx_seq, n_genes, n_contigs, names_bytes;
  uint64_t sizeof_ref, sizeof_tx, sizeof_exon, sizeof_span;
  uint64_t sa_bytes;  // bytes per suffix-array entry: 4, or 8 for an index with 64-bit coordinates
  uint64_t checksum;  // FNV-1a over every byte after the header
};

uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) {
    h ^= b[i];
    h *= 0x100000001b3ull;
  }
  return h;
}

void pack_names(const std::vector<std::string>& v, std::string& blob) {
  for (const auto& s : v) {
    blob += s;
    blob.push_back('\0');
  }
}

}  // namespace

// no exception leaves the C ABI (a huge or corrupt input must not terminate the host process)
template <class F>
static int32_t guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(THM_ERR_OOM, "out of memory");
  } catch (const std::exception& e) {
    return fail(THM_ERR_INTERNAL, std::string("internal error: ") + e.what());
  } catch (...) {
    return fail(THM_ERR_INTERNAL, "internal error");
  }
}

extern "C" {

int32_t thm_index_set_names(thm_index* ix, const char* const* contig_names, uint32_t n_contigs, const char* const* tx_ids,
                            uint32_t n_txs, const char* const* gene_ids, const char* const* gene_names, uint32_t n_genes) {
  if (!ix) return THM_ERR_INVALID_ARG;
  if ((n_contigs && !contig_names) || (n_txs && !tx_ids) || (n_genes && (!gene_ids || !gene_names))) return THM_ERR_INVALID_ARG;
  for (const thm_ref& r : ix->refs)
    if (r.name_id >= n_contigs) return fail(THM_ERR_INVALID_ARG, "thm_index_set_names: a ref's name_id has no name");
  if (n_txs != ix->txs.size() || n_genes != ix->genes.size())
    return fail(THM_ERR_INVALID_ARG, "thm_index_set_names: name counts do not match the tables");
  ix->contig_names.assign(contig_names, contig_names + n_contigs);
  ix->tx_ids.assign(tx_ids, tx_ids + n_txs);
  ix->gene_ids.assign(gene_ids, gene_ids + n_genes);
  ix->gene_names.assign(gene_names, gene_names + n_genes);
  return THM_OK;
}

static int32_t index_from_files_impl(const char* fasta_path, const char* gtf_path, thm_index** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!fasta_path || !gtf_path) return THM_ERR_INVALID_ARG;
  std::vector<Contig> contigs;
  std::vector<GtfGene> genes;
  std::vector<GtfTx> gtxs;
  int rc = parse_fasta(fasta_path, contigs);
  if (rc != THM_OK) return rc;
  rc = parse_gtf(gtf_path, genes, gtxs);
  if (rc != THM_OK) return rc;

  uint64_t n_total = 0;
  for (const auto& c : contigs) n_total += 2 * ((uint64_t)c.seq.size() + 1);
  std::vector<uint8_t> text(n_total);
  std::vector<thm_ref> refs(2 * contigs.size());
  std::map<std::pair<std::string, bool>, uint32_t> ref_of;
  uint64_t pos = 0;
  for (size_t ci = 0; ci < contigs.size(); ci++) {
    std::string& s = contigs[ci].seq;
    for (char& ch : s) {
      if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
      if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N')
        return fail(THM_ERR_FORMAT, "contig " + contigs[ci].name + " has bases outside ACGTN");
    }
    const uint64_t L = s.size();
    for (int strand = 1; strand >= 0; strand--) {
      thm_ref& r = refs[2 * ci + (strand ? 0 : 1)];
      memset(&r, 0, sizeof r);
      r.start_idx = pos;
      if (strand)
        memcpy(text.data() + pos, s.data(), L);
      else
        for (uint64_t i = 0; i < L; i++) text[pos + i] = comp((uint8_t)s[L - 1 - i]);
      pos += L;
      text[pos++] = '$';
      r.end_idx = pos;
      r.len = L;
      r.name_id = (uint32_t)ci;
      r.strand = (uint8_t)strand;
      ref_of[{contigs[ci].name, strand != 0}] = (uint32_t)(2 * ci + (strand ? 0 : 1));  // later duplicates win (HashMap insert, :77,92)
    }
  }
  // rank of each contig name in byte order (filter_overlapping sorts by ref_name, src/aligner.rs:322-327)
  std::vector<std::string> uniq;
  for (const auto& c : contigs) uniq.push_back(c.name);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  std::vector<uint32_t> name_rank(contigs.size());
  for (size_t ci = 0; ci < contigs.size(); ci++)
    name_rank[ci] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), contigs[ci].name) - uniq.begin());

  uint64_t n_exons = 0;
  for (const auto& t : gtxs) n_exons += t.exons.size();
  std::vector<thm_exon> exons(n_exons);
  std::vector<thm_tx> txs(gtxs.size());
  std::vector<thm_span> gspans(genes.size());
  for (auto& g : gspans) {
    g.start = n_total;  // (sa.bwt().len(), 0), src/index.rs:134
    g.end = 0;
  }
  std::vector<uint8_t> tx_seq;
  uint64_t eb = 0;
  for (size_t ti = 0; ti < gtxs.size(); ti++) {
    const GtfTx& t = gtxs[ti];
    auto it = ref_of.find({t.chrom, t.strand});
    if (it == ref_of.end()) return fail(THM_ERR_FORMAT, "transcript " + t.id + " is on a contig the FASTA does not have: " + t.chrom);
    if (t.exons.empty()) return fail(THM_ERR_FORMAT, "transcript " + t.id + " has no exons");
    const thm_ref& r = refs[it->second];
    const uint64_t s0 = r.start_idx, e1 = r.end_idx;
    const uint64_t tstart = t.exons.front().first, tend = t.exons.back().second;
    if (tend > r.len) return fail(THM_ERR_FORMAT, "transcript " + t.id + " runs past the end of " + t.chrom);
    uint64_t tx_start, tx_end;
    if (t.strand) {
      tx_start = tstart + s0;
      tx_end = tend + s0;
    } else {
      tx_start = e1 - 1 - tend;
      tx_end = e1 - 1 - tstart;
    }
    thm_span& g = gspans[t.gene_idx];
    g.start = std::min(g.start, tx_start);
    g.end = std::max(g.end, tx_end);
    thm_tx& x = txs[ti];
    memset(&x, 0, sizeof x);
    x.exon_begin = eb;
    x.n_exons = (uint32_t)t.exons.size();
    x.gene_idx = t.gene_idx;
    x.strand = t.strand ? 1 : 0;
    x.seq_off = tx_seq.size();
    const size_t ne = t.exons.size();
    for (size_t k = 0; k < ne; k++) {
      const auto& ex = t.strand ? t.exons[k] : t.exons[ne - 1 - k];  // reversed for '-' (src/index.rs:192-195)
      thm_exon& e = exons[eb + k];
      memset(&e, 0, sizeof e);
      if (t.strand) {
        e.start = ex.first + s0;
        e.end = ex.second + s0;
      } else {
        e.start = e1 - 1 - ex.second;
        e.end = e1 - 1 - ex.first;
      }
      e.tx_idx = (uint32_t)ti;
      tx_seq.insert(tx_seq.end(), text.begin() + (ptrdiff_t)e.start, text.begin() + (ptrdiff_t)e.end);
    }
    x.seq_len = tx_seq.size() - x.seq_off;
    eb += ne;
  }

  thm_index* ix = nullptr;
  rc = thm_index_create_in_memory(text.data(), n_total, refs.data(), (uint32_t)refs.size(), txs.data(), (uint32_t)txs.size(),
                                  exons.data(), n_exons, tx_seq.data(), tx_seq.size(), gspans.data(), (uint32_t)gspans.size(),
                                  name_rank.data(), (uint32_t)name_rank.size(), nullptr, &ix);
  if (rc != THM_OK) return rc;
  for (const auto& c : contigs) ix->contig_names.push_back(c.name);
  for (const auto& t : gtxs) ix->tx_ids.push_back(t.id);
  for (const auto& g : genes) {
    ix->gene_ids.push_back(g.id);
    ix->gene_names.push_back(g.name);
  }
  *out = ix;
  return THM_OK;
}

int32_t thm_index_tables(const thm_index* ix, thm_tables_view* v) {
  if (!ix || !v) return THM_ERR_INVALID_ARG;
  v->n_text = ix->n;
  v->text = ix->text.data();
  v->n_refs = (uint32_t)ix->refs.size();
  v->refs = ix->refs.data();
  v->n_txs = (uint32_t)ix->txs.size();
  v->txs = ix->txs.data();
  v->n_exons = ix->exons.size();
  v->exons = ix->exons.data();
  v->n_tx_seq = ix->tx_seq.size() >= 16 ? ix->tx_seq.size() - 16 : 0;  // without the staging pad
  v->tx_seq = ix->tx_seq.data();
  v->n_genes = (uint32_t)ix->genes.size();
  v->genes = ix->genes.data();
  v->name_rank = ix->name_rank.data();
  v->n_contigs = (uint32_t)ix->contig_names.size();
  return THM_OK;
}

const char* thm_index_contig_name(const thm_index* ix, uint32_t i) {
  return (ix && i < ix->contig_names.size()) ? ix->contig_names[i].c_str() : nullptr;
}
const char* thm_index_tx_id(const thm_index* ix, uint32_t i) { return (ix && i < ix->tx_ids.size()) ? ix->tx_ids[i].c_str() : nullptr; }
const char* thm_index_gene_id(const thm_index* ix, uint32_t i) {
  return (ix && i < ix->gene_ids.size()) ? ix->gene_ids[i].c_str() : nullptr;
}
const char* thm_index_gene_name(const thm_index* ix, uint32_t i) {
  return (ix && i < ix->gene_names.size()) ? ix->gene_names[i].c_str() : nullptr;
}

static int32_t index_save_impl(const thm_index* ix, const char* path) {
  if (!ix || !path) return THM_ERR_INVALID_ARG;
  std::string names;
  pack_names(ix->contig_names, names);
  pack_names(ix->tx_ids, names);
  pack_names(ix->gene_ids, names);
  pack_names(ix->gene_names, names);
  // name_rank per contig (the index keeps it per ref)
  std::vector<uint32_t> rank_of_name;
  for (const thm_ref& r : ix->refs) {
    if (r.name_id >= rank_of_name.size()) rank_of_name.resize(r.name_id + 1, 0);
  }
  for (size_t i = 0; i < ix->refs.size(); i++) rank_of_name[ix->refs[i].name_id] = ix->name_rank[i];
  FileHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, MAGIC, 8);
  h.n_text = ix->n;
  h.n_refs = ix->refs.size();
  h.n_txs = ix->txs.size();
  h.n_exons = ix->exons.size();
  h.n_tx_seq = ix->tx_seq.size() >= 16 ? ix->tx_seq.size() - 16 : 0;
  h.n_genes = ix->genes.size();
  h.n_contigs = rank_of_name.size();
  h.names_bytes = names.size();
  h.sizeof_ref = sizeof(thm_ref);
  h.sizeof_tx = sizeof(thm_tx);
  h.sizeof_exon = sizeof(thm_exon);
  h.sizeof_span = sizeof(thm_span);
  h.sa_bytes = ix->wide ? 8 : 4;
  struct Sec {
    const void* p;
    size_t n;
  };
  const uint64_t has_names = ix->contig_names.empty() ? 0 : 1;
  const Sec secs[] = {{ix->text.data(), (size_t)h.n_text},
                      {ix->wide ? (const void*)ix->sa64.data() : (const void*)ix->sa.data(), (size_t)h.n_text * (size_t)h.sa_bytes},
                      {ix->refs.data(), (size_t)h.n_refs * sizeof(thm_ref)},
                      {rank_of_name.data(), (size_t)h.n_contigs * 4},
                      {ix->txs.data(), (size_t)h.n_txs * sizeof(thm_tx)},
                      {ix->exons.data(), (size_t)h.n_exons * sizeof(thm_exon)},
                      {ix->tx_seq.data(), (size_t)h.n_tx_seq},
                      {ix->genes.data(), (size_t)h.n_genes * sizeof(thm_span)},
                      {&has_names, 8},
                      {names.data(), names.size()}};
  uint64_t ck = 0xcbf29ce484222325ull;
  for (const Sec& s : secs) ck = fnv1a(ck, s.p, s.n);
  h.checksum = ck;
  FILE* f = fopen(path, "wb");
  if (!f) return fail(THM_ERR_IO, std::string("cannot create ") + path);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  for (const Sec& s : secs) ok = ok && (s.n == 0 || fwrite(s.p, 1, s.n, f) == s.n);
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(THM_ERR_IO, std::string("short write to ") + path);
  return THM_OK;
}

static int32_t index_load_impl(const char* path, thm_index** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!path) return THM_ERR_INVALID_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(THM_ERR_IO, std::string("cannot open ") + path);
  FileHeader h;
  memset(&h, 0, sizeof h);
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, MAGIC, 8) != 0) {
    fclose(f);
    if (memcmp(h.magic, "THMIDX01", 8) == 0)
      return fail(THM_ERR_FORMAT, std::string(path) + " is a THMIDX01 index file (the format before the suffix-array width field): "
                                                       "rebuild it from the FASTA and GTF with this version");
    return fail(THM_ERR_FORMAT, std::string(path) + " is not a " + std::string(MAGIC, 8) + " index file");
  }
  fseek(f, 0, SEEK_END);
  const uint64_t have = (uint64_t)ftell(f);
  fseek(f, (long)sizeof h, SEEK_SET);
  // no count may exceed the file itself (so the products below cannot wrap around)
  const uint64_t counts[] = {h.n_text, h.n_refs, h.n_txs, h.n_exons, h.n_tx_seq, h.n_genes, h.n_contigs, h.names_bytes};
  bool sane = h.sizeof_ref == sizeof(thm_ref) && h.sizeof_tx == sizeof(thm_tx) && h.sizeof_exon == sizeof(thm_exon) &&
              h.sizeof_span == sizeof(thm_span) && (h.sa_bytes == 4 || h.sa_bytes == 8) && h.n_text != 0 &&
              (h.sa_bytes == 8 || h.n_text < 0x7FFFFFF0ull) && h.n_refs <= 0xFFFFFFFFull && h.n_txs <= 0xFFFFFFFFull &&
              h.n_genes <= 0xFFFFFFFFull && h.n_contigs <= h.n_refs;
  for (uint64_t c : counts) sane = sane && c <= have;
  if (!sane) {
    fclose(f);
    return fail(THM_ERR_FORMAT, std::string(path) + ": header fields out of range");
  }
  // the sizes the header announces must be what the file holds (before any allocation)
  const uint64_t expect = sizeof h + h.n_text * (1 + h.sa_bytes) + h.n_refs * sizeof(thm_ref) + h.n_contigs * 4 +
                          h.n_txs * sizeof(thm_tx) + h.n_exons * sizeof(thm_exon) + h.n_tx_seq + h.n_genes * sizeof(thm_span) + 8 +
                          h.names_bytes;
  if (have != expect) {
    fclose(f);
    return fail(THM_ERR_FORMAT, std::string(path) + ": file size does not match its header (truncated?)");
  }
  std::vector<uint8_t> text(h.n_text), tx_seq(h.n_tx_seq);
  std::vector<uint8_t> sa((size_t)h.n_text * (size_t)h.sa_bytes);
  std::vector<uint32_t> rank_of_name(h.n_contigs);
  std::vector<thm_ref> refs(h.n_refs);
  std::vector<thm_tx> txs(h.n_txs);
  std::vector<thm_exon> exons(h.n_exons);
  std::vector<thm_span> genes(h.n_genes);
  std::string names(h.names_bytes, '\0');
  uint64_t has_names = 0;
  uint64_t ck = 0xcbf29ce484222325ull;
  bool ok = true;
  auto rd = [&](void* p, size_t n) {
    if (!ok) return;
    if (n && fread(p, 1, n, f) != n) ok = false;
    if (ok) ck = fnv1a(ck, p, n);
  };
  rd(text.data(), text.size());
  rd(sa.data(), sa.size());
  rd(refs.data(), refs.size() * sizeof(thm_ref));
  rd(rank_of_name.data(), rank_of_name.size() * 4);
  rd(txs.data(), txs.size() * sizeof(thm_tx));
  rd(exons.data(), exons.size() * sizeof(thm_exon));
  rd(tx_seq.data(), tx_seq.size());
  rd(genes.data(), genes.size() * sizeof(thm_span));
  rd(&has_names, 8);
  rd(&names[0], names.size());
  fclose(f);
  if (!ok) return fail(THM_ERR_IO, std::string("short read from ") + path);
  if (ck != h.checksum) return fail(THM_ERR_FORMAT, std::string(path) + ": checksum mismatch");
  thm_index* ix = nullptr;
  int rc = thm_index_create_in_memory_ex(text.data(), h.n_text, refs.data(), (uint32_t)h.n_refs, txs.data(), (uint32_t)h.n_txs,
                                         exons.data(), h.n_exons, tx_seq.data(), h.n_tx_seq, genes.data(), (uint32_t)h.n_genes,
                                         rank_of_name.data(), (uint32_t)h.n_contigs, sa.data(), (uint32_t)h.sa_bytes,
                                         h.sa_bytes == 8 ? THM_INDEX_WIDE : 0u, &ix);
  if (rc != THM_OK) return rc;
  if (has_names) {
    std::vector<std::string> all;
    size_t s = 0;
    while (s < names.size()) {
      const size_t e = names.find('\0', s);
      if (e == std::string::npos) break;
      all.push_back(names.substr(s, e - s));
      s = e + 1;
    }
    const size_t want = (size_t)h.n_contigs + h.n_txs + 2 * h.n_genes;
    if (all.size() != want) {
      thm_index_free(ix);
      return fail(THM_ERR_FORMAT, std::string(path) + ": name table does not match the counts");
    }
    size_t k = 0;
    ix->contig_names.assign(all.begin() + k, all.begin() + k + h.n_contigs);
    k += h.n_contigs;
    ix->tx_ids.assign(all.begin() + k, all.begin() + k + h.n_txs);
    k += h.n_txs;
    ix->gene_ids.assign(all.begin() + k, all.begin() + k + h.n_genes);
    k += h.n_genes;
    ix->gene_names.assign(all.begin() + k, all.begin() + k + h.n_genes);
  }
  *out = ix;
  return THM_OK;
}

int32_t thm_index_create_from_files(const char* fasta_path, const char* gtf_path, thm_index** out) {
  return guarded([&] { return index_from_files_impl(fasta_path, gtf_path, out); });
}
int32_t thm_index_save(const thm_index* ix, const char* path) {
  return guarded([&] { return index_save_impl(ix, path); });
}
int32_t thm_index_load(const char* path, thm_index** out) {
  return guarded([&] { return index_load_impl(path, out); });
}

}  // extern "C"
