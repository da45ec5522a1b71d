// thermite_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of thermite's seed-and-extend hot path, written to follow the
// reference statement by statement so that it can serve as the parity checker
// for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it (see thermite_oracle.h).
//
// Every function cites the reference lines it restates (paths relative to the
// reference repository root).  Pieces that live in third-party crates absent
// from /root/reference are restated from their published algorithms and are
// tagged [bio 0.37.1, recalled] -- PARITY UNPINNED for those (seed order,
// interval-tree iteration order); everything from src/swg.rs, src/aligner.rs
// and src/txome.rs is pinned by the reference's own known-answer tests
// (tests/test_oracle_kats.py).
#include "thermite_oracle.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

typedef uint64_t usize;

// bio::alignment::pairwise::MIN_SCORE [bio 0.37.1, recalled]; SURVEY Appendix A:
// any value far below -(L+bw)*2 that cannot overflow behaves identically.
constexpr int32_t MIN_SCORE = -858993459;

enum : uint8_t { OP_MATCH = 0, OP_SUBST = 1, OP_DEL = 2, OP_INS = 3, OP_XCLIP = 4, OP_YCLIP = 5 };

// bio::alignment::AlignmentOperation
struct Op {
  uint8_t kind;
  usize n;  // only for Xclip / Yclip
  bool operator==(const Op& o) const { return kind == o.kind && ((kind < OP_XCLIP) || n == o.n); }
};
inline Op op(uint8_t k) { return Op{k, 0}; }

// bio::alignment::Alignment (mode omitted: always Custom on this path)
struct Alignment {
  int32_t score = 0;
  usize ystart = 0, xstart = 0, yend = 0, xend = 0, ylen = 0, xlen = 0;
  std::vector<Op> operations;
};

// bio::alignment::pairwise::Scoring::from_scores(gap_open, gap_extend, match, mismatch)
struct Scoring {
  int32_t gap_open, gap_extend, match, mismatch;
  inline int32_t score(uint8_t a, uint8_t b) const { return a == b ? match : mismatch; }
};

void serialize_ops(const std::vector<Op>& ops, std::vector<uint8_t>& out) {
  for (const Op& o : ops) {
    out.push_back(o.kind);
    if (o.kind >= OP_XCLIP) {
      uint32_t n = (uint32_t)o.n;
      for (int b = 0; b < 4; b++) out.push_back((uint8_t)(n >> (8 * b)));
    }
  }
}
bool deserialize_ops(const uint8_t* p, usize len, std::vector<Op>& out) {
  usize i = 0;
  while (i < len) {
    uint8_t k = p[i++];
    if (k > OP_YCLIP) return false;
    Op o{k, 0};
    if (k >= OP_XCLIP) {
      if (i + 4 > len) return false;
      uint32_t n = 0;
      for (int b = 0; b < 4; b++) n |= (uint32_t)p[i + b] << (8 * b);
      o.n = n;
      i += 4;
    }
    out.push_back(o);
  }
  return true;
}

// ===========================================================================
// src/swg.rs:6-241  SwgExtend
// ===========================================================================
struct SwgExtend {
  std::vector<int32_t> D, C, R;
  std::vector<uint8_t> trace;  // AlignmentOperation per cell (kind only)
  Scoring scoring;
  usize max_band_width;
  // instrumentation (not in the reference)
  usize n_cells = 0, n_cols = 0, n_calls = 0, n_phase1_breaks = 0;
  usize n_win_bytes = 0;  // reference / transcript window bytes of the targets extended (bookkeeping, not the reference's)
  bool fault = false;  // the reference would have panicked (out-of-bounds trace)

  // src/swg.rs:17-26
  SwgExtend(usize max_bw, Scoring s)
      : D(max_bw * 2 + 1, 0), C(max_bw * 2 + 1, 0), R(max_bw * 2 + 1, 0), scoring(s), max_band_width(max_bw) {}

  // src/swg.rs:210-217
  void set_trace(usize j, usize i, uint8_t o) {
    usize w = max_band_width * 2 + 1;
    if (trace.size() <= j * w) trace.resize(trace.size() + w, OP_MATCH);
    if (j * w + i >= trace.size()) {  // index panic in the reference
      fault = true;
      return;
    }
    trace[j * w + i] = o;
  }
  // src/swg.rs:220-223
  uint8_t get_trace(usize j, usize i) {
    usize w = max_band_width * 2 + 1;
    if (j * w + i >= trace.size()) {
      fault = true;
      return OP_MATCH;
    }
    return trace[j * w + i];
  }
  // src/swg.rs:226-240
  inline std::pair<int32_t, uint8_t> triple_max(int32_t d, int32_t c, int32_t r, bool m) const {
    int32_t score = std::max(std::max(d, c), r);
    uint8_t dir;
    if (score == d)
      dir = m ? OP_MATCH : OP_SUBST;
    else if (score == c)
      dir = OP_DEL;
    else
      dir = OP_INS;
    return {score, dir};
  }

  // src/swg.rs:170-207
  std::vector<Op> trace_back(usize i, usize j, usize len, usize band_width) {
    std::vector<Op> traceback;
    traceback.reserve(i + j + 4);
    if (i < len) traceback.push_back(Op{OP_XCLIP, len - i});
    while (i > 0 || j > 0) {
      usize band_idx = i - (j > band_width ? j - band_width : 0);  // i - j.saturating_sub(bw)
      uint8_t t = get_trace(j, band_idx);
      traceback.push_back(op(t));
      switch (t) {
        case OP_MATCH:
        case OP_SUBST:
          if (i == 0 || j == 0) { fault = true; return traceback; }  // usize underflow panic
          i -= 1;
          j -= 1;
          break;
        case OP_INS:
          if (i == 0) { fault = true; return traceback; }
          i -= 1;
          break;
        case OP_DEL:
          if (j == 0) { fault = true; return traceback; }
          j -= 1;
          break;
        default:
          fault = true;  // unreachable!()
          return traceback;
      }
    }
    std::reverse(traceback.begin(), traceback.end());
    return traceback;
  }

  // src/swg.rs:31-167
  Alignment extend(const uint8_t* x, usize xlen, const uint8_t* y, usize ylen, usize band_width, int32_t x_drop) {
    n_calls++;
    Alignment a;
    a.ylen = ylen;
    a.xlen = xlen;
    // :39-55
    if (xlen == 0 || ylen == 0) {
      if (xlen > 0) a.operations.push_back(Op{OP_XCLIP, xlen});
      return a;
    }
    const int32_t ge = scoring.gap_extend, go = scoring.gap_open;
    usize w = band_width * 2 + 1;
    int32_t max_score = 0;
    usize max_i = 0, max_j = 0;

    // :62-71 leftmost column
    D[0] = 0;
    C[0] = 0;
    R[0] = 0;
    set_trace(0, 0, OP_INS);
    for (usize i = 1; i < w; i++) {
      C[i] = MIN_SCORE;
      R[i] = (int32_t)i * ge + go;
      D[i] = R[i];
      set_trace(0, i, OP_INS);
    }

    // :75-113 band anchored at row 0
    usize p1_end = std::min(band_width, ylen);
    for (usize j = 1; j <= p1_end; j++) {
      int32_t band_max = MIN_SCORE;
      int32_t prev_D = MIN_SCORE;
      usize rows = std::min(w, xlen + 1);
      for (usize i = 0; i < rows; i++) {
        C[i] = std::max(C[i] + ge, D[i] + ge + go);
        R[i] = (i == 0) ? MIN_SCORE : std::max(R[i - 1] + ge, D[i - 1] + ge + go);
        int32_t d = (i == 0) ? MIN_SCORE : prev_D + scoring.score(x[i - 1], y[j - 1]);
        prev_D = D[i];
        auto md = triple_max(d, C[i], R[i], i > 0 && x[i - 1] == y[j - 1]);
        D[i] = md.first;
        set_trace(j, i, md.second);
        if (D[i] > max_score) {
          max_score = D[i];
          max_i = i;
          max_j = j;
        }
        band_max = std::max(band_max, D[i]);
      }
      n_cells += rows;
      n_cols++;
      if (band_max < max_score - x_drop) {
        n_phase1_breaks++;
        break;  // leaves ONLY this loop (:110-116)
      }
    }

    // :116-154 band slides down one row per column
    for (usize j = band_width + 1; j < ylen + 1; j++) {
      int32_t band_max = MIN_SCORE;
      usize i0 = j - band_width;
      usize i1 = std::min(j - band_width + w, xlen + 1);
      for (usize i = i0; i < i1; i++) {
        usize band_idx = i - (j - band_width);
        C[band_idx] = (band_idx >= w - 1) ? MIN_SCORE : std::max(C[band_idx + 1] + ge, D[band_idx + 1] + ge + go);
        R[band_idx] = (band_idx == 0) ? MIN_SCORE : std::max(R[band_idx - 1] + ge, D[band_idx - 1] + ge + go);
        int32_t d = D[band_idx] + scoring.score(x[i - 1], y[j - 1]);
        auto md = triple_max(d, C[band_idx], R[band_idx], x[i - 1] == y[j - 1]);
        D[band_idx] = md.first;
        set_trace(j, band_idx, md.second);
        if (D[band_idx] > max_score) {
          max_score = D[band_idx];
          max_i = i;
          max_j = j;
        }
        band_max = std::max(band_max, D[band_idx]);
      }
      if (i1 > i0) n_cells += i1 - i0;
      n_cols++;
      if (band_max < max_score - x_drop) break;
    }

    a.score = max_score;
    a.yend = max_j;
    a.xend = max_i;
    a.operations = trace_back(max_i, max_j, xlen, band_width);
    return a;
  }
};

// ===========================================================================
// src/index.rs:383-399  Mem, Ref ; src/txome.rs:9-69 data model
// ===========================================================================
struct Mem {
  usize ref_idx, query_idx, len;
};
struct Ref {
  uint32_t name_id;
  bool strand;
  usize len, start_idx, end_idx;
};
struct Exon {
  usize start, end, tx_idx;
  usize len() const { return end - start; }
};
struct Tx {
  bool strand;
  std::vector<Exon> exons;
  const uint8_t* seq;
  usize seq_len;
  usize gene_idx;
};
enum AlnTypeKind : uint8_t { EXONIC = 0, INTRONIC = 1, INTERGENIC = 2 };
struct GenomeAlignment {
  Alignment gx_aln;
  AlnTypeKind aln_type;
  Alignment tx_aln;  // Exonic only
  usize tx_idx = 0;  // Exonic
  usize gene_idx = 0;  // Intronic
  usize ref_id;        // stands for ref_name
  bool strand;
  bool primary;
};

// ===========================================================================
// bio::data_structures::interval_tree::IntervalTree  [bio 0.37.1, recalled]
// AVL tree keyed by interval start; find() is a stack traversal.
// ===========================================================================
struct ITNode {
  usize start, end;  // interval
  usize value;
  usize max;
  int64_t height;
  std::unique_ptr<ITNode> left, right;
  ITNode(usize s, usize e, usize v) : start(s), end(e), value(v), max(e), height(1) {}

  void update_height() {
    int64_t lh = left ? left->height : 0, rh = right ? right->height : 0;
    height = 1 + std::max(lh, rh);
  }
  void update_max() {
    max = end;
    if (left && max < left->max) max = left->max;
    if (right && max < right->max) max = right->max;
  }
  static void swap_interval_data(ITNode& a, ITNode& b) {
    std::swap(a.start, b.start);
    std::swap(a.end, b.end);
    std::swap(a.value, b.value);
  }
  void rotate_right() {
    std::unique_ptr<ITNode> new_root = std::move(left);
    std::unique_ptr<ITNode> t1 = std::move(new_root->left);
    std::unique_ptr<ITNode> t2 = std::move(new_root->right);
    std::unique_ptr<ITNode> t3 = std::move(right);
    swap_interval_data(*this, *new_root);
    new_root->left = std::move(t2);
    new_root->right = std::move(t3);
    new_root->update_height();
    new_root->update_max();
    left = std::move(t1);
    right = std::move(new_root);
    update_height();
    update_max();
  }
  void rotate_left() {
    std::unique_ptr<ITNode> new_root = std::move(right);
    std::unique_ptr<ITNode> t1 = std::move(left);
    std::unique_ptr<ITNode> t2 = std::move(new_root->left);
    std::unique_ptr<ITNode> t3 = std::move(new_root->right);
    swap_interval_data(*this, *new_root);
    new_root->left = std::move(t1);
    new_root->right = std::move(t2);
    new_root->update_height();
    new_root->update_max();
    right = std::move(t3);
    left = std::move(new_root);
    update_height();
    update_max();
  }
  void repair() {
    int64_t lh = left ? left->height : 0, rh = right ? right->height : 0;
    if (std::llabs(lh - rh) <= 1) {
      update_height();
      update_max();
    } else if (rh > lh) {
      {
        ITNode* r = right.get();
        int64_t rlh = r->left ? r->left->height : 0, rrh = r->right ? r->right->height : 0;
        if (rlh > rrh) r->rotate_right();
      }
      rotate_left();
    } else {
      {
        ITNode* l = left.get();
        int64_t lrh = l->right ? l->right->height : 0, llh = l->left ? l->left->height : 0;
        if (lrh > llh) l->rotate_left();
      }
      rotate_right();
    }
  }
  void insert(usize s, usize e, usize v) {
    if (s <= start) {
      if (left)
        left->insert(s, e, v);
      else
        left.reset(new ITNode(s, e, v));
    } else if (right) {
      right->insert(s, e, v);
    } else {
      right.reset(new ITNode(s, e, v));
    }
    repair();
  }
};
struct IntervalTree {
  std::unique_ptr<ITNode> root;
  void insert(usize s, usize e, usize v) {
    if (root)
      root->insert(s, e, v);
    else
      root.reset(new ITNode(s, e, v));
  }
  // IntervalTreeIterator::next, collected: pop; if q.start < node.max push left;
  // if q.end > node.start push right and yield on overlap.
  template <class F>
  void find(usize qs, usize qe, F&& yield) const {
    std::vector<const ITNode*> nodes;
    if (root) nodes.push_back(root.get());
    while (!nodes.empty()) {
      const ITNode* c = nodes.back();
      nodes.pop_back();
      if (qs < c->max) {
        if (c->left) nodes.push_back(c->left.get());
        if (qe > c->start) {
          if (c->right) nodes.push_back(c->right.get());
          if (qs < c->end && c->start < qe) yield(c->value);
        }
      }
    }
  }
};

// ===========================================================================
// bio FM / FMD index  [bio 0.37.1, recalled]; built as in src/index.rs:103-111
// ===========================================================================
// bio::alphabets::dna::complement
static uint8_t COMP[256];
static bool comp_init = [] {
  for (int i = 0; i < 256; i++) COMP[i] = (uint8_t)i;
  const char* a = "ACGTRYSWKMBDHVN";
  const char* b = "TGCAYRSWMKVHDBN";
  for (int i = 0; a[i]; i++) {
    COMP[(uint8_t)a[i]] = (uint8_t)b[i];
    COMP[(uint8_t)(a[i] + 32)] = (uint8_t)(b[i] + 32);
  }
  return true;
}();

struct BiInterval {
  usize lower, lower_rev, size, match_size;
  BiInterval swapped() const { return BiInterval{lower_rev, lower, size, match_size}; }
};

struct FMD {
  usize n = 0;
  std::vector<uint8_t> bwt;
  std::vector<usize> less;  // 257 entries: # symbols < c
  uint32_t occ_k = 128;
  int sym_of[256];
  // Every position and rank is a usize, as in the reference (src/index.rs:103-111: divsufsort64 -> Vec<usize>; a
  // GRCh38-sized text has 6.2 G symbols).  The suffix array handed in may be 32- or 64-bit.
  std::vector<usize> occ;  // [n/k + 1][6]
  // SampledSuffixArray
  uint32_t sa_s = 32;
  std::vector<usize> sa_sample;
  std::unordered_map<usize, usize> extra_rows;
  // the plain suffix array is kept only for the matching-statistics cross-check (not for texts of billions of symbols)
  std::vector<usize> sa_full;

  template <class SA>
  void build(const uint8_t* text, usize n_, const SA* sa, uint32_t sa_rate, uint32_t occ_rate, bool keep_full = true) {
    n = n_;
    occ_k = occ_rate;
    sa_s = sa_rate;
    for (int i = 0; i < 256; i++) sym_of[i] = -1;
    const char* syms = "$ACGNT";
    for (int i = 0; i < 6; i++) sym_of[(uint8_t)syms[i]] = i;
    // bwt(): src/index.rs:106
    bwt.resize(n);
    for (usize r = 0; r < n; r++) bwt[r] = sa[r] > 0 ? text[sa[r] - 1] : (uint8_t)'$';
    // less(): src/index.rs:109
    std::vector<usize> cnt(257, 0);
    for (usize r = 0; r < n; r++) cnt[bwt[r]]++;
    less.assign(258, 0);
    for (int c = 1; c < 258; c++) less[c] = less[c - 1] + cnt[c - 1];
    // Occ::new: src/index.rs:110
    usize rows = n / occ_k + 1;
    occ.assign(rows * 6, 0);
    usize cur[6] = {0, 0, 0, 0, 0, 0};
    for (usize i = 0; i < n; i++) {
      int s = sym_of[bwt[i]];
      if (s >= 0) cur[s]++;
      if (i % occ_k == 0)
        for (int t = 0; t < 6; t++) occ[(i / occ_k) * 6 + t] = cur[t];
    }
    // RawSuffixArray::sample: src/index.rs:111
    sa_sample.clear();
    for (usize i = 0; i < n; i++) {
      if (i % sa_s == 0)
        sa_sample.push_back((usize)sa[i]);
      else if (bwt[i] == '$')
        extra_rows[i] = (usize)sa[i];
    }
    sa_full.clear();
    if (keep_full) sa_full.assign(sa, sa + n);
  }
  inline usize less_of(uint8_t a) const { return less[a]; }
  // Occ::get: occurrences of a in bwt[0..=r]
  inline usize occ_get(usize r, uint8_t a) const {
    int s = sym_of[a];
    if (s < 0) return 0;
    usize lo = r / occ_k;
    usize c = occ[lo * 6 + s];
    const uint8_t* p = bwt.data();
    for (usize i = lo * occ_k + 1; i <= r; i++) c += (p[i] == a);
    return c;
  }
  // SampledSuffixArray::get
  usize sa_get(usize index) const {
    usize pos = index, offset = 0;
    for (;;) {
      if (pos % sa_s == 0) return sa_sample[pos / sa_s] + offset;
      uint8_t c = bwt[pos];
      if (c == '$') return extra_rows.at(pos) + offset;
      pos = less_of(c) + occ_get(pos - 1, c);
      offset++;
    }
  }
  // FMDIndex::init_interval_with
  BiInterval init_interval_with(uint8_t a) const {
    uint8_t ca = COMP[a];
    usize lower = less_of(a);
    return BiInterval{lower, less_of(ca), less[(usize)a + 1] - lower, 1};
  }
  // FMDIndex::backward_ext
  BiInterval backward_ext(const BiInterval& iv, uint8_t a) const {
    usize s = 0, o = 0, l = iv.lower_rev;
    static const uint8_t order[] = {'$', 'T', 'G', 'C', 'N', 'A', 't', 'g', 'c', 'n', 'a'};
    if (iv.size == 0) return BiInterval{less_of(a), l, 0, iv.match_size + 1};
    for (uint8_t b : order) {
      l += s;
      o = (iv.lower == 0) ? 0 : occ_get(iv.lower - 1, b);
      s = occ_get(iv.lower + iv.size - 1, b) - o;
      if (b == a) break;
    }
    bool in_order = false;
    for (uint8_t b : order) in_order |= (b == a);
    if (!in_order) s = 0;  // symbol outside the FMD alphabet: never matches (out of contract in bio)
    return BiInterval{less_of(a) + o, l, s, iv.match_size + 1};
  }
  // FMDIndex::forward_ext
  BiInterval forward_ext(const BiInterval& iv, uint8_t a) const { return backward_ext(iv.swapped(), COMP[a]).swapped(); }

  struct Smem {
    BiInterval iv;
    usize pos, len;
  };
  // FMDIndex::smems(pattern, i, l)
  void smems(const uint8_t* pattern, usize plen, usize i, usize l, std::vector<Smem>& matches) const {
    std::vector<std::pair<BiInterval, usize>> curr, prev;
    usize match_len = 0;
    BiInterval interval = init_interval_with(pattern[i]);
    if (interval.size != 0) match_len += 1;
    for (usize p = i + 1; p < plen; p++) {
      uint8_t a = pattern[p];
      BiInterval fwd = forward_ext(interval, a);
      if (interval.size != fwd.size) curr.push_back({interval, match_len});
      if (fwd.size == 0) break;
      interval = fwd;
      match_len += 1;
    }
    curr.push_back({interval, match_len});
    std::reverse(curr.begin(), curr.end());
    std::swap(curr, prev);
    int64_t j = (int64_t)plen;
    for (int64_t k = (int64_t)i - 1; k >= -1; k--) {
      uint8_t a = (k == -1) ? (uint8_t)'$' : pattern[k];
      curr.clear();
      int64_t last_size = -1;
      for (auto& pr : prev) {
        const BiInterval& iv = pr.first;
        usize mlen = pr.second;
        BiInterval ext = backward_ext(iv, a);
        if ((ext.size == 0 || k == -1) && curr.empty() && k < j && mlen >= l) {
          j = k;
          matches.push_back(Smem{iv, (usize)(k + 1), mlen});
        }
        if (ext.size != 0 && (int64_t)ext.size != last_size) {
          last_size = (int64_t)ext.size;
          curr.push_back({ext, mlen + 1});
        }
      }
      if (curr.empty()) break;
      std::swap(curr, prev);
    }
  }
  // FMDIndex::all_smems(pattern, l)
  void all_smems(const uint8_t* pattern, usize plen, usize l, std::vector<Smem>& out) const {
    usize i0 = 0;
    while (i0 < plen) {
      std::vector<Smem> cur;
      smems(pattern, plen, i0, l, cur);
      usize next_i0 = i0 + 1;
      for (auto& m : cur)
        if (m.pos + m.len > next_i0) next_i0 = m.pos + m.len;
      i0 = next_i0;
      out.insert(out.end(), cur.begin(), cur.end());
    }
  }
};

}  // namespace

// ===========================================================================
// src/index.rs:39-44 Index (our own in-memory form)
// ===========================================================================
struct orc_index {
  std::vector<uint8_t> text;
  std::vector<Ref> refs;
  std::vector<Tx> txs;
  std::vector<uint8_t> tx_seq;
  std::vector<uint32_t> name_rank;
  FMD fmd;
  IntervalTree exon_to_tx, gene_intervals;
  usize n_genes = 0;

  // src/index.rs:287-290  refs.partition_point(|x| x.end_idx <= idx)
  usize ref_of(usize idx) const {
    usize lo = 0, hi = refs.size();
    while (lo < hi) {
      usize mid = (lo + hi) / 2;
      if (refs[mid].end_idx <= idx)
        lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  }
  // src/index.rs:304-323; the reverse-strand branch re-derives the slice from
  // the forward contig exactly as the reference does
  std::vector<uint8_t> seq_slice(usize start, usize end) const {
    usize ri = ref_of(start);
    const Ref& cur = refs[ri];
    std::vector<uint8_t> out;
    if (cur.strand) {
      out.assign(text.begin() + start, text.begin() + end);
    } else {
      const Ref& prev = refs[ri - 1];
      usize chrom_start = cur.end_idx - 1 - end;
      usize chrom_end = cur.end_idx - 1 - start;
      const uint8_t* s = text.data() + prev.start_idx;  // prev_ref.seq
      out.resize(chrom_end - chrom_start);
      for (usize t = 0; t < out.size(); t++) out[t] = COMP[s[chrom_end - 1 - t]];  // dna::revcomp
    }
    return out;
  }
  // src/index.rs:228-255
  std::vector<Mem> all_smems(const uint8_t* query, usize qlen, usize min_seed_len, usize* n_smems) const {
    std::vector<Mem> mems;
    std::vector<FMD::Smem> ivs;
    fmd.all_smems(query, qlen, min_seed_len, ivs);
    if (n_smems) *n_smems = ivs.size();
    for (auto& s : ivs) {
      // interval.0.forward().occ(&self.sa): rows lower..lower+size ascending
      for (usize r = s.iv.lower; r < s.iv.lower + s.iv.size; r++)
        mems.push_back(Mem{fmd.sa_get(r), s.pos, s.len});
    }
    std::stable_sort(mems.begin(), mems.end(), [](const Mem& a, const Mem& b) { return a.len < b.len; });
    std::reverse(mems.begin(), mems.end());
    return mems;
  }
  // Same list from the implementation-independent definition:
  // MS[i] by binary search on the plain suffix array; SMEM iff end[i] > end[i-1];
  // emission order rebuilt from the all_smems i0 walk.
  std::vector<Mem> all_smems_ms(const uint8_t* q, usize L, usize k, usize* n_smems) const {
    const std::vector<usize>& sa = fmd.sa_full;  // (kept by orc_index_create unless told otherwise)
    const usize n = text.size();
    struct S {
      usize pos, len, lo, hi;
    };
    std::vector<S> sm;
    usize prev_end = 0;
    bool prev_valid = false;
    for (usize i = 0; i < L; i++) {
      usize lo = 0, hi = n, d = 0;
      while (i + d < L) {
        uint8_t c = q[i + d];
        // narrow [lo,hi) to suffixes with text[sa+d] == c
        usize a = lo, b = hi;
        while (a < b) {
          usize m = (a + b) / 2;
          usize p = (usize)sa[m] + d;
          int t = p < n ? text[p] : -1;
          if (t < (int)c)
            a = m + 1;
          else
            b = m;
        }
        usize nlo = a;
        b = hi;
        while (a < b) {
          usize m = (a + b) / 2;
          usize p = (usize)sa[m] + d;
          int t = p < n ? text[p] : -1;
          if (t <= (int)c)
            a = m + 1;
          else
            b = m;
        }
        if (a == nlo || c == '$') break;
        lo = nlo;
        hi = a;
        d++;
      }
      usize end = i + d;
      bool is_smem = d > 0 && (!prev_valid || end > prev_end);
      if (is_smem && d >= k) sm.push_back(S{i, d, lo, hi});
      prev_end = end;
      prev_valid = true;
    }
    if (n_smems) *n_smems = sm.size();
    // emission order: i0 walk; group = smems covering i0, descending start
    std::vector<S> emitted;
    usize i0 = 0;
    std::vector<char> used(sm.size(), 0);
    while (i0 < L) {
      usize next = i0 + 1;
      for (usize t = sm.size(); t-- > 0;) {
        if (!used[t] && sm[t].pos <= i0 && i0 < sm[t].pos + sm[t].len) {
          used[t] = 1;
          emitted.push_back(sm[t]);
          next = std::max(next, sm[t].pos + sm[t].len);
        }
      }
      i0 = next;
    }
    std::vector<Mem> mems;
    for (auto& s : emitted)
      for (usize r = s.lo; r < s.hi; r++) mems.push_back(Mem{sa[r], s.pos, s.len});
    std::stable_sort(mems.begin(), mems.end(), [](const Mem& a, const Mem& b) { return a.len < b.len; });
    std::reverse(mems.begin(), mems.end());
    return mems;
  }
};

namespace {

// ===========================================================================
// src/txome.rs:77-160
// ===========================================================================
inline bool intersect(usize a0, usize a1, usize b0, usize b1) {  // :77-79
  return (a0 >= b0 && a0 < b1) || (b0 >= a0 && b0 < a1);
}
inline usize sat_sub(usize a, usize b) { return a > b ? a - b : 0; }

// :82-103 ; returns false where the reference hits unreachable!()
bool lift_mem_to_tx(const Mem& mem, const std::vector<Exon>& exons, Mem* out) {
  usize exon_sum = 0;
  for (const Exon& exon : exons) {
    if (intersect(mem.ref_idx, mem.ref_idx + mem.len, exon.start, exon.end)) {
      usize start = sat_sub(mem.ref_idx, exon.start) + exon_sum;
      usize start_offset = sat_sub(exon.start, mem.ref_idx);
      usize end = std::min(mem.ref_idx + mem.len, exon.end) - exon.start + exon_sum;
      *out = Mem{start, mem.query_idx + start_offset, end - start};
      return true;
    }
    exon_sum += exon.len();
  }
  return false;
}

// :110-160 ; returns false on the reference's assert_eq!(i, tx_aln.yend) or an index panic
bool lift_tx_to_gx(const Alignment& tx_aln, const std::vector<Exon>& exons, Alignment* out) {
  Alignment aln = tx_aln;
  aln.operations.clear();
  usize i = tx_aln.ystart;
  usize op_idx = 0;
  usize exon_sum = 0;
  usize exon_idx = 0;
  while (true) {
    if (exon_idx >= exons.size()) return false;
    if (!(exon_sum + exons[exon_idx].len() <= i)) break;
    exon_sum += exons[exon_idx].len();
    exon_idx += 1;
  }
  usize diff = i - exon_sum;
  aln.ystart = exons[exon_idx].start + diff;
  while (op_idx < tx_aln.operations.size()) {
    if (exon_idx + 1 < exons.size() && exon_sum + exons[exon_idx].len() <= i) {
      exon_sum += exons[exon_idx].len();
      exon_idx += 1;
      aln.operations.push_back(Op{OP_YCLIP, exons[exon_idx].start - exons[exon_idx - 1].end});
    }
    uint8_t k = tx_aln.operations[op_idx].kind;
    if (k == OP_MATCH || k == OP_SUBST || k == OP_DEL) i += 1;
    aln.operations.push_back(tx_aln.operations[op_idx]);
    op_idx += 1;
  }
  if (i != tx_aln.yend) return false;
  diff = i - exon_sum;
  aln.yend = exons[exon_idx].start + diff;
  *out = aln;
  return true;
}

// ===========================================================================
// src/aligner.rs:352-449
// ===========================================================================
// :352-407
Alignment extend_left_right(const uint8_t* ref_seq, usize ref_len, const Mem& hit, const uint8_t* read, usize read_len,
                            SwgExtend& swg, usize band_width, int32_t x_drop) {
  const uint8_t* xr = read + hit.query_idx + hit.len;
  usize xr_len = read_len - (hit.query_idx + hit.len);
  const uint8_t* yr = ref_seq + hit.ref_idx + hit.len;
  usize yr_len = ref_len - (hit.ref_idx + hit.len);
  Alignment right_aln = swg.extend(xr, xr_len, yr, yr_len, band_width, x_drop);

  std::vector<uint8_t> x(read, read + hit.query_idx);
  std::reverse(x.begin(), x.end());
  usize y0 = sat_sub(hit.ref_idx, read_len + band_width);
  std::vector<uint8_t> y(ref_seq + y0, ref_seq + hit.ref_idx);
  std::reverse(y.begin(), y.end());
  Alignment left_aln = swg.extend(x.data(), x.size(), y.data(), y.size(), band_width, x_drop);

  Alignment a;
  a.ystart = hit.ref_idx - left_aln.yend;
  a.yend = hit.ref_idx + hit.len + right_aln.yend;
  a.xstart = hit.query_idx - left_aln.xend;
  a.xend = hit.query_idx + hit.len + right_aln.xend;
  a.score = left_aln.score + (int32_t)hit.len + right_aln.score;
  a.operations.reserve(left_aln.operations.size() + hit.len + right_aln.operations.size());
  for (usize t = left_aln.operations.size(); t-- > 0;) a.operations.push_back(left_aln.operations[t]);
  for (usize t = 0; t < hit.len; t++) a.operations.push_back(op(OP_MATCH));
  for (auto& o : right_aln.operations) a.operations.push_back(o);
  a.ylen = ref_len;
  a.xlen = read_len;
  return a;
}

// :410-426
void extend_seed_match(const uint8_t* ref_seq, usize ref_len, Mem& hit, const uint8_t* read, usize read_len) {
  while (hit.ref_idx + hit.len < ref_len && hit.query_idx + hit.len < read_len &&
         ref_seq[hit.ref_idx + hit.len] == read[hit.query_idx + hit.len])
    hit.len += 1;
  while (hit.ref_idx > 0 && hit.query_idx > 0 && ref_seq[hit.ref_idx - 1] == read[hit.query_idx - 1]) {
    hit.ref_idx -= 1;
    hit.query_idx -= 1;
    hit.len += 1;
  }
}

// :429-449
Alignment concat_to_chr_aln(const orc_index& index, Alignment aln) {
  const Ref& r = index.refs[index.ref_of(aln.ystart)];
  if (r.strand) {
    aln.ystart -= r.start_idx;
    aln.yend -= r.start_idx;
    aln.ylen = r.len;
  } else {
    usize ys = r.len - (aln.yend - r.start_idx);
    usize ye = r.len - (aln.ystart - r.start_idx);
    aln.ystart = ys;
    aln.yend = ye;
    aln.ylen = r.len;
    std::reverse(aln.operations.begin(), aln.operations.end());
  }
  return aln;
}

struct Counters {
  uint64_t c[16] = {0};
};

// :198-314
GenomeAlignment align_seed_hit(const orc_index& index, const uint8_t* read, usize read_len, const Mem& hit,
                               SwgExtend& swg, usize band_width, int32_t x_drop, bool* fault) {
  usize ref_id = index.ref_of(hit.ref_idx);
  const Ref& aln_ref = index.refs[ref_id];

  Alignment gx_aln;
  {
    usize seq_start = std::max(sat_sub(hit.ref_idx, read_len + band_width), aln_ref.start_idx);
    usize seq_end = std::min(hit.ref_idx + hit.len + read_len + band_width, aln_ref.end_idx - 1);
    std::vector<uint8_t> ref_seq = index.seq_slice(seq_start, seq_end);
    swg.n_win_bytes += seq_end - seq_start;
    Mem rel = hit;
    rel.ref_idx -= seq_start;
    gx_aln = extend_left_right(ref_seq.data(), ref_seq.size(), rel, read, read_len, swg, band_width, x_drop);
    gx_aln.ystart += seq_start;
    gx_aln.yend += seq_start;
  }

  bool have_best = false;
  usize best_tx_idx = 0;
  Alignment best_tx_aln;
  std::vector<usize> tx_idxs;
  index.exon_to_tx.find(hit.ref_idx, hit.ref_idx + hit.len, [&](usize v) { tx_idxs.push_back(v); });
  for (usize tx_idx : tx_idxs) {
    const Tx& tx = index.txs[tx_idx];
    Mem tx_seed;
    if (!lift_mem_to_tx(hit, tx.exons, &tx_seed)) {
      *fault = true;
      break;
    }
    // the part of the transcript the two extensions can reach (the reference passes the whole tx.seq and lets
    // extend_left_right slice it, :360-375): [seed - (L + bw), seed end + L + bw + 1), the term 2(L+bw)+len of SURVEY.md 8(d)
    swg.n_win_bytes += std::min(tx.seq_len, tx_seed.ref_idx + tx_seed.len + read_len + band_width + 1) -
                       sat_sub(tx_seed.ref_idx, read_len + band_width);
    extend_seed_match(tx.seq, tx.seq_len, tx_seed, read, read_len);
    Alignment tx_aln = extend_left_right(tx.seq, tx.seq_len, tx_seed, read, read_len, swg, band_width, x_drop);
    int32_t tx_aln_score = tx_aln.score;
    if (!have_best || tx_aln_score > best_tx_aln.score) {
      have_best = true;
      best_tx_idx = tx_idx;
      best_tx_aln = std::move(tx_aln);
    }
    if (tx_aln_score >= (int32_t)(read_len * 1)) break;
  }

  GenomeAlignment ga;
  ga.ref_id = ref_id;
  ga.strand = aln_ref.strand;
  ga.primary = false;
  if (have_best && best_tx_aln.score >= gx_aln.score) {
    Alignment lifted;
    if (!lift_tx_to_gx(best_tx_aln, index.txs[best_tx_idx].exons, &lifted)) *fault = true;
    ga.gx_aln = concat_to_chr_aln(index, std::move(lifted));
    ga.aln_type = EXONIC;
    ga.tx_aln = std::move(best_tx_aln);
    ga.tx_idx = best_tx_idx;
  } else {
    bool any = false;
    usize first_gene = 0;
    index.gene_intervals.find(gx_aln.ystart, gx_aln.yend, [&](usize v) {
      if (!any) {
        any = true;
        first_gene = v;
      }
    });
    ga.gx_aln = concat_to_chr_aln(index, std::move(gx_aln));
    if (!any) {
      ga.aln_type = INTERGENIC;
    } else {
      ga.aln_type = INTRONIC;
      ga.gene_idx = first_gene;
    }
  }
  return ga;
}

// :317-349
std::vector<GenomeAlignment> filter_overlapping(const orc_index* index, std::vector<GenomeAlignment> alns,
                                                const uint32_t* direct_rank = nullptr) {
  if (alns.empty()) return alns;
  auto rank = [&](const GenomeAlignment& a) -> uint32_t {
    if (direct_rank) return direct_rank[a.ref_id];
    return index->name_rank[index->refs[a.ref_id].name_id];
  };
  std::stable_sort(alns.begin(), alns.end(), [&](const GenomeAlignment& a, const GenomeAlignment& b) {
    uint32_t ra = rank(a), rb = rank(b);
    if (ra != rb) return ra < rb;
    if (a.strand != b.strand) return (int)a.strand < (int)b.strand;
    return a.gx_aln.ystart < b.gx_aln.ystart;
  });
  usize max_end = 0;
  std::vector<GenomeAlignment> res;
  res.reserve(alns.size());
  for (auto& aln : alns) {
    if (aln.gx_aln.ystart >= max_end || rank(aln) != rank(res.back()) || aln.strand != res.back().strand) {
      max_end = aln.gx_aln.yend;
      res.push_back(std::move(aln));
    } else {
      GenomeAlignment& curr = res.back();
      if (aln.gx_aln.score > curr.gx_aln.score) curr = std::move(aln);
      max_end = std::max(max_end, curr.gx_aln.yend);
    }
  }
  return res;
}

// Rust `f32 as i32`: truncate toward zero, saturate, NaN -> 0
inline int32_t f32_as_i32(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}

// :123-190
std::vector<GenomeAlignment> align_read(const orc_index& index, const uint8_t* read_in, usize read_len,
                                        const orc_opts& opts, Counters& cnt, bool* fault) {
  std::vector<uint8_t> read(read_in, read_in + read_len);
  for (auto& c : read)
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);  // to_ascii_uppercase

  usize n_smems = 0;
  std::vector<Mem> mems = index.all_smems(read.data(), read_len, opts.min_seed_len, &n_smems);
  cnt.c[7] += n_smems;
  cnt.c[8] += mems.size();

  std::vector<GenomeAlignment> gx_alns;
  volatile float prod = opts.min_aln_score_percent * (float)read_len;  // binary32 product
  int32_t min_aln_score = std::max(f32_as_i32(prod), opts.min_aln_score);
  int32_t max_aln_score = min_aln_score;
  usize band_width = sat_sub(read_len, (usize)(int64_t)min_aln_score);  // `as usize` of a negative wraps
  if (min_aln_score < 0) band_width = 0;
  usize x_drop = band_width;
  const int32_t range = (int32_t)opts.multimap_score_range;

  Scoring scoring{-1, -1, 1, -1};
  SwgExtend swg(band_width, scoring);

  // diagnostic for tools/tail_diag.py (ORC_TRACE_STATE=<min hits>): at which hits the loop-carried state moves
  static const long trace_min = getenv("ORC_TRACE_STATE") ? atol(getenv("ORC_TRACE_STATE")) : 0;
  const bool trace_state = trace_min > 0 && (long)mems.size() >= trace_min;
  std::string trace_line;
  usize hit_no = 0;
  for (const Mem& hit : mems) {
    hit_no++;
    GenomeAlignment gx_aln = align_seed_hit(index, read.data(), read_len, hit, swg, band_width, (int32_t)x_drop, fault);
    if (!opts.intron_mode && gx_aln.aln_type != EXONIC) continue;
    int32_t sc = gx_aln.gx_aln.score;
    if (sc < opts.min_aln_score || sc < min_aln_score || sc < max_aln_score - range) continue;
    usize lim = sc < 0 ? 0 : sat_sub(read_len + opts.multimap_score_range, (usize)sc);
    if (trace_state && (std::min(band_width, lim) != band_width || std::min(x_drop, lim) != x_drop || sc > max_aln_score))
      trace_line += " " + std::to_string(hit_no) + ":" + std::to_string(sc);
    band_width = std::min(band_width, lim);
    x_drop = std::min(x_drop, lim);
    max_aln_score = std::max(max_aln_score, sc);
    gx_alns.push_back(std::move(gx_aln));
  }
  if (trace_state) fprintf(stderr, "ORC_TRACE_STATE hits=%zu accepted=%zu state moved at (hit:score)%s\n", mems.size(), gx_alns.size(), trace_line.c_str());
  cnt.c[9] += swg.n_calls;
  cnt.c[13] += swg.n_win_bytes;
  cnt.c[10] += swg.n_cells;
  cnt.c[11] += swg.n_cols;
  if (swg.fault) *fault = true;

  gx_alns.erase(std::remove_if(gx_alns.begin(), gx_alns.end(),
                               [&](const GenomeAlignment& a) { return !(a.gx_aln.score >= max_aln_score - range); }),
                gx_alns.end());
  gx_alns = filter_overlapping(&index, std::move(gx_alns));
  std::stable_sort(gx_alns.begin(), gx_alns.end(),
                   [](const GenomeAlignment& a, const GenomeAlignment& b) { return -a.gx_aln.score < -b.gx_aln.score; });
  if (!gx_alns.empty()) gx_alns.front().primary = true;
  return gx_alns;
}

}  // namespace

// ===========================================================================
// result container + C ABI
// ===========================================================================
struct orc_result {
  uint64_t n = 0;
  std::vector<uint64_t> offsets;
  std::vector<orc_aln> alns;
  std::vector<orc_mem> mems;
  std::vector<orc_swg_aln> swg;
  std::vector<uint8_t> ops;
  uint64_t counters[16] = {0};
  uint64_t n_items = 0;
};

struct orc_swg {
  SwgExtend s;
  orc_swg(usize bw, Scoring sc) : s(bw, sc) {}
};

static void append_aln(orc_result* r, const GenomeAlignment& g) {
  orc_aln a;
  memset(&a, 0, sizeof(a));
  a.ystart = g.gx_aln.ystart;
  a.yend = g.gx_aln.yend;
  a.ylen = g.gx_aln.ylen;
  a.score = g.gx_aln.score;
  a.ref_id = (uint32_t)g.ref_id;
  a.xstart = (uint32_t)g.gx_aln.xstart;
  a.xend = (uint32_t)g.gx_aln.xend;
  a.xlen = (uint32_t)g.gx_aln.xlen;
  a.strand = g.strand;
  a.primary = g.primary;
  a.aln_type = g.aln_type;
  a.ops_off = r->ops.size();
  serialize_ops(g.gx_aln.operations, r->ops);
  a.ops_len = (uint32_t)(r->ops.size() - a.ops_off);
  a.tx_or_gene_idx = 0xFFFFFFFFu;
  if (g.aln_type == EXONIC) {
    a.tx_or_gene_idx = (uint32_t)g.tx_idx;
    a.tx_score = g.tx_aln.score;
    a.tx_ystart = g.tx_aln.ystart;
    a.tx_yend = g.tx_aln.yend;
    a.tx_ylen = g.tx_aln.ylen;
    a.tx_xstart = (uint32_t)g.tx_aln.xstart;
    a.tx_xend = (uint32_t)g.tx_aln.xend;
    a.tx_ops_off = r->ops.size();
    serialize_ops(g.tx_aln.operations, r->ops);
    a.tx_ops_len = (uint32_t)(r->ops.size() - a.tx_ops_off);
  } else if (g.aln_type == INTRONIC) {
    a.tx_or_gene_idx = (uint32_t)g.gene_idx;
  }
  r->alns.push_back(a);
}

extern "C" {

orc_swg* orc_swg_new(uint64_t max_bw, int32_t go, int32_t ge, int32_t m, int32_t mm) {
  return new orc_swg(max_bw, Scoring{go, ge, m, mm});
}
void orc_swg_free(orc_swg* s) { delete s; }
uint64_t orc_swg_phase1_breaks(const orc_swg* s) { return s->s.n_phase1_breaks; }
uint64_t orc_swg_cells(const orc_swg* s) { return s->s.n_cells; }

static int32_t put_ops(const std::vector<Op>& ops, uint8_t* buf, uint64_t cap, uint64_t* len) {
  std::vector<uint8_t> b;
  serialize_ops(ops, b);
  *len = b.size();
  if (b.size() > cap) return -1;
  if (!b.empty()) memcpy(buf, b.data(), b.size());
  return 0;
}

int32_t orc_swg_extend(orc_swg* s, const uint8_t* x, uint64_t xlen, const uint8_t* y, uint64_t ylen, uint64_t bw,
                       int32_t xd, int32_t* score, uint64_t* xend, uint64_t* yend, uint8_t* ops_buf, uint64_t ops_cap,
                       uint64_t* ops_len) {
  if (bw > s->s.max_band_width) return -5;  // assert!, src/swg.rs:32
  s->s.fault = false;
  Alignment a = s->s.extend(x, xlen, y, ylen, bw, xd);
  if (s->s.fault) return -5;
  *score = a.score;
  *xend = a.xend;
  *yend = a.yend;
  return put_ops(a.operations, ops_buf, ops_cap, ops_len);
}

orc_result* orc_swg_extend_batch(const uint8_t* xb, const uint64_t* xo, const uint8_t* yb, const uint64_t* yo,
                                 const uint32_t* bw, const int32_t* xd, uint32_t max_bw, uint64_t n) {
  orc_result* r = new orc_result();
  r->n = n;
  r->n_items = n;
  for (uint64_t i = 0; i < n; i++) {
    SwgExtend s(max_bw, Scoring{-1, -1, 1, -1});
    Alignment a = s.extend(xb + xo[i], xo[i + 1] - xo[i], yb + yo[i], yo[i + 1] - yo[i], bw[i], xd[i]);
    orc_swg_aln o;
    o.ops_off = r->ops.size();
    serialize_ops(a.operations, r->ops);
    o.ops_len = (uint32_t)(r->ops.size() - o.ops_off);
    o.score = a.score;
    o.xend = (uint32_t)a.xend;
    o.yend = (uint32_t)a.yend;
    r->swg.push_back(o);
    r->counters[9] += 1;
    r->counters[10] += s.n_cells;
    r->counters[11] += s.n_cols;
  }
  return r;
}

int32_t orc_extend_left_right(orc_swg* s, const uint8_t* ref_seq, uint64_t ref_len, uint64_t hr, uint64_t hq,
                              uint64_t hl, const uint8_t* read, uint64_t read_len, uint64_t bw, int32_t xd,
                              int32_t* score, uint64_t* ystart, uint64_t* xstart, uint64_t* yend, uint64_t* xend,
                              uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_len) {
  if (bw > s->s.max_band_width) return -5;
  s->s.fault = false;
  Alignment a = extend_left_right(ref_seq, ref_len, Mem{hr, hq, hl}, read, read_len, s->s, bw, xd);
  if (s->s.fault) return -5;
  *score = a.score;
  *ystart = a.ystart;
  *xstart = a.xstart;
  *yend = a.yend;
  *xend = a.xend;
  return put_ops(a.operations, ops_buf, ops_cap, ops_len);
}

void orc_extend_seed_match(const uint8_t* ref_seq, uint64_t ref_len, orc_mem* hit, const uint8_t* read,
                           uint64_t read_len) {
  Mem m{hit->ref_idx, hit->query_idx, hit->len};
  extend_seed_match(ref_seq, ref_len, m, read, read_len);
  hit->ref_idx = m.ref_idx;
  hit->query_idx = (uint32_t)m.query_idx;
  hit->len = (uint32_t)m.len;
}

int32_t orc_intersect(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) { return intersect(a0, a1, b0, b1) ? 1 : 0; }

static std::vector<Exon> to_exons(const orc_exon* e, uint64_t n) {
  std::vector<Exon> v;
  for (uint64_t i = 0; i < n; i++) v.push_back(Exon{e[i].start, e[i].end, e[i].tx_idx});
  return v;
}

int32_t orc_lift_mem_to_tx(const orc_mem* mem, const orc_exon* exons, uint64_t n_exons, orc_mem* out) {
  Mem o;
  if (!lift_mem_to_tx(Mem{mem->ref_idx, mem->query_idx, mem->len}, to_exons(exons, n_exons), &o)) return -5;
  out->ref_idx = o.ref_idx;
  out->query_idx = (uint32_t)o.query_idx;
  out->len = (uint32_t)o.len;
  return 0;
}

int32_t orc_lift_tx_to_gx(const uint8_t* ops, uint64_t ops_len, uint64_t ystart, uint64_t yend, const orc_exon* exons,
                          uint64_t n_exons, uint64_t* out_ystart, uint64_t* out_yend, uint8_t* out_ops,
                          uint64_t out_cap, uint64_t* out_len) {
  Alignment a;
  a.ystart = ystart;
  a.yend = yend;
  if (!deserialize_ops(ops, ops_len, a.operations)) return -1;
  Alignment o;
  if (!lift_tx_to_gx(a, to_exons(exons, n_exons), &o)) return -5;
  *out_ystart = o.ystart;
  *out_yend = o.yend;
  return put_ops(o.operations, out_ops, out_cap, out_len);
}

uint64_t orc_filter_overlapping(const uint32_t* name_rank, const uint8_t* strand, const uint64_t* ystart,
                                const uint64_t* yend, const int32_t* score, uint64_t n, uint64_t* kept_idx) {
  std::vector<GenomeAlignment> v(n);
  std::vector<uint32_t> rank(n);
  for (uint64_t i = 0; i < n; i++) {
    v[i].gx_aln.ystart = ystart[i];
    v[i].gx_aln.yend = yend[i];
    v[i].gx_aln.score = score[i];
    v[i].gx_aln.xlen = i;  // carries the input index through the filter
    v[i].strand = strand[i] != 0;
    v[i].ref_id = i;
    rank[i] = name_rank[i];
    v[i].aln_type = INTERGENIC;
    v[i].primary = false;
  }
  std::vector<GenomeAlignment> res = filter_overlapping(nullptr, std::move(v), rank.data());
  for (uint64_t i = 0; i < res.size(); i++) kept_idx[i] = res[i].gx_aln.xlen;
  return res.size();
}

void orc_suffix_array_naive(const uint8_t* text, uint64_t n, uint32_t* sa) {
  for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
  std::sort(sa, sa + n, [&](uint32_t a, uint32_t b) {
    uint64_t la = n - a, lb = n - b;
    int c = memcmp(text + a, text + b, std::min(la, lb));
    if (c != 0) return c < 0;
    return la < lb;
  });
}

int32_t orc_suffix_array_verify(const uint8_t* text, uint64_t n, const uint32_t* sa) {
  if (n == 0) return 1;
  std::vector<uint32_t> rank(n + 1, 0);
  std::vector<char> seen(n, 0);
  for (uint64_t r = 0; r < n; r++) {
    if (sa[r] >= n || seen[sa[r]]) return 0;
    seen[sa[r]] = 1;
    rank[sa[r]] = (uint32_t)r + 1;  // rank[n] = 0: the empty suffix sorts first
  }
  for (uint64_t r = 0; r + 1 < n; r++) {
    uint32_t a = sa[r], b = sa[r + 1];
    if (text[a] > text[b]) return 0;
    if (text[a] == text[b] && !(rank[a + 1] < rank[b + 1])) return 0;
  }
  return 1;
}

int32_t orc_suffix_array_verify64(const uint8_t* text, uint64_t n, const uint64_t* sa) {
  if (n == 0) return 1;
  std::vector<uint64_t> rank(n + 1, 0);
  std::vector<char> seen(n, 0);
  for (uint64_t r = 0; r < n; r++) {
    if (sa[r] >= n || seen[sa[r]]) return 0;
    seen[sa[r]] = 1;
    rank[sa[r]] = r + 1;  // rank[n] = 0: the empty suffix sorts first
  }
  for (uint64_t r = 0; r + 1 < n; r++) {
    uint64_t a = sa[r], b = sa[r + 1];
    if (text[a] > text[b]) return 0;
    if (text[a] == text[b] && !(rank[a + 1] < rank[b + 1])) return 0;
  }
  return 1;
}

// sa32 or sa64 (exactly one): the suffix array of `text`.  flags bit 0: take it unverified (texts of billions of symbols: the
// rank array of the check alone is 8 n bytes; the library's builder has its own check); bit 1: do not keep the plain suffix
// array (the matching-statistics cross-check, orc_smems_ms, is not available then).
static orc_index* index_create(const uint8_t* text, uint64_t n, const orc_ref* refs, uint32_t n_refs, const orc_tx* txs,
                               uint32_t n_txs, const orc_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                               uint64_t n_tx_seq, const orc_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                               uint32_t n_names, const uint32_t* sa32, const uint64_t* sa64, uint32_t sa_rate, uint32_t occ_rate,
                               uint32_t flags) {
  if (!sa64 && n >= 0xFFFFFFFFull) return nullptr;
  orc_index* ix = new orc_index();
  ix->text.assign(text, text + n);
  for (uint32_t i = 0; i < n_refs; i++)
    ix->refs.push_back(Ref{refs[i].name_id, refs[i].strand != 0, refs[i].len, refs[i].start_idx, refs[i].end_idx});
  ix->name_rank.assign(name_rank, name_rank + n_names);
  ix->tx_seq.assign(tx_seq, tx_seq + n_tx_seq);
  std::vector<uint32_t> own_sa;
  if (!sa32 && !sa64) {
    own_sa.resize(n);
    orc_suffix_array_naive(text, n, own_sa.data());
    sa32 = own_sa.data();
  } else if (!(flags & 1u) && !(sa64 ? orc_suffix_array_verify64(text, n, sa64) : orc_suffix_array_verify(text, n, sa32))) {
    delete ix;
    return nullptr;
  }
  const bool keep_full = !(flags & 2u);
  if (sa64)
    ix->fmd.build(text, n, sa64, sa_rate ? sa_rate : 32, occ_rate ? occ_rate : 128, keep_full);
  else
    ix->fmd.build(text, n, sa32, sa_rate ? sa_rate : 32, occ_rate ? occ_rate : 128, keep_full);
  // transcripts + exon tree: src/index.rs:137-206.  Exons are inserted in the
  // order the `transcriptome` crate lists them (genomic order), i.e. before the
  // reverse() applied to '-' strand transcripts at :192-195.
  for (uint32_t t = 0; t < n_txs; t++) {
    Tx tx;
    tx.strand = txs[t].strand != 0;
    tx.seq = ix->tx_seq.data() + txs[t].seq_off;
    tx.seq_len = txs[t].seq_len;
    tx.gene_idx = txs[t].gene_idx;
    for (uint32_t e = 0; e < txs[t].n_exons; e++) {
      const orc_exon& x = exons[txs[t].exon_begin + e];
      tx.exons.push_back(Exon{x.start, x.end, x.tx_idx});
    }
    if (tx.strand) {
      for (auto& e : tx.exons) ix->exon_to_tx.insert(e.start, e.end, e.tx_idx);
    } else {
      for (usize e = tx.exons.size(); e-- > 0;) ix->exon_to_tx.insert(tx.exons[e].start, tx.exons[e].end, tx.exons[e].tx_idx);
    }
    ix->txs.push_back(std::move(tx));
  }
  (void)n_exons;
  // gene tree: src/index.rs:208-213
  for (uint32_t g = 0; g < n_genes; g++) ix->gene_intervals.insert(genes[g].start, genes[g].end, g);
  ix->n_genes = n_genes;
  return ix;
}
orc_index* orc_index_create(const uint8_t* text, uint64_t n, const orc_ref* refs, uint32_t n_refs, const orc_tx* txs,
                            uint32_t n_txs, const orc_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                            uint64_t n_tx_seq, const orc_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                            uint32_t n_names, const uint32_t* sa, uint32_t sa_rate, uint32_t occ_rate) {
  return index_create(text, n, refs, n_refs, txs, n_txs, exons, n_exons, tx_seq, n_tx_seq, genes, n_genes, name_rank, n_names, sa, nullptr,
                      sa_rate, occ_rate, 0);
}
// the same with a 64-bit suffix array: any text length, like the reference's Vec<usize> (src/index.rs:103-111, :364-388)
orc_index* orc_index_create64(const uint8_t* text, uint64_t n, const orc_ref* refs, uint32_t n_refs, const orc_tx* txs,
                              uint32_t n_txs, const orc_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                              uint64_t n_tx_seq, const orc_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                              uint32_t n_names, const uint64_t* sa, uint32_t sa_rate, uint32_t occ_rate, uint32_t flags) {
  if (!sa) return nullptr;
  return index_create(text, n, refs, n_refs, txs, n_txs, exons, n_exons, tx_seq, n_tx_seq, genes, n_genes, name_rank, n_names, nullptr, sa,
                      sa_rate, occ_rate, flags);
}
void orc_index_free(orc_index* ix) { delete ix; }

static orc_result* smems_batch(const orc_index* ix, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t k,
                               bool ms) {
  orc_result* r = new orc_result();
  r->n = n;
  r->offsets.push_back(0);
  for (uint64_t i = 0; i < n; i++) {
    std::vector<uint8_t> read(bases + off[i], bases + off[i + 1]);
    for (auto& c : read)
      if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
    usize ns = 0;
    std::vector<Mem> m = ms ? ix->all_smems_ms(read.data(), read.size(), k, &ns) : ix->all_smems(read.data(), read.size(), k, &ns);
    for (auto& x : m) r->mems.push_back(orc_mem{x.ref_idx, (uint32_t)x.query_idx, (uint32_t)x.len});
    r->offsets.push_back(r->mems.size());
    r->counters[7] += ns;
    r->counters[8] += m.size();
  }
  r->n_items = r->mems.size();
  return r;
}
orc_result* orc_all_smems_batch(const orc_index* ix, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t k) {
  return smems_batch(ix, bases, off, n, k, false);
}
orc_result* orc_all_smems_batch_ms(const orc_index* ix, const uint8_t* bases, const uint64_t* off, uint64_t n,
                                   uint64_t k) {
  return smems_batch(ix, bases, off, n, k, true);
}

uint64_t orc_exon_tree_find(const orc_index* ix, uint64_t s, uint64_t e, uint32_t* out, uint64_t cap) {
  uint64_t c = 0;
  ix->exon_to_tx.find(s, e, [&](usize v) {
    if (c < cap) out[c] = (uint32_t)v;
    c++;
  });
  return c;
}
uint64_t orc_gene_tree_find(const orc_index* ix, uint64_t s, uint64_t e, uint32_t* out, uint64_t cap) {
  uint64_t c = 0;
  ix->gene_intervals.find(s, e, [&](usize v) {
    if (c < cap) out[c] = (uint32_t)v;
    c++;
  });
  return c;
}

orc_result* orc_align_batch(const orc_index* ix, const orc_opts* opts, const uint8_t* bases, const uint64_t* off,
                            uint64_t n, uint32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n && n > 0) n_threads = (uint32_t)n;
  std::vector<orc_result*> parts(n_threads, nullptr);
  std::vector<char> faults(n_threads, 0);
  auto work = [&](uint32_t t) {
    uint64_t b = n * t / n_threads, e = n * (t + 1) / n_threads;
    orc_result* r = new orc_result();
    Counters cnt;
    bool fault = false;
    for (uint64_t i = b; i < e; i++) {
      std::vector<GenomeAlignment> alns = align_read(*ix, bases + off[i], off[i + 1] - off[i], *opts, cnt, &fault);
      cnt.c[0]++;
      if (alns.empty())
        cnt.c[2]++;
      else
        cnt.c[1]++;
      for (auto& g : alns) {
        cnt.c[3]++;
        cnt.c[4 + (int)g.aln_type]++;
        append_aln(r, g);
      }
      r->offsets.push_back(r->alns.size());
    }
    cnt.c[12] = r->ops.size();
    memcpy(r->counters, cnt.c, sizeof(cnt.c));
    faults[t] = fault;
    parts[t] = r;
  };
  if (n_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  orc_result* r = new orc_result();
  r->n = n;
  r->offsets.push_back(0);
  for (uint32_t t = 0; t < n_threads; t++) {
    orc_result* p = parts[t];
    uint64_t aln_base = r->alns.size(), op_base = r->ops.size();
    for (auto a : p->alns) {
      a.ops_off += op_base;
      if (a.aln_type == EXONIC) a.tx_ops_off += op_base;
      r->alns.push_back(a);
    }
    r->ops.insert(r->ops.end(), p->ops.begin(), p->ops.end());
    for (uint64_t o : p->offsets) r->offsets.push_back(aln_base + o);
    for (int c = 0; c < 16; c++) r->counters[c] += p->counters[c];
    if (faults[t]) r->counters[15] += 1;  // reads where the reference would have panicked
    delete p;
  }
  r->n_items = r->alns.size();
  return r;
}

void orc_result_free(orc_result* r) { delete r; }
uint64_t orc_result_n(const orc_result* r) { return r->n; }
uint64_t orc_result_n_items(const orc_result* r) { return r->n_items; }
uint64_t orc_result_n_op_bytes(const orc_result* r) { return r->ops.size(); }
const uint64_t* orc_result_offsets(const orc_result* r) { return r->offsets.data(); }
const orc_aln* orc_result_alns(const orc_result* r) { return r->alns.data(); }
const orc_mem* orc_result_mems(const orc_result* r) { return r->mems.data(); }
const orc_swg_aln* orc_result_swg(const orc_result* r) { return r->swg.data(); }
const uint8_t* orc_result_ops(const orc_result* r) { return r->ops.data(); }
const uint64_t* orc_result_counters(const orc_result* r) { return r->counters; }

}  // extern "C"
