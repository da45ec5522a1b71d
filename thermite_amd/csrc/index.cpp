// index.cpp -- host side of thm_index: validates the caller's tables and builds
// the search structures the kernels use.
//
//   reference                         here
//   ---------                         ----
//   FMD index over BWT/Occ/Less,      plain text + full suffix array + a table
//   sampled SA (src/index.rs:103-111)   of suffix-array intervals of all ACGT
//                                       kt-mers (any exact-match index yields
//                                       the same SMEM set)
//   bio IntervalTree x2               the same AVL trees (same insertion order,
//   (src/index.rs:135,182,208-213)      same rebalancing), flattened to arrays
//                                       so the device can replay find()'s visit
//                                       order
//
// Pure host C++ (no HIP); device upload lives in aligner.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <type_traits>

#include "thermite_internal.h"

namespace thm {

// last error of the calling thread (thm_last_error(NULL)): per thread, so that the parser, GPU and writer threads of
// the file driver never read a string another thread is assigning
static thread_local std::string g_err;
void set_global_error(const std::string& msg) { g_err = msg; }
const char* global_error_cstr() { return g_err.c_str(); }

template <class C>
bool verify_suffix_array(const uint8_t* text, uint64_t n, const C* sa) {
  if (n == 0) return true;
  std::vector<C> rank(n + 1, 0);  // rank[n] = 0: the empty suffix sorts first
  std::vector<uint8_t> seen(n, 0);
  for (uint64_t r = 0; r < n; r++) {
    if (sa[r] >= n || seen[sa[r]]) return false;
    seen[sa[r]] = 1;
    rank[sa[r]] = (C)r + 1;
  }
  for (uint64_t r = 0; r + 1 < n; r++) {
    C a = sa[r], b = sa[r + 1];
    if (text[a] > text[b]) return false;
    if (text[a] == text[b] && !(rank[a + 1] < rank[b + 1])) return false;
  }
  return true;
}
template bool verify_suffix_array<uint32_t>(const uint8_t*, uint64_t, const uint32_t*);
template bool verify_suffix_array<uint64_t>(const uint8_t*, uint64_t, const uint64_t*);

namespace {

// AVL interval tree with bio's rules: key = interval start, ties go left
// (`start <= node.start`), rebalance when heights differ by more than one,
// double rotation when the inner grandchild is the taller one.
struct Avl {
  std::vector<TreeNode> nd;
  std::vector<int32_t> ht;
  int32_t h(int32_t x) const { return x < 0 ? 0 : ht[x]; }
  void upd(int32_t x) {
    TreeNode& t = nd[x];
    ht[x] = 1 + std::max(h(t.left), h(t.right));
    t.max = t.end;
    if (t.left >= 0 && t.max < nd[t.left].max) t.max = nd[t.left].max;
    if (t.right >= 0 && t.max < nd[t.right].max) t.max = nd[t.right].max;
  }
  int32_t rot_right(int32_t x) {
    int32_t y = nd[x].left;
    nd[x].left = nd[y].right;
    nd[y].right = x;
    upd(x);
    upd(y);
    return y;
  }
  int32_t rot_left(int32_t x) {
    int32_t y = nd[x].right;
    nd[x].right = nd[y].left;
    nd[y].left = x;
    upd(x);
    upd(y);
    return y;
  }
  int32_t insert(int32_t x, uint64_t s, uint64_t e, uint32_t v) {
    if (x < 0) {
      TreeNode t;
      t.start = s;
      t.end = e;
      t.max = e;
      t.value = v;
      t.left = t.right = -1;
      t.pad_ = 0;
      nd.push_back(t);
      ht.push_back(1);
      return (int32_t)nd.size() - 1;
    }
    if (s <= nd[x].start) {
      int32_t c = insert(nd[x].left, s, e, v);
      nd[x].left = c;
    } else {
      int32_t c = insert(nd[x].right, s, e, v);
      nd[x].right = c;
    }
    int32_t lh = h(nd[x].left), rh = h(nd[x].right);
    if (std::abs(lh - rh) <= 1) {
      upd(x);
      return x;
    }
    if (rh > lh) {
      int32_t r = nd[x].right;
      if (h(nd[r].left) > h(nd[r].right)) nd[x].right = rot_right(r);
      return rot_left(x);
    }
    int32_t l = nd[x].left;
    if (h(nd[l].right) > h(nd[l].left)) nd[x].left = rot_left(l);
    return rot_right(x);
  }
};

// ranks in the order IntervalTree::find visits nodes: node, right subtree, left subtree
void preorder_ranks(const std::vector<TreeNode>& nd, int32_t root, std::vector<uint32_t>& rank) {
  rank.assign(nd.size(), 0);
  std::vector<int32_t> st;
  if (root >= 0) st.push_back(root);
  uint32_t r = 0;
  while (!st.empty()) {
    const int32_t x = st.back();
    st.pop_back();
    rank[x] = r++;
    if (nd[x].left >= 0) st.push_back(nd[x].left);    // popped after the right child
    if (nd[x].right >= 0) st.push_back(nd[x].right);
  }
}

template <class C>
void build_grid(const std::vector<TreeNode>& nd, int32_t root, uint64_t n, std::vector<uint32_t>& off,
                std::vector<GridEntryT<C>>& entries) {
  typedef GridEntryT<C> GridEntry;
  std::vector<uint32_t> rank;
  preorder_ranks(nd, root, rank);
  const uint64_t nbins = (n >> GRID_SHIFT) + 1;
  off.assign(nbins + 1, 0);
  for (const TreeNode& t : nd) {
    if (t.end <= t.start) continue;  // an empty interval overlaps nothing
    for (uint64_t b = t.start >> GRID_SHIFT; b <= ((t.end - 1) >> GRID_SHIFT) && b < nbins; b++) off[b + 1]++;
  }
  for (uint64_t b = 0; b < nbins; b++) off[b + 1] += off[b];
  entries.assign(off[nbins], GridEntry{0, 0, 0, 0});
  std::vector<uint32_t> fill(off.begin(), off.end() - 1);
  for (size_t x = 0; x < nd.size(); x++) {
    const TreeNode& t = nd[x];
    if (t.end <= t.start) continue;
    for (uint64_t b = t.start >> GRID_SHIFT; b <= ((t.end - 1) >> GRID_SHIFT) && b < nbins; b++)
      entries[fill[b]++] = GridEntry{(C)t.start, (C)t.end, t.value, (rank[x] << 8) | (uint32_t)(b & 0xff)};
  }
  for (uint64_t b = 0; b < nbins; b++)
    std::sort(entries.begin() + off[b], entries.begin() + off[b + 1],
              [](const GridEntry& a, const GridEntry& c) { return a.rank < c.rank; });
}

// exon grid: the same bins and order as build_grid, entries enriched with their transcript and exon (thermite_internal.h)
template <class C>
void build_exon_grid(const thm_index* ix, std::vector<uint32_t>& off, std::vector<ExonEntryT<C>>& entries) {
  std::vector<GridEntryT<C>> plain;
  build_grid<C>(ix->exon_tree, ix->exon_root, ix->n, off, plain);
  // the plain builder keeps (start, end, tx, rank); the exon is found again through the tree node it came from:
  // (tx, start, end) identifies an exon of a transcript unless the transcript lists the same interval twice, in
  // which case either copy describes the same bases except for its transcript offset -- so map through nodes
  std::vector<uint32_t> rank;
  preorder_ranks(ix->exon_tree, ix->exon_root, rank);
  std::vector<uint32_t> exon_of_rank(rank.size(), 0);
  for (size_t x = 0; x < rank.size(); x++) exon_of_rank[rank[x]] = ix->exon_node_exon[x];
  // ascending, disjoint exons in transcript order (src/index.rs:149-195) are what the fast path assumes
  std::vector<uint8_t> tx_ascending(ix->txs.size(), 1);
  for (size_t t = 0; t < ix->txs.size(); t++) {
    const thm_tx& tx = ix->txs[t];
    for (uint32_t q = 1; q < tx.n_exons; q++)
      if (ix->exons[tx.exon_begin + q - 1].end > ix->exons[tx.exon_begin + q].start) tx_ascending[t] = 0;
  }
  entries.resize(plain.size());
  for (size_t k = 0; k < plain.size(); k++) {
    const GridEntryT<C>& g = plain[k];
    const uint32_t gi = exon_of_rank[g.rank >> 8];
    const thm_exon& ex = ix->exons[gi];
    const thm_tx& tx = ix->txs[ex.tx_idx];
    const uint32_t ei = (uint32_t)(gi - tx.exon_begin);
    ExonEntryT<C> e;
    e.start = g.start;
    e.end = g.end;
    e.value = g.value;
    e.rank = g.rank;
    const bool ascending = tx_ascending[ex.tx_idx] != 0;
    e.prev_end = !ascending ? (C)~(C)0 : (ei == 0 ? (C)0 : (C)ix->exons[gi - 1].end);
    e.txoff = (uint32_t)ix->exon_txoff[gi];
    e.exon_idx = ei;
    e.seq_off = tx.seq_off;
    e.seq_len = (uint32_t)tx.seq_len;
    e.n_exons = tx.n_exons;
    entries[k] = e;
  }
}

inline int base_code(uint8_t c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

// k-mer prefix table: for every ACGT-only kt-mer c the suffix-array interval [lo, hi) of the suffixes
// that start with it.  Built by counting, without touching the suffix array (one sequential pass over
// the text instead of n random accesses):
//   hi(c) - lo(c) = number of text positions where c occurs;
//   lo(c) = number of suffixes that sort before the string c
//         = (occurrences of codes < c) + (suffixes whose first kt symbols are not all ACGT and that
//           sort before c).
// A suffix of the second kind reads u x ..., u = m < kt ACGT symbols, x outside ACGT ('$', 'N', or the
// end of the text): in byte order it sorts before exactly the codes c >= T, where T = u followed by the
// smallest ACGT letter above x and then A's (x above 'T': the first code after the block that starts
// with u).  So one difference array over the codes, filled in the same pass, gives the second term.
template <class C>
void build_lut(thm_index* ix, std::vector<LutEntryT<C>>& lut) {
  const uint64_t n = ix->n;
  uint32_t kt = 1;
  // about one suffix per occupied bucket: 4^kt <= 4n, at most 14 (2 GiB of 8-byte entries)
  while (kt < 14 && (1ull << (2 * (kt + 1))) <= 4 * n) kt++;
  if (const char* e = getenv("THM_KT")) {  // tuning knob
    const int v = atoi(e);
    if (v >= 1 && v <= 15) kt = (uint32_t)v;
  }
  ix->kt = kt;
  const uint64_t nk = 1ull << (2 * kt);
  // counted in place: lut[c].hi = occurrences of c, lut[c].lo = suffixes of the second kind with threshold c
  // (a threshold past the last code precedes nothing and is dropped)
  lut.assign(nk, LutEntryT<C>{0, 0});
  const uint8_t* text = ix->text.data();
  // Counting, over the text in parallel: thread t owns the positions [s, e) of its range -- it counts the kt-mers that
  // END there, and handles every run end and every non-ACGT position inside.  The text is cut into maximal ACGT runs
  // [a, b) (b = first position outside ACGT, or n); a range that starts inside a run recovers what it needs of the run
  // before it by looking back kt - 1 symbols.  Increments are relaxed atomics on the table (the counts do not
  // depend on the order).
  auto add_hi = [&](uint64_t c) { __atomic_fetch_add(&lut[c].hi, (C)1, __ATOMIC_RELAXED); };
  auto add_lo = [&](uint64_t T) {
    if (T < nk) __atomic_fetch_add(&lut[T].lo, (C)1, __ATOMIC_RELAXED);
  };
  auto count_range = [&](uint64_t s, uint64_t e) {
    const uint64_t mask = nk - 1;
    uint64_t p = s;
    // the part of a run that lies before s, as far as it matters: `have` symbols (at most kt - 1) and their code
    uint64_t have = 0, code = 0;
    if (s < n && base_code(text[s]) >= 0) {
      while (have < kt - 1 && have < s && base_code(text[s - 1 - have]) >= 0) have++;
      for (uint64_t q = s - have; q < s; q++) code = ((code << 2) | (uint64_t)base_code(text[q])) & mask;
    }
    // the run that ends at bnd (its first symbol outside ACGT, or the end of the text): its last min(kt - 1, length)
    // positions start a suffix with m < kt ACGT symbols.  Whoever owns bnd handles them, wherever they lie.
    auto run_end = [&](uint64_t bnd) {
      const uint8_t x = bnd >= n ? 0 : text[bnd];  // end of text: smaller than every symbol
      uint64_t back = 0;
      while (back < kt - 1 && back < bnd && base_code(text[bnd - 1 - back]) >= 0) back++;
      for (uint64_t d0 = bnd - back; d0 < bnd; d0++) {
        const uint32_t m = (uint32_t)(bnd - d0);
        uint64_t u = 0;
        for (uint64_t r = d0; r < bnd; r++) u = (u << 2) | (uint64_t)base_code(text[r]);
        const uint32_t rest = kt - m;  // symbols of a code after u; >= 1
        uint64_t T;
        if (x < 'A')
          T = u << (2 * rest);
        else if (x < 'C')
          T = ((u << 2) | 1) << (2 * (rest - 1));
        else if (x < 'G')
          T = ((u << 2) | 2) << (2 * (rest - 1));
        else if (x < 'T')
          T = ((u << 2) | 3) << (2 * (rest - 1));
        else
          T = (u + 1) << (2 * rest);
        add_lo(T);
      }
    };
    while (p < e) {
      if (base_code(text[p]) < 0) {  // a suffix that starts outside ACGT: m = 0, u empty, x = text[p]
        run_end(p);  // (nothing to do when text[p - 1] is outside ACGT too)
        const uint8_t x = text[p];
        const uint64_t T = x < 'A' ? 0 : x < 'C' ? (1ull << (2 * (kt - 1))) : x < 'G' ? (2ull << (2 * (kt - 1)))
                           : x < 'T' ? (3ull << (2 * (kt - 1))) : nk;
        add_lo(T);
        p++;
        have = 0;
        code = 0;
        continue;
      }
      // inside a run: roll on to its end, or to the end of the range
      uint64_t q = p;
      while (q < e && base_code(text[q]) >= 0) {
        code = ((code << 2) | (uint64_t)base_code(text[q])) & mask;
        if (have + 1 >= kt) add_hi(code);  // the kt-mer ending at q
        else have++;
        q++;
      }
      if (q == n) run_end(n);  // the run ends with the text
      p = q;
    }
  };
  {
    unsigned T = (unsigned)std::min<uint64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n / (4u << 20) + 1);
    if (const char* e = getenv("THM_INDEX_THREADS")) T = (unsigned)std::max(1, std::min(64, atoi(e)));
    if (T <= 1) {
      count_range(0, n);
    } else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; t++) th.emplace_back(count_range, n * t / T, n * (t + 1) / T);
      for (auto& x : th) x.join();
    }
  }
  uint64_t before = 0;  // suffixes that sort before code c
  for (uint64_t c = 0; c < nk; c++) {
    before += lut[c].lo;
    const uint64_t occ = lut[c].hi;
    if (occ) {
      lut[c].lo = (C)before;
      lut[c].hi = (C)(before + occ);
    } else {
      lut[c].lo = lut[c].hi = 0;
    }
    before += occ;
  }
}

// the same table read off the suffix array (n random accesses): the check of build_lut in the tests
template <class C>
bool check_lut(const thm_index* ix, const std::vector<LutEntryT<C>>& lut, const C* sa) {
  const uint64_t n = ix->n, kt = ix->kt, nk = 1ull << (2 * kt);
  std::vector<LutEntryT<C>> ref(nk, LutEntryT<C>{0, 0});
  for (uint64_t r = 0; r < n; r++) {
    const uint64_t p = sa[r];
    if (p + kt > n) continue;
    uint64_t code = 0;
    bool ok = true;
    for (uint64_t t = 0; t < kt && ok; t++) {
      const int c = base_code(ix->text[p + t]);
      ok = c >= 0;
      code = (code << 2) | (uint64_t)(c & 3);
    }
    if (!ok) continue;
    if (ref[code].hi == 0) ref[code].lo = (C)r;
    ref[code].hi = (C)r + 1;
  }
  for (uint64_t c = 0; c < nk; c++)
    if (ref[c].lo != lut[c].lo || ref[c].hi != lut[c].hi) return false;
  return true;
}

}  // namespace
}  // namespace thm

using namespace thm;

extern "C" {

int32_t thm_build_suffix_array(const uint8_t* text, uint64_t n, uint32_t* sa_out) {
  if ((!text && n) || !sa_out) return THM_ERR_INVALID_ARG;
  int rc = build_suffix_array(text, n, sa_out);
  return rc == 0 ? THM_OK : THM_ERR_UNSUPPORTED;
}
int32_t thm_build_suffix_array64(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
  if ((!text && n) || !sa_out) return THM_ERR_INVALID_ARG;
  try {
    int rc = build_suffix_array64(text, n, sa_out);
    return rc == 0 ? THM_OK : THM_ERR_UNSUPPORTED;
  } catch (const std::bad_alloc&) {
    set_global_error("out of memory building the suffix array");
    return THM_ERR_OOM;
  }
}

}  // extern "C"

static int32_t index_create_impl(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs, const thm_tx* txs,
                                 uint32_t n_txs, const thm_exon* exons, uint64_t n_exons, const uint8_t* tx_seq,
                                 uint64_t n_tx_seq, const thm_span* genes, uint32_t n_genes, const uint32_t* name_rank,
                                 uint32_t n_names, const void* sa, uint32_t sa_elem_bytes, uint32_t flags, thm_index** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!text || n == 0 || !refs || n_refs == 0) {
    set_global_error("thm_index_create_in_memory: empty text or refs");
    return THM_ERR_INVALID_ARG;
  }
  if (sa && sa_elem_bytes != 4 && sa_elem_bytes != 8) {
    set_global_error("suffix array entries must be 4 or 8 bytes");
    return THM_ERR_INVALID_ARG;
  }
  // coordinate width: 64-bit positions and ranks when the text does not fit 31 bits (with room for the window
  // arithmetic), when the caller asks for it, or when THM_FORCE_WIDE=1 (runs the wide code path on small texts)
  bool wide = n >= 0x7FFFFFF0ull || (flags & THM_INDEX_WIDE) != 0;
  if (const char* e = getenv("THM_FORCE_WIDE"))
    if (atoi(e) != 0) wide = true;
  if (n >= (1ull << 47)) {
    set_global_error("text longer than 2^47 symbols");
    return THM_ERR_UNSUPPORTED;
  }
  if ((n_txs && (!txs || !exons)) || (n_genes && !genes) || (n_tx_seq && !tx_seq)) return THM_ERR_INVALID_ARG;
  // refs must tile the text: [start,end) consecutive, end-1 is '$'
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n_refs; i++) {
    if (refs[i].start_idx != pos || refs[i].end_idx != pos + refs[i].len + 1 || refs[i].end_idx > n ||
        text[refs[i].end_idx - 1] != '$') {
      set_global_error("refs do not tile the text (src/index.rs:72-100 layout expected)");
      return THM_ERR_INVALID_ARG;
    }
    if (name_rank && refs[i].name_id >= n_names) return THM_ERR_INVALID_ARG;
    pos = refs[i].end_idx;
  }
  if (pos != n) {
    set_global_error("refs do not cover the text");
    return THM_ERR_INVALID_ARG;
  }
  uint32_t max_tx_exons = 0;
  for (uint32_t t = 0; t < n_txs; t++) {
    if (txs[t].exon_begin + txs[t].n_exons > n_exons || txs[t].seq_off + txs[t].seq_len > n_tx_seq ||
        txs[t].n_exons == 0 || (n_genes && txs[t].gene_idx >= n_genes)) {
      set_global_error("transcript table out of range");
      return THM_ERR_INVALID_ARG;
    }
    if (txs[t].seq_len >= 0x7FFF0000ull) {
      set_global_error("transcript longer than 2^31 bases");
      return THM_ERR_UNSUPPORTED;
    }
    max_tx_exons = std::max(max_tx_exons, txs[t].n_exons);
    uint64_t sum = 0;
    for (uint32_t e = 0; e < txs[t].n_exons; e++) {
      const thm_exon& x = exons[txs[t].exon_begin + e];
      if (x.start >= x.end || x.end > n || x.tx_idx != t) {
        set_global_error("exon table inconsistent");
        return THM_ERR_INVALID_ARG;
      }
      sum += x.end - x.start;
    }
    if (sum != txs[t].seq_len) {
      set_global_error("transcript sequence length != sum of exon lengths");
      return THM_ERR_INVALID_ARG;
    }
  }
  for (uint32_t g = 0; g < n_genes; g++)
    if (genes[g].start > genes[g].end) {  // Interval::new(start..end).unwrap() panics, src/index.rs:212
      set_global_error("gene with an empty transcript set (start > end)");
      return THM_ERR_OUT_OF_CONTRACT;
    }

  thm_index* ix = new thm_index();
  ix->n = n;
  ix->wide = wide;
  ix->max_tx_exons = max_tx_exons;
  ix->text.assign(text, text + n);
  ix->text.resize(n + 128, (uint8_t)'$');  // padding: batched 64-byte compares / 16-byte window loads may run past the end
  ix->refs.assign(refs, refs + n_refs);
  ix->name_rank.resize(n_refs);
  for (uint32_t i = 0; i < n_refs; i++) ix->name_rank[i] = name_rank ? name_rank[refs[i].name_id] : refs[i].name_id;
  ix->txs.assign(txs, txs + n_txs);
  ix->exons.assign(exons, exons + n_exons);
  ix->exon_txoff.assign(n_exons, 0);
  for (uint32_t t = 0; t < n_txs; t++) {
    uint64_t sum = 0;
    for (uint32_t e = 0; e < txs[t].n_exons; e++) {
      ix->exon_txoff[txs[t].exon_begin + e] = sum;
      sum += exons[txs[t].exon_begin + e].end - exons[txs[t].exon_begin + e].start;
    }
  }
  ix->tx_seq.assign(tx_seq, tx_seq + n_tx_seq);
  ix->tx_seq.resize(n_tx_seq + 16, (uint8_t)'$');
  ix->genes.assign(genes, genes + n_genes);

  // suffix array in the index's width (a supplied array is checked, and converted when its width differs)
  bool sa_on_gpu = false;
  auto fill_sa = [&](auto& dst) -> int {
    typedef typename std::remove_reference<decltype(dst)>::type::value_type C;
    dst.resize(n);
    if (sa) {
      if (sa_elem_bytes == sizeof(C)) {
        memcpy(dst.data(), sa, n * sizeof(C));
      } else if (sa_elem_bytes == 4) {
        const uint32_t* s4 = (const uint32_t*)sa;
        for (uint64_t i = 0; i < n; i++) dst[i] = (C)s4[i];
      } else {
        const uint64_t* s8 = (const uint64_t*)sa;
        for (uint64_t i = 0; i < n; i++) {
          if (s8[i] >= n) return THM_ERR_INVALID_ARG;
          dst[i] = (C)s8[i];
        }
      }
      if (!verify_suffix_array<C>(text, n, dst.data())) {
        set_global_error("supplied suffix array is not the suffix array of the text");
        return THM_ERR_INVALID_ARG;
      }
      return THM_OK;
    }
    // on the GPU where there is one (sa_gpu.hip: seconds against minutes); the host builder otherwise
    if (n >= (4u << 20) && !getenv("THM_SA_HOST")) {
      const int grc = build_suffix_array_gpu(text, n, dst.data(), (int)sizeof(C));
      if (grc == 0) {
        sa_on_gpu = true;
        return THM_OK;
      }
      // -1 no device, -2 entries too narrow, -3 not enough free device memory, -4 HIP error (an allocation, mostly)
      if (getenv("THM_INDEX_TIMING")) fprintf(stderr, "thm_index_create: device suffix sort declined (%d): sorting on the host\n", grc);
    }
    if (sizeof(C) == 4) return build_suffix_array(text, n, (uint32_t*)dst.data()) == 0 ? THM_OK : THM_ERR_UNSUPPORTED;
    return build_suffix_array64(text, n, (uint64_t*)dst.data()) == 0 ? THM_OK : THM_ERR_UNSUPPORTED;
  };
  // THM_INDEX_TIMING=1: seconds per phase of the build on stderr (tools/big_text.py --index-timing)
  const bool timing = getenv("THM_INDEX_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t0) {
    const auto t1 = now();
    if (timing) fprintf(stderr, "thm_index_create: %-28s %8.2f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  auto t_phase = now();
  const int src = wide ? fill_sa(ix->sa64) : fill_sa(ix->sa);
  if (src != THM_OK) {
    delete ix;
    return src;
  }
  lap(sa ? "suffix array (checked)" : sa_on_gpu ? "suffix array (GPU)" : "suffix array (SA-IS)", t_phase);
  if (wide)
    build_lut<uint64_t>(ix, ix->lut64);
  else
    build_lut<uint32_t>(ix, ix->lut);
  lap("k-mer table", t_phase);

  // exon_to_tx: src/index.rs:164-191 inserts each transcript's exons in the
  // order the annotation lists them (genomic order) BEFORE reversing the
  // '-' strand list at :192-195, i.e. stored order for '+', reverse for '-'.
  {
    Avl a;
    int32_t root = -1;
    for (uint32_t t = 0; t < n_txs; t++) {
      const thm_tx& tx = ix->txs[t];
      for (uint32_t k = 0; k < tx.n_exons; k++) {
        uint32_t e = tx.strand ? k : tx.n_exons - 1 - k;
        const thm_exon& x = ix->exons[tx.exon_begin + e];
        root = a.insert(root, x.start, x.end, x.tx_idx);
        ix->exon_node_exon.push_back((uint32_t)(tx.exon_begin + e));  // node indices are insertion order (rotations relink, never move)
      }
    }
    ix->exon_tree.swap(a.nd);
    ix->exon_root = root;
  }
  // gene_intervals: IntervalTree::from_iter in gene order, src/index.rs:208-213
  {
    Avl a;
    int32_t root = -1;
    for (uint32_t g = 0; g < n_genes; g++) root = a.insert(root, genes[g].start, genes[g].end, g);
    ix->gene_tree.swap(a.nd);
    ix->gene_root = root;
  }
  if (wide) {
    build_exon_grid<uint64_t>(ix, ix->exon_grid_off, ix->exon_grid64);
    build_grid<uint64_t>(ix->gene_tree, ix->gene_root, n, ix->gene_grid_off, ix->gene_grid64);
  } else {
    build_exon_grid<uint32_t>(ix, ix->exon_grid_off, ix->exon_grid);
    build_grid<uint32_t>(ix->gene_tree, ix->gene_root, n, ix->gene_grid_off, ix->gene_grid);
  }
  lap("interval trees and grids", t_phase);
  // idx_to_ref as two loads (thermite_internal.h, RefRecT)
  {
    const uint64_t nbins = (n >> GRID_SHIFT) + 1;
    ix->ref_bin.assign(nbins, 0);
    uint32_t r = 0;
    for (uint64_t b = 0; b < nbins; b++) {
      const uint64_t pos0 = b << GRID_SHIFT;
      while (r + 1 < n_refs && ix->refs[r].end_idx <= pos0) r++;
      ix->ref_bin[b] = r;
    }
    auto fill = [&](auto& v) {
      typedef typename std::remove_reference<decltype(v)>::type::value_type R;
      v.resize(n_refs);
      for (uint32_t i = 0; i < n_refs; i++) {
        R x;
        x.start = (decltype(x.start))ix->refs[i].start_idx;
        x.end = (decltype(x.end))ix->refs[i].end_idx;
        x.len = (decltype(x.len))ix->refs[i].len;
        x.name_rank = ix->name_rank[i];
        x.strand = ix->refs[i].strand ? 1u : 0u;
        v[i] = x;
      }
    };
    if (wide)
      fill(ix->ref_recs64);
    else
      fill(ix->ref_recs);
  }
  ix->dev_mu = new std::mutex();
  *out = ix;
  return THM_OK;
}

// no exception leaves the C ABI: allocation failures become THM_ERR_OOM
template <class F>
static int32_t guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    set_global_error("out of memory");
    return THM_ERR_OOM;
  } catch (const std::exception& e) {
    set_global_error(std::string("internal error: ") + e.what());
    return THM_ERR_INTERNAL;
  } catch (...) {
    set_global_error("internal error");
    return THM_ERR_INTERNAL;
  }
}

extern "C" {

int32_t thm_index_create_in_memory(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs,
                                   const thm_tx* txs, uint32_t n_txs, const thm_exon* exons, uint64_t n_exons,
                                   const uint8_t* tx_seq, uint64_t n_tx_seq, const thm_span* genes, uint32_t n_genes,
                                   const uint32_t* name_rank, uint32_t n_names, const uint32_t* sa, thm_index** out) {
  return guarded([&] {
    return index_create_impl(text, n, refs, n_refs, txs, n_txs, exons, n_exons, tx_seq, n_tx_seq, genes, n_genes, name_rank,
                             n_names, sa, 4, 0, out);
  });
}

int32_t thm_index_create_in_memory_ex(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs,
                                      const thm_tx* txs, uint32_t n_txs, const thm_exon* exons, uint64_t n_exons,
                                      const uint8_t* tx_seq, uint64_t n_tx_seq, const thm_span* genes, uint32_t n_genes,
                                      const uint32_t* name_rank, uint32_t n_names, const void* sa, uint32_t sa_elem_bytes,
                                      uint32_t flags, thm_index** out) {
  return guarded([&] {
    return index_create_impl(text, n, refs, n_refs, txs, n_txs, exons, n_exons, tx_seq, n_tx_seq, genes, n_genes, name_rank,
                             n_names, sa, sa_elem_bytes, flags, out);
  });
}

uint32_t thm_index_coord_bytes(const thm_index* ix) { return ix ? (ix->wide ? 8u : 4u) : 0u; }
const uint64_t* thm_index_suffix_array64(const thm_index* ix) { return (ix && ix->wide) ? ix->sa64.data() : nullptr; }

// test hook: the k-mer table (built by counting) against the one read off the suffix array
int32_t thm_debug_check_lut(const thm_index* ix) {
  if (!ix) return THM_ERR_INVALID_ARG;
  const bool ok = ix->wide ? check_lut<uint64_t>(ix, ix->lut64, ix->sa64.data()) : check_lut<uint32_t>(ix, ix->lut, ix->sa.data());
  return ok ? THM_OK : THM_ERR_INTERNAL;
}

uint64_t thm_index_text_len(const thm_index* ix) { return ix ? ix->n : 0; }
const uint32_t* thm_index_suffix_array(const thm_index* ix) { return (ix && !ix->wide) ? ix->sa.data() : nullptr; }

// Index::idx_to_ref, src/index.rs:287-290
int32_t thm_index_idx_to_ref(const thm_index* ix, uint64_t idx, uint64_t* offset) {
  if (!ix || idx >= ix->n) return THM_ERR_INVALID_ARG;
  size_t lo = 0, hi = ix->refs.size();
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (ix->refs[mid].end_idx <= idx)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (offset) *offset = idx - ix->refs[lo].start_idx;
  return (int32_t)lo;
}

}  // extern "C"
