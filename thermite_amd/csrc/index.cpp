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
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "thermite_internal.h"

namespace thm {

static std::mutex g_err_mu;
static std::string g_err;
void set_global_error(const std::string& msg) {
  std::lock_guard<std::mutex> g(g_err_mu);
  g_err = msg;
}
const char* global_error_cstr() {
  std::lock_guard<std::mutex> g(g_err_mu);
  return g_err.c_str();
}

bool verify_suffix_array(const uint8_t* text, uint64_t n, const uint32_t* sa) {
  if (n == 0) return true;
  std::vector<uint32_t> rank(n + 1, 0);  // rank[n] = 0: the empty suffix sorts first
  std::vector<uint8_t> seen(n, 0);
  for (uint64_t r = 0; r < n; r++) {
    if (sa[r] >= n || seen[sa[r]]) return false;
    seen[sa[r]] = 1;
    rank[sa[r]] = (uint32_t)r + 1;
  }
  for (uint64_t r = 0; r + 1 < n; r++) {
    uint32_t a = sa[r], b = sa[r + 1];
    if (text[a] > text[b]) return false;
    if (text[a] == text[b] && !(rank[a + 1] < rank[b + 1])) return false;
  }
  return true;
}

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

void build_grid(const std::vector<TreeNode>& nd, int32_t root, uint64_t n, std::vector<uint32_t>& off,
                std::vector<GridEntry>& entries) {
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
      entries[fill[b]++] = GridEntry{(uint32_t)t.start, (uint32_t)t.end, t.value, (rank[x] << 8) | (uint32_t)(b & 0xff)};
  }
  for (uint64_t b = 0; b < nbins; b++)
    std::sort(entries.begin() + off[b], entries.begin() + off[b + 1],
              [](const GridEntry& a, const GridEntry& c) { return a.rank < c.rank; });
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

void build_lut(thm_index* ix) {
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
  ix->lut.assign(nk, LutEntry{0, 0});
  // code_at[p] = 2-bit code of text[p..p+kt) or 0xFFFFFFFF if it is not all-ACGT
  std::vector<uint32_t> code_at(n, 0xFFFFFFFFu);
  {
    uint64_t code = 0;
    uint32_t run = 0;  // number of valid bases ending at p
    const uint64_t mask = nk - 1;
    for (uint64_t p = 0; p < n; p++) {
      int c = base_code(ix->text[p]);
      if (c < 0) {
        run = 0;
        code = 0;
      } else {
        code = ((code << 2) | (uint64_t)c) & mask;
        if (run < kt) run++;
      }
      if (run >= kt) code_at[p + 1 - kt] = (uint32_t)code;
    }
  }
  const uint32_t* sa = ix->sa.data();
  for (uint64_t r = 0; r < n; r++) {
    uint32_t c = code_at[sa[r]];
    if (c == 0xFFFFFFFFu) continue;
    LutEntry& e = ix->lut[c];
    if (e.hi == 0) e.lo = (uint32_t)r;
    e.hi = (uint32_t)r + 1;
  }
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

int32_t thm_index_create_in_memory(const uint8_t* text, uint64_t n, const thm_ref* refs, uint32_t n_refs,
                                   const thm_tx* txs, uint32_t n_txs, const thm_exon* exons, uint64_t n_exons,
                                   const uint8_t* tx_seq, uint64_t n_tx_seq, const thm_span* genes, uint32_t n_genes,
                                   const uint32_t* name_rank, uint32_t n_names, const uint32_t* sa, thm_index** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!text || n == 0 || !refs || n_refs == 0) {
    set_global_error("thm_index_create_in_memory: empty text or refs");
    return THM_ERR_INVALID_ARG;
  }
  if (n >= 0x7FFFFFF0ull) {
    set_global_error("text longer than 2^31 symbols: 64-bit suffix array not built in this round");
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
  for (uint32_t t = 0; t < n_txs; t++) {
    if (txs[t].exon_begin + txs[t].n_exons > n_exons || txs[t].seq_off + txs[t].seq_len > n_tx_seq ||
        txs[t].n_exons == 0 || (n_genes && txs[t].gene_idx >= n_genes)) {
      set_global_error("transcript table out of range");
      return THM_ERR_INVALID_ARG;
    }
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

  ix->sa.resize(n);
  if (sa) {
    if (!verify_suffix_array(text, n, sa)) {
      delete ix;
      set_global_error("supplied suffix array is not the suffix array of the text");
      return THM_ERR_INVALID_ARG;
    }
    memcpy(ix->sa.data(), sa, n * sizeof(uint32_t));
  } else if (build_suffix_array(text, n, ix->sa.data()) != 0) {
    delete ix;
    return THM_ERR_UNSUPPORTED;
  }
  build_lut(ix);

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
  build_grid(ix->exon_tree, ix->exon_root, n, ix->exon_grid_off, ix->exon_grid);
  build_grid(ix->gene_tree, ix->gene_root, n, ix->gene_grid_off, ix->gene_grid);
  ix->dev_mu = new std::mutex();
  *out = ix;
  return THM_OK;
}

uint64_t thm_index_text_len(const thm_index* ix) { return ix ? ix->n : 0; }
const uint32_t* thm_index_suffix_array(const thm_index* ix) { return ix ? ix->sa.data() : nullptr; }

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
