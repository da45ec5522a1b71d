// thermite_internal.h -- structures shared by the host index builder, the
// launch code and the HIP kernels.  Not part of the ABI (include/thermite.h is).
#ifndef THERMITE_INTERNAL_H
#define THERMITE_INTERNAL_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/thermite.h"

namespace thm {

// ---- scoring: Scoring::from_scores(-1, -1, 1, -1), reference src/aligner.rs:140 ----
constexpr int32_t GAP_OPEN = -1;
constexpr int32_t GAP_EXTEND = -1;
constexpr int32_t MATCH_SCORE = 1;
constexpr int32_t MISMATCH_SCORE = -1;
// bio::alignment::pairwise::MIN_SCORE (reference src/swg.rs:1)
constexpr int32_t MIN_SCORE = -858993459;

// ---- flattened interval tree (same shape and visit order as bio's AVL
// IntervalTree built by the reference, src/index.rs:135,182-183,208-213); host
// side only: the device uses the interval grids derived from it ----
struct TreeNode {
  uint64_t start, end, max;
  uint32_t value;
  int32_t left, right;  // node indices, -1 = none
  int32_t pad_;
};
static_assert(sizeof(TreeNode) == 40, "TreeNode layout");

// Interval grid: the device-side form of an interval tree.  The text is cut into
// bins of 2^GRID_SHIFT symbols; a bin lists every interval overlapping it, sorted
// by the interval's rank in the (node, right, left) pre-order of the AVL tree --
// the order bio's IntervalTree::find yields overlapping intervals in (pruning
// only skips subtrees, it never reorders).  A query reads one or two adjacent
// bins in two memory round trips instead of walking ~log2(n) dependent nodes.
constexpr uint32_t GRID_SHIFT = 10;

// Coordinate width.  C = uint32_t: text positions and suffix-array ranks fit 32 bits (text below 2^31
// symbols; the default: half the index bytes, and wave-uniform coordinate arithmetic stays on the 32-bit
// scalar unit).  C = uint64_t: any text length -- the reference's usize coordinates (src/index.rs:364-388;
// a GRCh38-sized text has 6.2 G symbols).  Every structure that holds a text position or a rank is
// templated on C, and so are the kernels that read them.
template <class C>
struct GridEntryT {
  C start, end;
  uint32_t value;
  uint32_t rank;  // (pre-order rank << 8) | (bin & 0xff): the low byte tells the copies of one interval apart
};
static_assert(sizeof(GridEntryT<uint32_t>) == 16 && sizeof(GridEntryT<uint64_t>) == 24, "GridEntry layout");

// Entry of the exon grid (exon_to_tx): the interval is one exon of one transcript, and the entry carries what
// align_seed_hit needs from that transcript and exon, so that the common case (the seed lies in this exon, the
// alignment stays inside it) costs no further dependent loads of transcript and exon records:
//   lift_mem_to_tx (src/txome.rs:82-103) takes the FIRST exon in transcript order that intersects the seed; exons of a
//   transcript ascend in concatenated coordinates (src/index.rs:149-195), so that is this exon unless the previous
//   exon reaches into the seed (prev_end > seed start; an index whose exons do not ascend sets prev_end to the
//   maximum, which always takes the general path).
template <class C>
struct ExonEntryT {
  C start, end;
  uint32_t value;     // tx_idx
  uint32_t rank;      // as in GridEntryT
  C prev_end;         // end of the previous exon of the transcript, 0 if this is its first
  uint32_t txoff;     // transcript offset of the exon's first base (exon_txoff)
  uint32_t exon_idx;  // index of the exon within the transcript
  uint64_t seq_off;   // Tx::seq = tx_seq[seq_off .. seq_off + seq_len)
  uint32_t seq_len;
  uint32_t n_exons;
};
static_assert(sizeof(ExonEntryT<uint32_t>) == 48 && sizeof(ExonEntryT<uint64_t>) == 56, "ExonEntry layout");

// Index::idx_to_ref without a search: ref_bin[idx >> GRID_SHIFT] is the contig copy that holds the first symbol of
// the bin (the one that holds idx is that one or, when a boundary falls into the bin, a later one), RefRecT what
// the kernels need of a Ref (src/index.rs:391-399) in one load.
template <class C>
struct RefRecT {
  C start, end, len;
  uint32_t name_rank;
  uint32_t strand;
};

// k-mer prefix table entry: suffix-array interval of one ACGT-only kt-mer
template <class C>
struct LutEntryT {
  C lo, hi;
};

// device-side view of the index (all pointers in HBM)
template <class C>
struct DeviceIndexT {
  typedef C coord_t;
  const uint8_t* text;  // n symbols + 128 bytes of '$' padding
  const C* sa;          // n
  const LutEntryT<C>* lut;  // 4^kt
  const thm_ref* refs;
  const uint32_t* name_rank;  // per ref
  const RefRecT<C>* ref_recs;  // per ref
  const uint32_t* ref_bin;     // [n_bins]
  const thm_tx* txs;
  const thm_exon* exons;
  const uint64_t* exon_txoff;  // per exon: offset of its first base in the transcript
  const uint8_t* tx_seq;
  const uint32_t* exon_grid_off;  // [n_bins + 1]
  const ExonEntryT<C>* exon_grid;
  const uint32_t* gene_grid_off;
  const GridEntryT<C>* gene_grid;
  uint64_t n;
  uint32_t n_refs, n_txs;
  uint32_t kt;
  uint32_t max_tx_exons;  // most exons any transcript has (bounds the introns one alignment can span)
};

// one SMEM as the seed kernel emits it: occurrences are sa[lo..hi)
template <class C>
struct SmemT {
  C lo, hi;
  uint16_t qpos, len;
};
static_assert(sizeof(SmemT<uint32_t>) == 12 && sizeof(SmemT<uint64_t>) == 24, "Smem layout");

int build_suffix_array(const uint8_t* text, uint64_t n, uint32_t* out);
int build_suffix_array64(const uint8_t* text, uint64_t n, uint64_t* out);
template <class C>
bool verify_suffix_array(const uint8_t* text, uint64_t n, const C* sa);

void set_global_error(const std::string& msg);

// sa_gpu.hip: suffix array by prefix doubling on the current HIP device; nonzero: sort on the host instead
int build_suffix_array_gpu(const uint8_t* text, uint64_t n, void* out, int elem_bytes);

}  // namespace thm

// Host index: owns the tables; device copies are created lazily per device.
struct thm_index {
  std::vector<uint8_t> text;
  // exactly one of the two coordinate widths is populated (thermite_internal.h, "Coordinate width")
  bool wide = false;
  std::vector<uint32_t> sa;
  std::vector<uint64_t> sa64;
  std::vector<thm::LutEntryT<uint32_t>> lut;
  std::vector<thm::LutEntryT<uint64_t>> lut64;
  uint32_t kt = 0;
  uint32_t max_tx_exons = 0;
  std::vector<thm_ref> refs;
  std::vector<uint32_t> name_rank;  // per ref
  std::vector<thm_tx> txs;
  std::vector<thm_exon> exons;
  std::vector<uint64_t> exon_txoff;
  std::vector<uint8_t> tx_seq;
  std::vector<thm_span> genes;
  std::vector<thm::TreeNode> exon_tree, gene_tree;
  int32_t exon_root = -1, gene_root = -1;
  std::vector<uint32_t> exon_grid_off, gene_grid_off;
  std::vector<thm::ExonEntryT<uint32_t>> exon_grid;
  std::vector<thm::ExonEntryT<uint64_t>> exon_grid64;
  std::vector<thm::GridEntryT<uint32_t>> gene_grid;
  std::vector<thm::GridEntryT<uint64_t>> gene_grid64;
  std::vector<uint32_t> exon_node_exon;  // exon_tree node -> index into exons[]
  std::vector<uint32_t> ref_bin;
  std::vector<thm::RefRecT<uint32_t>> ref_recs;
  std::vector<thm::RefRecT<uint64_t>> ref_recs64;
  uint64_t n = 0;
  // names for the writer (Ref::name, Tx::id, Gene::{id,name}); empty when not supplied
  std::vector<std::string> contig_names, tx_ids, gene_ids, gene_names;
  // per-device uploaded copy (guarded by dev_mu)
  struct DevCopy;
  std::vector<DevCopy*> dev;  // indexed by device id
  void* dev_mu = nullptr;     // std::mutex*
};

#endif
