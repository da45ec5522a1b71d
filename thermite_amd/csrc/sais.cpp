// sais.cpp -- suffix array construction by induced sorting (SA-IS).
//
// Offline index-build step.  The reference calls libdivsufsort
// (reference src/index.rs:103-105); the suffix array is mathematically unique,
// so any correct builder yields the same array.  Plain byte order, a shorter
// suffix sorts before any longer suffix it prefixes (the implicit end-of-text
// sentinel is the smallest symbol) -- the order divsufsort produces.
//
// Host code, no device dependency.
#include <cstdint>
#include <cstring>
#include <vector>

#include "thermite_internal.h"

namespace thm {
namespace {

// I = index type: int32_t for texts below 2^31 symbols, int64_t above (the reference builds with
// divsufsort64, src/index.rs:103-105, 364-380)
template <class CharT, class I>
struct Sais {
  const CharT* s;
  I* SA;  // n + 1 entries; SA[0] is the virtual sentinel suffix n
  I n, K;
  std::vector<uint8_t> t;  // 1 = S-type, 0 = L-type; t[n] = S
  std::vector<I> cnt, bkt;

  inline bool is_lms(I i) const { return i > 0 && t[i] && !t[i - 1]; }

  void buckets(bool end) {
    I sum = 1;  // slot 0 belongs to the sentinel
    for (I c = 0; c < K; c++) {
      sum += cnt[c];
      bkt[c] = end ? sum : sum - cnt[c];
    }
  }
  void induce() {
    buckets(false);
    for (I i = 0; i <= n; i++) {
      I j = SA[i];
      if (j > 0 && !t[j - 1]) SA[bkt[s[j - 1]]++] = j - 1;
    }
    buckets(true);
    for (I i = n; i >= 1; i--) {
      I j = SA[i];
      if (j > 0 && t[j - 1]) SA[--bkt[s[j - 1]]] = j - 1;
    }
  }
  bool lms_equal(I a, I b) const {
    if (a == n || b == n) return false;
    for (I d = 0;; d++) {
      I pa = a + d, pb = b + d;
      if (pa == n || pb == n) return false;
      if (s[pa] != s[pb] || t[pa] != t[pb]) return false;
      if (d > 0) {
        bool la = is_lms(pa), lb = is_lms(pb);
        if (la || lb) return la && lb;
      }
    }
  }

  void run() {
    if (n == 0) {
      SA[0] = 0;
      return;
    }
    t.assign((size_t)n + 1, 0);
    t[n] = 1;
    t[n - 1] = 0;
    for (I i = n - 2; i >= 0; i--) t[i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && t[i + 1])) ? 1 : 0;
    cnt.assign(K, 0);
    bkt.assign(K, 0);
    for (I i = 0; i < n; i++) cnt[s[i]]++;

    // 1. sort the LMS substrings
    for (I i = 0; i <= n; i++) SA[i] = -1;
    SA[0] = n;
    buckets(true);
    I n1 = 1;
    for (I i = 1; i < n; i++)
      if (is_lms(i)) {
        SA[--bkt[s[i]]] = i;
        n1++;
      }
    induce();

    // 2. name them
    std::vector<I> sorted_lms;
    sorted_lms.reserve(n1);
    for (I i = 0; i <= n; i++)
      if (is_lms(SA[i])) sorted_lms.push_back(SA[i]);
    std::vector<I> name_at((size_t)n / 2 + 2, -1);
    I name = 0;
    name_at[sorted_lms[0] >> 1] = 0;
    for (I r = 1; r < n1; r++) {
      if (!lms_equal(sorted_lms[r - 1], sorted_lms[r])) name++;
      name_at[sorted_lms[r] >> 1] = name;
    }
    // reduced string in text order; its last symbol is the sentinel (name 0, unique)
    std::vector<I> lms_pos;
    lms_pos.reserve(n1);
    for (I i = 1; i <= n; i++)
      if (is_lms(i)) lms_pos.push_back(i);
    std::vector<I> sa1((size_t)n1);
    if (name + 1 == n1) {
      for (I k = 0; k < n1; k++) sa1[name_at[lms_pos[k] >> 1]] = k;
    } else {
      std::vector<I> s1((size_t)n1 - 1);
      for (I k = 0; k + 1 < n1; k++) s1[k] = name_at[lms_pos[k] >> 1] - 1;
      Sais<I, I> sub;
      sub.s = s1.data();
      sub.SA = sa1.data();
      sub.n = n1 - 1;
      sub.K = name;
      sub.run();
    }
    std::vector<I>().swap(name_at);
    std::vector<I>().swap(sorted_lms);

    // 3. induce the full order from the sorted LMS suffixes
    for (I i = 0; i <= n; i++) SA[i] = -1;
    SA[0] = n;
    buckets(true);
    for (I r = n1 - 1; r >= 1; r--) {
      I p = lms_pos[sa1[r]];
      SA[--bkt[s[p]]] = p;
    }
    induce();
  }
};

}  // namespace

int build_suffix_array(const uint8_t* text, uint64_t n, uint32_t* out) {
  if (n >= 0x7FFFFFF0ull) return -4;
  std::vector<int32_t> sa((size_t)n + 1);
  Sais<uint8_t, int32_t> top;
  top.s = text;
  top.SA = sa.data();
  top.n = (int32_t)n;
  top.K = 256;
  top.run();
  for (uint64_t i = 0; i < n; i++) out[i] = (uint32_t)sa[i + 1];
  return 0;
}

// 64-bit ranks and positions: any text length (memory: 9 bytes per symbol plus the recursion)
int build_suffix_array64(const uint8_t* text, uint64_t n, uint64_t* out) {
  if (n >= (1ull << 62)) return -4;
  std::vector<int64_t> sa((size_t)n + 1);
  Sais<uint8_t, int64_t> top;
  top.s = text;
  top.SA = sa.data();
  top.n = (int64_t)n;
  top.K = 256;
  top.run();
  for (uint64_t i = 0; i < n; i++) out[i] = (uint64_t)sa[i + 1];
  return 0;
}

}  // namespace thm
