// io_deflate.cpp -- DEFLATE (RFC 1951) compressor for the BGZF blocks of the BAM writer (csrc/io_writer.cpp).
//
// The reference writes BAM through noodles' bgzf writer, i.e. flate2 at its default level (src/aligner.rs:41-46).
// Parity is defined on the inflated stream; what this compressor has to be is quick, because sixteen formatting
// threads deflating with zlib were the slowest stage of FASTQ -> BAM by a factor of three.  One block of at most
// 64 KiB at a time, nothing carried from block to block (BGZF blocks are independent):
//   * greedy LZ77 over a hash of 4 bytes, one candidate per hash (no chains), matches extended 8 bytes at a time;
//   * one dynamic-Huffman block per BGZF block: symbol frequencies from the parse, code lengths by the two-queue
//     Huffman construction, limited to 15 (7 for the code-length code) on the counts per length so that the code
//     stays complete, the code lengths themselves run-length coded as the format provides;
//   * a 64-bit bit buffer flushed 4 bytes at a time;
//   * a stored block when that is smaller (packed bases with random qualities hardly compress).
// Output is checked by inflating it with zlib and with csrc/io_inflate.cpp (tests/test_io_host.py).
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "io_internal.h"

namespace thm {
namespace {

constexpr int HASH_BITS = 13;
constexpr uint32_t MAX_DIST = 32768, MAX_MATCH = 258;  // (matches are found through a hash of 4 bytes: none shorter)

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Tables {
  uint8_t len_sym[259];   // match length -> length symbol - 257
  uint8_t dist_sym[512];  // distances 1..256 by (d - 1), larger ones by 256 + ((d - 1) >> 7)
  Tables() {
    for (int s = 0; s < 29; s++)
      for (int l = LEN_BASE[s]; l <= (s == 28 ? 258 : LEN_BASE[s + 1] - 1); l++) len_sym[l] = (uint8_t)s;
    len_sym[258] = 28;
    for (int s = 0; s < 30; s++) {
      const int lo = DIST_BASE[s], hi = s == 29 ? 32768 : DIST_BASE[s + 1] - 1;
      for (int d = lo; d <= hi; d++) {
        if (d <= 256) dist_sym[d - 1] = (uint8_t)s;
        else dist_sym[256 + ((d - 1) >> 7)] = (uint8_t)s;  // (codes of 7 and more extra bits: whole 128-blocks)
      }
    }
  }
  int dist_code(uint32_t d) const { return d <= 256 ? dist_sym[d - 1] : dist_sym[256 + ((d - 1) >> 7)]; }
};
const Tables T;

// Huffman code lengths (at most max_len) for n symbols with the given frequencies; unused symbols get 0
void code_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len) {
  struct Node {
    uint64_t w;
    int left, right;  // children; leaves: left = -1, right = symbol
  };
  Node nodes[2 * 288];
  int leaves[288], n_leaves = 0;
  for (int i = 0; i < n; i++) {
    len[i] = 0;
    if (freq[i]) leaves[n_leaves++] = i;
  }
  if (n_leaves == 0) return;
  if (n_leaves == 1) {
    len[leaves[0]] = 1;
    return;
  }
  std::sort(leaves, leaves + n_leaves, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
  for (int i = 0; i < n_leaves; i++) nodes[i] = {freq[leaves[i]], -1, leaves[i]};
  // two queues: leaves in weight order, internal nodes in creation order (which is weight order too)
  int parent[2 * 288];
  int qa = 0, qb = n_leaves, next = n_leaves;
  auto take = [&] {
    if (qa < n_leaves && (qb >= next || nodes[qa].w <= nodes[qb].w)) return qa++;
    return qb++;
  };
  while ((n_leaves - qa) + (next - qb) > 1) {
    const int a = take(), b = take();
    nodes[next] = {nodes[a].w + nodes[b].w, a, b};
    parent[a] = parent[b] = next;
    next++;
  }
  // depths from the root (the last node) down, clamped to max_len; every clamped node counts as overflow, and the
  // overflow is resolved on the counts per length the way zlib's gen_bitlen does it: one leaf moves down a level, an
  // overflowing one becomes its brother -- the code stays complete
  uint8_t bits[2 * 288];
  int bl_count[16] = {0}, overflow = 0;
  bits[next - 1] = 0;
  for (int i = next - 2; i >= 0; i--) {
    int b = bits[parent[i]] + 1;
    if (b > max_len) {
      b = max_len;
      overflow++;
    }
    bits[i] = (uint8_t)b;
    if (i < n_leaves) bl_count[b]++;
  }
  while (overflow > 0) {
    int b = max_len - 1;
    while (bl_count[b] == 0) b--;
    bl_count[b]--;
    bl_count[b + 1] += 2;
    bl_count[max_len]--;
    overflow -= 2;
  }
  // the rarest symbols take the longest codes
  int h = 0;
  for (int b = max_len; b >= 1; b--)
    for (int k = 0; k < bl_count[b]; k++) len[leaves[h++]] = (uint8_t)b;
}

// canonical codes, bit-reversed for the LSB-first bit stream
void make_codes(const uint8_t* len, int n, uint16_t* code) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[len[i]]++;
  count[0] = 0;
  uint32_t next[16], c = 0;
  for (int l = 1; l <= 15; l++) {
    c = (c + (uint32_t)count[l - 1]) << 1;
    next[l] = c;
  }
  for (int i = 0; i < n; i++) {
    const int l = len[i];
    if (!l) {
      code[i] = 0;
      continue;
    }
    uint32_t v = next[l]++, r = 0;
    for (int b = 0; b < l; b++) r |= ((v >> b) & 1u) << (l - 1 - b);
    code[i] = (uint16_t)r;
  }
}

struct BitWriter {
  uint8_t* p;
  uint64_t buf = 0;
  int cnt = 0;
  explicit BitWriter(uint8_t* out) : p(out) {}
  inline void put(uint32_t v, int n) {  // n <= 32
    buf |= (uint64_t)v << cnt;
    cnt += n;
    if (cnt >= 32) {
      memcpy(p, &buf, 4);
      p += 4;
      buf >>= 32;
      cnt -= 32;
    }
  }
  uint8_t* finish() {
    while (cnt > 0) {
      *p++ = (uint8_t)buf;
      buf >>= 8;
      cnt -= 8;
    }
    return p;
  }
};

inline uint32_t load32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
inline uint64_t load64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

}  // namespace

size_t deflate_block_bound(size_t n) { return n + n / 8 + 512; }

size_t deflate_block(const uint8_t* in, size_t n, uint8_t* out, DeflateScratch& sc) {
  if (n == 0) {  // an empty fixed-Huffman block: what every deflate writes for no input (the BGZF end-of-file marker is these bytes)
    out[0] = 3;
    out[1] = 0;
    return 2;
  }
  // ---- parse ----
  uint32_t* tok = sc.tok;  // literal: the byte; match: 1 << 31 | (len - 3) << 16 | (dist - 1)
  size_t n_tok = 0;
  uint32_t lfreq[288] = {0}, dfreq[32] = {0};
  uint16_t* head = sc.head;
  memset(head, 0xFF, sizeof(uint16_t) << HASH_BITS);
  size_t i = 0;
  const size_t last_hashable = n >= 8 ? n - 8 : 0;  // (8 bytes are loaded at a match candidate)
  while (i < n) {
    if (i < last_hashable) {
      const uint32_t v = load32(in + i);
      const uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
      const uint32_t cand = head[h];
      head[h] = (uint16_t)i;
      if (cand != 0xFFFF && i - cand <= MAX_DIST && load32(in + cand) == v) {
        // extend: 8 bytes at a time, never past the end of the block
        size_t len = 4;
        const size_t max_len = std::min<size_t>(MAX_MATCH, n - i);
        while (len + 8 <= max_len) {
          const uint64_t x = load64(in + i + len) ^ load64(in + cand + len);
          if (x) {
            len += (size_t)(__builtin_ctzll(x) >> 3);
            goto extended;
          }
          len += 8;
        }
        while (len < max_len && in[i + len] == in[cand + len]) len++;
      extended:
        const uint32_t dist = (uint32_t)(i - cand);
        tok[n_tok++] = (1u << 31) | ((uint32_t)(len - 3) << 16) | (dist - 1);
        lfreq[257 + T.len_sym[len]]++;
        dfreq[T.dist_code(dist)]++;
        // (one more position of the match goes into the table: the next record's same field often starts inside)
        if (i + 1 < last_hashable) head[(load32(in + i + 1) * 2654435761u) >> (32 - HASH_BITS)] = (uint16_t)(i + 1);
        i += len;
        continue;
      }
    }
    tok[n_tok++] = in[i];
    lfreq[in[i]]++;
    i++;
  }
  lfreq[256] = 1;
  // ---- codes ----
  uint8_t llen[288], dlen[32];
  uint16_t lcode[288], dcode[32];
  code_lengths(lfreq, 286, 15, llen);
  code_lengths(dfreq, 30, 15, dlen);
  int n_d = 0;
  for (int k = 0; k < 30; k++) n_d += dlen[k] != 0;
  if (n_d == 0) dlen[0] = 1;  // (at least one distance code must be sent)
  make_codes(llen, 286, lcode);
  make_codes(dlen, 30, dcode);
  int hlit = 286, hdist = 30;
  while (hlit > 257 && llen[hlit - 1] == 0) hlit--;
  while (hdist > 1 && dlen[hdist - 1] == 0) hdist--;
  // the code lengths, run-length coded (symbols 16: repeat previous 3-6, 17: zeros 3-10, 18: zeros 11-138)
  uint8_t all[288 + 32], rl_sym[288 + 32], rl_extra[288 + 32];
  memcpy(all, llen, (size_t)hlit);
  memcpy(all + hlit, dlen, (size_t)hdist);
  const int n_all = hlit + hdist;
  int n_rl = 0;
  uint32_t cfreq[19] = {0};
  for (int k = 0; k < n_all;) {
    int run = 1;
    while (k + run < n_all && all[k + run] == all[k]) run++;
    if (all[k] == 0 && run >= 3) {
      const int r = std::min(run, 138);
      rl_sym[n_rl] = r <= 10 ? 17 : 18;
      rl_extra[n_rl++] = (uint8_t)(r <= 10 ? r - 3 : r - 11);
      k += r;
    } else if (run >= 4) {  // the value once, then repeats of 3..6
      rl_sym[n_rl] = all[k];
      rl_extra[n_rl++] = 0;
      int left = run - 1;
      k += 1;
      while (left >= 3) {
        const int r = std::min(left, 6);
        rl_sym[n_rl] = 16;
        rl_extra[n_rl++] = (uint8_t)(r - 3);
        left -= r;
        k += r;
      }
    } else {
      rl_sym[n_rl] = all[k];
      rl_extra[n_rl++] = 0;
      k += 1;
    }
  }
  for (int k = 0; k < n_rl; k++) cfreq[rl_sym[k]]++;
  uint8_t clen[19];
  uint16_t ccode[19];
  code_lengths(cfreq, 19, 7, clen);
  make_codes(clen, 19, ccode);
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && clen[order[hclen - 1]] == 0) hclen--;
  // ---- cost: stored when that is smaller ----
  uint64_t bits = 3 + 14 + 3 * (uint64_t)hclen;
  for (int k = 0; k < n_rl; k++) bits += clen[rl_sym[k]] + (rl_sym[k] == 16 ? 2 : rl_sym[k] == 17 ? 3 : rl_sym[k] == 18 ? 7 : 0);
  for (int s = 0; s < 286; s++) bits += (uint64_t)lfreq[s] * (llen[s] + (s > 264 ? LEN_EXTRA[s - 257] : 0));
  for (int s = 0; s < 30; s++) bits += (uint64_t)dfreq[s] * (dlen[s] + DIST_EXTRA[s]);
  if ((bits + 7) / 8 >= n + 5) {
    out[0] = 1;  // BFINAL = 1, BTYPE = 00; the rest of the byte is padding
    out[1] = (uint8_t)(n & 0xFF);
    out[2] = (uint8_t)(n >> 8);
    out[3] = (uint8_t)(~n & 0xFF);
    out[4] = (uint8_t)((~n >> 8) & 0xFF);
    memcpy(out + 5, in, n);
    return n + 5;
  }
  // ---- emit ----
  BitWriter w(out);
  w.put(1, 1);  // BFINAL
  w.put(2, 2);  // dynamic Huffman
  w.put((uint32_t)(hlit - 257), 5);
  w.put((uint32_t)(hdist - 1), 5);
  w.put((uint32_t)(hclen - 4), 4);
  for (int k = 0; k < hclen; k++) w.put(clen[order[k]], 3);
  for (int k = 0; k < n_rl; k++) {
    w.put(ccode[rl_sym[k]], clen[rl_sym[k]]);
    if (rl_sym[k] == 16) w.put(rl_extra[k], 2);
    else if (rl_sym[k] == 17) w.put(rl_extra[k], 3);
    else if (rl_sym[k] == 18) w.put(rl_extra[k], 7);
  }
  for (size_t k = 0; k < n_tok; k++) {
    const uint32_t t = tok[k];
    if (!(t >> 31)) {
      w.put(lcode[t], llen[t]);
      continue;
    }
    const uint32_t len = ((t >> 16) & 0xFF) + 3, dist = (t & 0xFFFF) + 1;
    const int ls = T.len_sym[len], ds = T.dist_code(dist);
    // length code + its extra bits in one call (at most 15 + 5), distance code + extra in another (15 + 13)
    w.put(lcode[257 + ls] | ((len - LEN_BASE[ls]) << llen[257 + ls]), llen[257 + ls] + LEN_EXTRA[ls]);
    w.put(dcode[ds] | ((dist - DIST_BASE[ds]) << dlen[ds]), dlen[ds] + DIST_EXTRA[ds]);
  }
  w.put(lcode[256], llen[256]);
  return (size_t)(w.finish() - out);
}

}  // namespace thm

// test hook: one DEFLATE block (raw, no gzip wrapper) of at most 65280 bytes -> *n_out bytes in out[0, cap)
extern "C" int32_t thm_debug_deflate_block(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* n_out) {
  if (!in || !out || !n_out || n > 0xff00 || cap < thm::deflate_block_bound((size_t)n)) return THM_ERR_INVALID_ARG;
  thm::DeflateScratch sc;
  *n_out = thm::deflate_block(in, (size_t)n, out, sc);
  return THM_OK;
}
