// io_inflate.cpp -- gzip (RFC 1952) / DEFLATE (RFC 1951) reader for the FASTQ batcher.
//
// The reference reads its query files through needletail::parse_fastx_file (reference src/aligner.rs:51-52), which
// hands `.gz` input to flate2.  Nothing of that crate is restated here: this is a table-driven inflate written for
// the one job the batcher has -- turn a gzip file into bytes as fast as one host thread can, so that inflating is
// not what the GPU waits for.  What makes it quick:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load (no per-byte loop);
//   * an 11-bit first-level literal/length table whose entries hold the symbol AND the bit count, with up to FOUR
//     literals in one entry where their codes fit the 11 bits together (FASTQ bases are 2- to 3-bit codes);
//   * length base / extra-bit count folded into the entry, so that a match costs two lookups and no further
//     table;
//   * 16-byte match copies that may overshoot (the caller's buffer has the slack), run fill for distance 1
//     (quality lines);
//   * CRC-32 by carry-less multiplication (PCLMULQDQ folding, the published Intel scheme) where the CPU has it.
// Every member's CRC-32 and ISIZE are checked; a truncated or corrupt stream is an error, never a short input.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>  // crc32() for the tail bytes and for CPUs without PCLMULQDQ

#include <sys/mman.h>
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "io_internal.h"
#include "thermite_internal.h"

#if defined(__x86_64__)
#include <immintrin.h>
#include <wmmintrin.h>
#endif

namespace thm {

// ---- CRC-32 ----
#if defined(__x86_64__)
#define THM_CLMUL __attribute__((target("sse4.2,pclmul")))
THM_CLMUL static inline __m128i ld128(const uint8_t* p) { return _mm_loadu_si128((const __m128i*)p); }
// x folded forward over the constants' distance, plus the next data y
THM_CLMUL static inline __m128i fold128(__m128i x, __m128i k, __m128i y) {
  return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), y);
}
// len >= 64 and a multiple of 16; crc is the running register (i.e. already inverted)
THM_CLMUL static uint32_t crc32_fold(const uint8_t* buf, size_t len, uint32_t crc) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};  // x^(512+64), x^512 mod P (reflected)
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};  // x^(128+64), x^128
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0};                // x^64
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};  // P and the Barrett constant
  __m128i x0 = _mm_load_si128((const __m128i*)k1k2);
  __m128i x1 = _mm_xor_si128(ld128(buf), _mm_cvtsi32_si128((int)crc)), x2 = ld128(buf + 16), x3 = ld128(buf + 32), x4 = ld128(buf + 48);
  buf += 64;
  len -= 64;
  while (len >= 64) {
    x1 = fold128(x1, x0, ld128(buf));
    x2 = fold128(x2, x0, ld128(buf + 16));
    x3 = fold128(x3, x0, ld128(buf + 32));
    x4 = fold128(x4, x0, ld128(buf + 48));
    buf += 64;
    len -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);
  x1 = fold128(x1, x0, x2);
  x1 = fold128(x1, x0, x3);
  x1 = fold128(x1, x0, x4);
  while (len >= 16) {
    x1 = fold128(x1, x0, ld128(buf));
    buf += 16;
    len -= 16;
  }
  // 128 -> 64 -> 32 bits, then Barrett reduction
  const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), x0, 0x00), x2);
  x0 = _mm_load_si128((const __m128i*)poly);
  x2 = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), x0, 0x10), m32);
  x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(x2, x0, 0x00));
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.2");
#endif

uint32_t crc32_fast(uint32_t crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  if (have_clmul && n >= 64) {
    const size_t m = n & ~(size_t)15;
    crc = ~crc32_fold(p, m, ~crc);
    p += m;
    n -= m;
  }
#endif
  while (n) {
    const unsigned k = (unsigned)(n > (1u << 30) ? (1u << 30) : n);
    crc = (uint32_t)crc32(crc, p, k);
    p += k;
    n -= k;
  }
  return crc;
}

// ---- tables ----
// Entry layout (u32), literal/length table:
//   bits 0..7   bits to take off the bit buffer (code length, plus the extra bits of a length code)
//   bits 8..11  code length (length codes only: the extra bits start there)
//   bit  12     exceptional: end of block (payload 0), sub-table (payload = first entry, bits 0..7 = index bits of
//               the sub-table, bits 8..11 = 0) or invalid code (payload 0xFFFF)
//   bit  15     literal
//   bits 16..31 payload: literal, or length base, or sub-table start
// Distance table: same, payload = distance base, never literals.
namespace {
constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr uint32_t E_LIT = 1u << 15, E_EXC = 1u << 12;
constexpr uint32_t E_INVALID = E_EXC | (0xFFFFu << 16) | 1u, E_EOB_PAYLOAD = 0;
// room for every sub-table any valid code can ask for (one of at most 2^4 / 2^7 entries per long symbol)
constexpr int MAX_LIT_ENTRIES = (1 << LIT_BITS) + 286 * 16, MAX_DIST_ENTRIES = (1 << DIST_BITS) + 30 * 128;

const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bitrev(uint32_t c, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; i++) r |= ((c >> i) & 1u) << (n - 1 - i);
  return r;
}

// symbol -> entry without the bit count
inline uint32_t lit_entry(int sym, int len) {
  if (sym < 256) return E_LIT | ((uint32_t)sym << 16) | (uint32_t)len;
  if (sym == 256) return E_EXC | (E_EOB_PAYLOAD << 16) | (uint32_t)len;
  if (sym > 285) return E_INVALID;
  const int k = sym - 257;
  return ((uint32_t)len_base[k] << 16) | ((uint32_t)len << 8) | (uint32_t)(len + len_extra[k]);
}
inline uint32_t dist_entry(int sym, int len) {
  if (sym > 29) return E_INVALID;
  return ((uint32_t)dist_base[sym] << 16) | ((uint32_t)len << 8) | (uint32_t)(len + dist_extra[sym]);
}

// Canonical Huffman decoding table from code lengths (0 = unused).  false: over-subscribed, or no room for the
// sub-tables.  An incomplete code leaves invalid entries behind (reported when the stream walks into one).
template <class MakeEntry>
bool build_table(const uint8_t* lens, int n_sym, int table_bits, uint32_t* table, int max_entries, MakeEntry make, bool* complete = nullptr) {
  int count[16] = {0};
  for (int i = 0; i < n_sym; i++) count[lens[i]]++;
  count[0] = 0;
  int left = 1;
  for (int l = 1; l <= 15; l++) {
    left = (left << 1) - count[l];
    if (left < 0) return false;
  }
  if (complete) *complete = left == 0;
  uint32_t next[16];
  uint32_t code = 0;
  for (int l = 1; l <= 15; l++) {
    code = (code + (uint32_t)count[l - 1]) << 1;
    next[l] = code;
  }
  const int primary = 1 << table_bits;
  for (int i = 0; i < primary; i++) table[i] = E_INVALID;
  int used = primary;
  // sub-tables: one per distinct table_bits-bit prefix (reversed: the LOW table_bits bits of the reversed code); the
  // codes of a prefix are consecutive in canonical order, so each sub-table is sized by the longest code under it
  // first pass: longest code per prefix
  uint8_t sub_bits[1 << LIT_BITS];
  bool any_long = false;
  for (int l = table_bits + 1; l <= 15; l++) any_long |= count[l] != 0;
  if (any_long) {
    memset(sub_bits, 0, (size_t)primary);
    uint32_t nx[16];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n_sym; s++) {
      const int l = lens[s];
      if (l <= table_bits) {
        if (l) nx[l]++;
        continue;
      }
      const uint32_t r = bitrev(nx[l]++, l), pre = r & (uint32_t)(primary - 1);
      if (l - table_bits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - table_bits);
    }
    for (int pre = 0; pre < primary; pre++)
      if (sub_bits[pre]) {
        const int n = 1 << sub_bits[pre];
        if (used + n > max_entries) return false;
        table[pre] = E_EXC | ((uint32_t)used << 16) | (uint32_t)sub_bits[pre];
        for (int i = 0; i < n; i++) table[used + i] = E_INVALID;
        used += n;
      }
  }
  for (int s = 0; s < n_sym; s++) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = bitrev(next[l]++, l);
    if (l <= table_bits) {
      const uint32_t e = make(s, l);
      for (uint32_t i = r; i < (uint32_t)primary; i += 1u << l) table[i] = e;
    } else {
      const uint32_t pre = r & (uint32_t)(primary - 1), top = table[pre];
      const int sb = (int)(top & 0xFF), l2 = l - table_bits;
      uint32_t* sub = table + (top >> 16);
      // the entry's bit count is that of the whole code: the sub-table lookup happens on the un-shifted buffer
      const uint32_t e = make(s, l);
      for (uint32_t i = r >> table_bits; i < (1u << sb); i += 1u << l2) sub[i] = e;
    }
  }
  return true;
}

// The first-level table the hot loop reads: 64-bit entries, the low half as above, and for literals up to FOUR of
// them in the high half (count in bits 8..10, bits 0..7 the sum of their code lengths) -- as many whole literal
// codes as the 11 index bits hold.  FASTQ is mostly literals with 2- to 4-bit codes.
void group_literals(const uint32_t* single, uint64_t* table) {
  for (uint32_t i = 0; i < (1u << LIT_BITS); i++) {
    const uint32_t e = single[i];
    if (!(e & E_LIT)) {
      table[i] = e;
      continue;
    }
    int bits = 0, count = 0;
    uint64_t lits = 0;
    while (count < 4) {
      const uint32_t e1 = single[i >> bits];  // (the index bits above the code are zero: any code they complete is too long)
      const int l = (int)(e1 & 0xFF);
      if (!(e1 & E_LIT) || bits + l > LIT_BITS) break;
      lits |= (uint64_t)((e1 >> 16) & 0xFF) << (8 * count);
      count++;
      bits += l;
    }
    table[i] = (uint64_t)(E_LIT | ((uint32_t)count << 8) | (uint32_t)bits) | (lits << 32);
  }
}

inline uint64_t load64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
}  // namespace

struct GzInflater::Impl {
  int fd = -1;
  std::string path;
  std::vector<uint8_t> inbuf;    // streaming input: compressed bytes [ipos, iend), then IN_PAD zero bytes once the file has ended
  const uint8_t* ib = nullptr;  // the input: inbuf.data(), or a whole file in memory (zero bytes readable behind its end)
  bool quiet = false;           // block search: failures are expected, no message is built
  bool strict = false;          // block search: only complete Huffman codes pass
  size_t ipos = 0, iend = 0;
  bool in_eof = false;
  uint64_t bitbuf = 0;
  int bitcnt = 0;
  enum State { MEMBER_HEADER, BLOCK_HEADER, STORED, HUFFMAN, TRAILER, END } state = MEMBER_HEADER;
  bool last_block = false;
  uint32_t stored_left = 0;
  uint32_t crc = 0;
  uint64_t member_out = 0;  // bytes of the current member (ISIZE is this modulo 2^32)
  uint64_t total_out = 0;
  uint64_t n_members = 0;
  uint32_t lit[MAX_LIT_ENTRIES], dist[MAX_DIST_ENTRIES];
  uint64_t lit4[1 << LIT_BITS];  // what the hot loop reads first (group_literals)
  bool fixed_loaded = false;
  std::string err;

  static constexpr size_t IN_CAP = 4u << 20, IN_PAD = 64;

  // at least `want` compressed bytes behind ipos, unless the file ends first (then zero padding stands behind iend)
  void fill(size_t want) {
    while (iend - ipos < want && !in_eof) {
      // (the 8 bytes before ipos stay: up to 7 whole bytes of them sit in the bit buffer and are handed back by
      // align_to_byte when a stored block or a member trailer follows)
      if (ipos > 8) {
        const size_t shift = ipos - 8;
        memmove(inbuf.data(), inbuf.data() + shift, iend - shift);
        iend -= shift;
        ipos = 8;
      }
      const long n = ::read(fd, inbuf.data() + iend, IN_CAP - iend);
      if (n < 0) {
        if (errno == EINTR) continue;
        err = "read error in " + path + ": " + strerror(errno);
        in_eof = true;
      } else if (n == 0) {
        in_eof = true;
      } else {
        iend += (size_t)n;
      }
      if (in_eof) memset(inbuf.data() + iend, 0, IN_PAD);
    }
  }
  // the bit reader's position as a byte position (whole bytes still in the bit buffer are handed back)
  void align_to_byte() {
    const int drop = bitcnt & 7;
    bitbuf >>= drop;
    bitcnt -= drop;
    ipos -= (size_t)(bitcnt >> 3);
    bitbuf = 0;
    bitcnt = 0;
  }
  bool bytes(size_t n) {  // n more compressed bytes present at ipos
    fill(n);
    return iend - ipos >= n;
  }
  bool fail(const char* what) {
    if (quiet) {
      if (err.empty()) err = "x";
      return false;
    }
    if (err.empty()) err = "gzip read error in " + path + ": " + what;
    return false;
  }
  // a whole file in memory: decoding starts at any bit
  void open_memory(const uint8_t* base, size_t size, const std::string& name) {
    ib = base;
    iend = size;
    in_eof = true;
    path = name;
  }
  void set_bit_position(uint64_t bit) {
    ipos = (size_t)(bit >> 3);
    const int skip = (int)(bit & 7);
    bitbuf = (uint64_t)(ib[ipos++] >> skip);
    bitcnt = 8 - skip;
  }
  uint64_t bit_position() const { return (uint64_t)ipos * 8 - (uint64_t)bitcnt; }

  bool member_header() {
    // between members: zero padding / nothing means the end
    if (!bytes(1)) {
      if (n_members == 0) return fail("empty file");
      state = END;
      return true;
    }
    if (!bytes(10)) return fail("truncated stream");
    const uint8_t* h = ib + ipos;
    if (h[0] != 0x1f || h[1] != 0x8b) {
      if (n_members == 0) return fail("not a gzip stream");
      // bytes behind the last member that are not a member: zlib's gzread ignores them ("trailing garbage")
      state = END;
      return true;
    }
    if (h[2] != 8) return fail("unknown compression method");
    const int flg = h[3];
    if (flg & 0xE0) return fail("reserved header flags set");
    ipos += 10;
    if (flg & 4) {  // FEXTRA
      if (!bytes(2)) return fail("truncated stream");
      const size_t xlen = ib[ipos] | ((size_t)ib[ipos + 1] << 8);
      ipos += 2;
      if (!bytes(xlen)) return fail("truncated stream");
      ipos += xlen;
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
      if (flg & f) {
        for (;;) {
          if (!bytes(1)) return fail("truncated stream");
          if (ib[ipos++] == 0) break;
        }
      }
    if (flg & 2) {  // FHCRC
      if (!bytes(2)) return fail("truncated stream");
      ipos += 2;
    }
    crc = 0;
    member_out = 0;
    n_members++;
    state = BLOCK_HEADER;
    return true;
  }

  // plain bit reading for the block headers (never the hot loop)
  bool need_bits(int n) {
    while (bitcnt < n) {
      if (!bytes(1)) return fail("truncated stream");
      bitbuf |= (uint64_t)ib[ipos++] << bitcnt;
      bitcnt += 8;
    }
    return true;
  }
  uint32_t take(int n) {
    const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1));
    bitbuf >>= n;
    bitcnt -= n;
    return v;
  }

  bool block_header() {
    if (!need_bits(3)) return false;
    last_block = take(1) != 0;
    const uint32_t type = take(2);
    if (type == 0) {
      align_to_byte();
      if (!bytes(4)) return fail("truncated stream");
      const uint32_t len = ib[ipos] | ((uint32_t)ib[ipos + 1] << 8), nlen = ib[ipos + 2] | ((uint32_t)ib[ipos + 3] << 8);
      if ((len ^ 0xFFFFu) != nlen) return fail("invalid stored block lengths");
      ipos += 4;
      stored_left = len;
      state = STORED;
      return true;
    }
    if (type == 1) {
      uint8_t l[288 + 32];
      int i = 0;
      for (; i < 144; i++) l[i] = 8;
      for (; i < 256; i++) l[i] = 9;
      for (; i < 280; i++) l[i] = 7;
      for (; i < 288; i++) l[i] = 8;
      build_table(l, 288, LIT_BITS, lit, MAX_LIT_ENTRIES, lit_entry);
      group_literals(lit, lit4);
      for (i = 0; i < 32; i++) l[i] = 5;
      build_table(l, 32, DIST_BITS, dist, MAX_DIST_ENTRIES, dist_entry);
      state = HUFFMAN;
      return true;
    }
    if (type == 3) return fail("invalid block type");
    if (!need_bits(14)) return false;
    const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
    if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) {
      if (!need_bits(3)) return false;
      cl[order[i]] = (uint8_t)take(3);
    }
    uint32_t clt[128 + 8];
    bool complete = false;
    if (!build_table(cl, 19, 7, clt, 128, [](int s, int l) { return ((uint32_t)s << 16) | (uint32_t)l; }, &complete) || (strict && !complete))
      return fail("invalid code lengths set");
    uint8_t lens[286 + 30 + 138];
    int n = 0;
    while (n < hlit + hdist) {
      if (!need_bits(7 + 7)) return false;  // (a block's end-of-block code and the member trailer always follow)
      const uint32_t e = clt[bitbuf & 127];
      if (e == E_INVALID) return fail("invalid code lengths set");
      const int l = (int)(e & 0xFF), sym = (int)(e >> 16);
      if (l > bitcnt) return fail("truncated stream");
      take(l);
      if (sym < 16) {
        lens[n++] = (uint8_t)sym;
        continue;
      }
      int rep, val = 0;
      if (sym == 16) {
        if (n == 0) return fail("invalid bit length repeat");
        if (bitcnt < 2) return fail("truncated stream");
        val = lens[n - 1];
        rep = 3 + (int)take(2);
      } else if (sym == 17) {
        if (bitcnt < 3) return fail("truncated stream");
        rep = 3 + (int)take(3);
      } else {
        if (bitcnt < 7) return fail("truncated stream");
        rep = 11 + (int)take(7);
      }
      if (n + rep > hlit + hdist) return fail("invalid bit length repeat");
      memset(lens + n, val, (size_t)rep);
      n += rep;
    }
    if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
    if (!build_table(lens, hlit, LIT_BITS, lit, MAX_LIT_ENTRIES, lit_entry, &complete) || (strict && !complete)) return fail("invalid literal/lengths set");
    group_literals(lit, lit4);
    int n_dist = 0;
    for (int i = 0; i < hdist; i++) n_dist += lens[hlit + i] != 0;
    // (one distance code of one bit, or none at all, is a valid incomplete set)
    if (!build_table(lens + hlit, hdist, DIST_BITS, dist, MAX_DIST_ENTRIES, dist_entry, &complete) || (strict && !complete && n_dist > 1)) return fail("invalid distances set");
    state = HUFFMAN;
    return true;
  }

  // The hot loop.  Runs while `out` is at least OUT_SLACK bytes below `out_end`; returns with the state moved on at
  // the end of the block, or unchanged when the output is full.  `floor` is the lowest address a match may reach.
  static constexpr size_t OUT_SLACK = 320;
  template <class T>
  static inline void put4(T* out, uint32_t v) {  // four literals (the caller advances by as many as are meant)
    if constexpr (sizeof(T) == 1) {
      memcpy(out, &v, 4);
    } else {  // bytes b3 b2 b1 b0 -> 00b3 00b2 00b1 00b0, one 8-byte store
      uint64_t x = v;
      x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
      x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
      memcpy(out, &x, 8);
    }
  }
  // T = uint8_t: bytes.  T = uint16_t: symbols, for decoding that starts in the middle of a stream -- the 32 Ki
  // entries before the output hold markers that stand for the unknown window, and are copied like bytes.
  template <class T>
  bool huffman(T*& out_io, T* out_end, const T* floor) {
    T* out = out_io;
    T* const out_stop = out_end - OUT_SLACK;
    uint64_t bb = bitbuf;
    int bc = bitcnt;
    const uint8_t* ip = ib + ipos;
    const uint8_t* ip_safe = ib + iend - (in_eof ? 0 : 32);  // past this: fetch more of the file first
    // at the end of the file the zero padding lets the loop read on; `ip_limit` is how far a valid stream can get
    const uint8_t* ip_limit = ib + iend + 8;
    const uint32_t* const lt = lit;
    const uint64_t* const l4 = lit4;
    const uint32_t* const dt = dist;
    bool ok = true;
    auto refill = [&] {
      bb |= load64(ip) << bc;
      ip += (63 - bc) >> 3;
      bc |= 56;
    };
    for (;;) {
      if (out >= out_stop) break;
      if (ip > ip_safe) {
        if (!in_eof) {
          // hand the position back, fetch, and go on
          ipos = (size_t)(ip - ib);
          fill(IN_CAP / 2);
          if (!err.empty()) {
            ok = false;
            break;
          }
          ip = ib + ipos;
          ip_safe = ib + iend - (in_eof ? 0 : 32);
          ip_limit = ib + iend + 8;
          continue;
        }
        // the bits taken so far must all have come out of the file (ip runs up to 8 bytes ahead of them)
        if (ip > ip_limit || (int64_t)(ip - ib) * 8 - bc > (int64_t)iend * 8) {
          ok = fail("truncated stream");
          break;
        }
      }
      refill();
      uint64_t e4 = l4[bb & ((1u << LIT_BITS) - 1)];
      // up to three table entries of literals per refill (3 x 11 bits at most)
      auto put_literals = [&] {
        put4(out, (uint32_t)(e4 >> 32));
        out += (e4 >> 8) & 7;
        bb >>= (e4 & 0xFF);
        bc -= (int)(e4 & 0xFF);
        e4 = l4[bb & ((1u << LIT_BITS) - 1)];
      };
      if (e4 & E_LIT) {
        put_literals();
        if (e4 & E_LIT) {
          put_literals();
          if (e4 & E_LIT) {
            put4(out, (uint32_t)(e4 >> 32));
            out += (e4 >> 8) & 7;
            bb >>= (e4 & 0xFF);
            bc -= (int)(e4 & 0xFF);
            continue;
          }
        }
        // at least 56 - 22 = 34 bits are left: enough for any literal/length code with its extra bits (20)
      }
      uint32_t e = (uint32_t)e4;
      if (e & E_EXC) {
        if ((e >> 16) == 0xFFFF) {
          ok = fail("invalid literal/length code");
          break;
        }
        if ((e >> 16) != 0) {  // sub-table (the end-of-block entry has payload 0)
          e = lt[(e >> 16) + ((bb >> LIT_BITS) & ((1u << (e & 0xFF)) - 1))];
          if (e & E_LIT) {
            *out++ = (T)((e >> 16) & 0xFF);
            bb >>= (e & 0xFF);
            bc -= (int)(e & 0xFF);
            continue;
          }
          if ((e & E_EXC) && (e >> 16) == 0xFFFF) {
            ok = fail("invalid literal/length code");
            break;
          }
        }
        if (e & E_EXC) {  // end of block
          bb >>= (e & 0xFF);
          bc -= (int)(e & 0xFF);
          state = last_block ? TRAILER : BLOCK_HEADER;
          break;
        }
      }
      // a length: base + extra bits
      const int cl = (int)((e >> 8) & 0xF), tot = (int)(e & 0xFF);
      uint32_t len = (e >> 16) + (uint32_t)((bb >> cl) & ((1u << (tot - cl)) - 1));
      bb >>= tot;
      bc -= tot;
      refill();
      uint32_t d = dt[bb & ((1u << DIST_BITS) - 1)];
      if (d & E_EXC) {
        if ((d >> 16) == 0xFFFF) {
          ok = fail("invalid distance code");
          break;
        }
        d = dt[(d >> 16) + ((bb >> DIST_BITS) & ((1u << (d & 0xFF)) - 1))];
        if (d & E_EXC) {
          ok = fail("invalid distance code");
          break;
        }
      }
      const int dcl = (int)((d >> 8) & 0xF), dtot = (int)(d & 0xFF);
      const uint32_t distance = (d >> 16) + (uint32_t)((bb >> dcl) & ((1u << (dtot - dcl)) - 1));
      bb >>= dtot;
      bc -= dtot;
      if ((size_t)(out - floor) < distance) {
        ok = fail("invalid distance too far back");
        break;
      }
      const T* src = out - distance;
      T* const oe = out + len;
      constexpr uint32_t E16 = 16 / sizeof(T), E8 = 8 / sizeof(T);  // elements per 16 / 8 bytes
      if (distance >= E16) {
        do {
          memcpy(out, src, 16);
          out += E16;
          src += E16;
        } while (out < oe);
      } else if (distance == 1) {
        const T v = *src;
        if constexpr (sizeof(T) == 1) {
          memset(out, v, len);
        } else {
          uint64_t x = v;
          x |= x << 16;
          x |= x << 32;
          T* q = out;
          do {  // (may overshoot like the other copies)
            memcpy(q, &x, 8);
            memcpy(q + 4, &x, 8);
            q += 8;
          } while (q < oe);
        }
      } else if (distance >= E8) {
        do {
          memcpy(out, src, 8);
          out += E8;
          src += E8;
        } while (out < oe);
      } else {
        do *out++ = *src++;
        while (out < oe);
      }
      out = oe;
    }
    bitbuf = bb;
    bitcnt = bc;
    ipos = (size_t)(ip - ib);
    out_io = out;
    return ok;
  }
};

GzInflater::GzInflater() : p_(new Impl()) {}
const std::string& GzInflater::error() const { return p_->err; }
long GzInflater::read(uint8_t* dst, size_t cap, size_t history) { return par_ ? par_read(dst, cap) : serial_read(*p_, dst, cap, history); }

long GzInflater::serial_read(Impl& s, uint8_t* dst, size_t cap, size_t history) {
  if (!s.err.empty()) return -1;
  if (cap < 2 * Impl::OUT_SLACK) {
    s.err = "internal: inflate buffer too small";
    return -1;
  }
  uint8_t* out = dst;
  uint8_t* const out_end = dst + cap;
  // a match reaches back into this member's output only
  const uint8_t* floor = dst - (s.state == Impl::MEMBER_HEADER || s.state == Impl::END ? 0 : std::min<uint64_t>(history, s.member_out));
  const uint8_t* crc_from = dst;
  auto crc_upto = [&](const uint8_t* upto) {
    s.crc = crc32_fast(s.crc, crc_from, (size_t)(upto - crc_from));
    s.member_out += (uint64_t)(upto - crc_from);
    crc_from = upto;
  };
  bool ok = true;
  while (ok && s.state != Impl::END && out < out_end - Impl::OUT_SLACK) {
    switch (s.state) {
      case Impl::MEMBER_HEADER:
        ok = s.member_header();
        // a member's matches must not reach into the member before it
        if (ok && s.state == Impl::BLOCK_HEADER) floor = out;
        break;
      case Impl::BLOCK_HEADER:
        ok = s.block_header();
        break;
      case Impl::STORED: {
        size_t n = std::min<size_t>(s.stored_left, (size_t)(out_end - out));
        s.fill(std::min<size_t>(n, Impl::IN_CAP / 2));
        n = std::min(n, s.iend - s.ipos);
        if (n == 0 && s.stored_left) {
          ok = s.fail("truncated stream");
          break;
        }
        memcpy(out, s.ib + s.ipos, n);
        out += n;
        s.ipos += n;
        s.stored_left -= (uint32_t)n;
        if (s.stored_left == 0) s.state = s.last_block ? Impl::TRAILER : Impl::BLOCK_HEADER;
        break;
      }
      case Impl::HUFFMAN:
        ok = s.huffman(out, out_end, floor);
        if (ok && !s.err.empty()) ok = false;
        break;
      case Impl::TRAILER: {
        s.align_to_byte();
        if (s.ipos > s.iend) {
          ok = s.fail("truncated stream");
          break;
        }
        if (!s.bytes(8)) {
          ok = s.fail("truncated stream");
          break;
        }
        crc_upto(out);
        const uint8_t* t = s.ib + s.ipos;
        const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t want_len = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (want_crc != s.crc) {
          ok = s.fail("incorrect data check");
          break;
        }
        if (want_len != (uint32_t)s.member_out) {
          ok = s.fail("incorrect length check");
          break;
        }
        s.ipos += 8;
        s.state = Impl::MEMBER_HEADER;
        break;
      }
      case Impl::END:
        break;
    }
  }
  if (!ok || !s.err.empty()) {
    if (s.err.empty()) s.err = "gzip read error in " + s.path;
    // what was inflated before the error is handed over first (a stream's bytes, then its error: the next call fails)
    if (out == dst) return -1;
    s.total_out += (uint64_t)(out - dst);
    return (long)(out - dst);
  }
  if (s.state != Impl::END) crc_upto(out);
  s.total_out += (uint64_t)(out - dst);
  return (long)(out - dst);
}




// ---- several threads on one gzip file ----
// A DEFLATE stream can only be decoded from its start: a match may reach 32 KiB back, into bytes a decoder that
// starts in the middle has not seen.  What such a decoder can do (the idea of pugz and rapidgzip, restated here from
// their published descriptions) is decode into 16-bit SYMBOLS, with 32 Ki marker symbols standing for the unknown
// window, and leave the markers to be replaced once the bytes before are known.  The file is cut into chunks of
// compressed bytes; for each chunk
//   find     the first bit at which a non-final dynamic-Huffman block starts: a header with complete codes, whose
//            block decodes without error and is followed by another valid header (worker thread);
//   decode   from there to the start found for the next chunk, into symbols; gzip member trailers and headers on
//            the way are recorded (worker thread);
//   resolve  symbols -> bytes with the last 32 KiB of the segment before, CRC-32 of the pieces (worker thread; only
//            the 32 KiB that the NEXT segment waits for are resolved on the consumer's thread, in stream order).
// The consumer hands the bytes out in order and checks every member's CRC-32 and ISIZE from the pieces
// (crc32_combine).  A segment that does not end exactly on the next start, or does not decode, is not trusted:
// from the end of the last good segment the rest of the file is inflated serially, window known, and a corrupt
// stream is reported by that decoder.
namespace {
struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  explicit Pool(unsigned n) {
    for (unsigned i = 0; i < n; i++)
      th.emplace_back([this] {
        for (;;) {
          std::function<void()> f;
          {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return stop || !q.empty(); });
            if (q.empty()) return;
            f = std::move(q.front());
            q.pop_front();
          }
          f();
        }
      });
  }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> g(mu);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};
constexpr size_t WINDOW = 32768;
constexpr uint16_t MARKER = 32768;  // symbols >= MARKER: window byte (symbol - MARKER)

// Segment buffers are kept for reuse at their size, across segments and across files: a fresh 20 MB buffer costs
// 5 000 page faults and their zero-filling before the decoder has written a symbol, which is most of the time of
// a short file.  A bounded number of them stays with the process.
template <class T>
struct BufferCache {
  std::mutex mu;
  std::vector<std::vector<T>> free;
  static constexpr size_t KEEP = 24;  // (two files with ten segments in flight each)
  void take(std::vector<T>& v) {
    std::lock_guard<std::mutex> g(mu);
    if (!free.empty()) {
      v.swap(free.back());
      free.pop_back();
    }
  }
  void give(std::vector<T>&& v) {
    if (v.capacity() == 0) return;
    std::lock_guard<std::mutex> g(mu);
    if (free.size() < KEEP) free.emplace_back(std::move(v));
  }
};
BufferCache<uint16_t> g_sym_cache;
BufferCache<uint8_t> g_byte_cache;
}  // namespace

struct GzInflater::Par {
  struct MemberEnd {
    size_t at;  // symbols of the segment before the member's end
    uint32_t crc, isize;
  };
  struct Chunk {
    uint64_t k = 0;
    // find
    bool find_done = false, found = false;
    bool at_member = false;  // the start is a gzip member header (bgzip files: every block is a member of its own)
    uint64_t start_bit = 0;
    // decode (chunks with a start)
    bool dispatched = false, dec_done = false, ok = false, at_end = false;
    uint64_t stop_bit = 0, end_bit = 0;
    std::vector<uint16_t> sym;  // WINDOW markers, then the segment's symbols
    size_t n = 0;
    std::vector<MemberEnd> members;
    std::string err;
    // resolve
    bool res_started = false, res_done = false;
    std::vector<uint8_t> window_in, bytes;
    std::vector<uint32_t> piece_crc;
  };
  const uint8_t* base = nullptr;
  size_t size = 0;
  void* map = nullptr;
  size_t map_len = 0;
  std::string path;
  size_t chunk_bytes = 0;
  uint64_t n_chunks = 0;
  unsigned lookahead = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, std::unique_ptr<Chunk>> chunks;
  bool cancel = false;
  uint64_t next_find = 1;   // next chunk to look for a start in
  uint64_t head = 0;        // chunk whose segment the consumer hands out next
  size_t head_off = 0;      // bytes of it already handed out
  uint64_t to_dispatch = 0; // lowest chunk with a start whose decode is not yet running
  uint64_t to_resolve = 0;  // next segment in stream order waiting for its window
  std::vector<uint8_t> window;  // last 32 KiB before segment `to_resolve`
  unsigned inflight = 0, max_inflight = 0;  // segments between the start of their decoding and the consumer
  size_t max_segment = (size_t)192 << 20;   // symbols one segment may grow to (THM_INFLATE_MAX_SEGMENT_MB for the tests)
  uint32_t crc = 0;
  uint64_t member_out = 0;
  bool finished = false;
  // serial tail (after a segment that could not be trusted)
  std::unique_ptr<Impl> tail;
  std::vector<uint8_t> tail_buf;
  size_t tail_hist = 0, tail_have = 0, tail_off = 0;
  std::unique_ptr<Pool> pool;  // last member: destroyed first, joins the workers

  ~Par() {
    {
      std::lock_guard<std::mutex> g(mu);
      cancel = true;
    }
    pool.reset();
    for (auto& kv : chunks) {
      g_sym_cache.give(std::move(kv.second->sym));
      g_byte_cache.give(std::move(kv.second->bytes));
    }
    if (map) munmap(map, map_len);
  }

  Chunk* get(uint64_t k) {  // (mu held)
    auto& c = chunks[k];
    if (!c) {
      c.reset(new Chunk());
      c->k = k;
    }
    return c.get();
  }

  // ---- find ----
  void find_task(Chunk* c) {
    uint64_t found_bit = 0;
    bool found = false, at_member = false;
    {
      std::unique_ptr<Impl> f(new Impl());
      f->open_memory(base, size, path);
      f->quiet = f->strict = true;
      std::vector<uint16_t> scratch(WINDOW + (1u << 20) + 1024, 0);
      const uint64_t lo = c->k * (uint64_t)chunk_bytes * 8, hi = std::min<uint64_t>((c->k + 1) * (uint64_t)chunk_bytes * 8, (uint64_t)size * 8);
      for (uint64_t bit = lo; bit < hi && !found; bit++) {
        const uint64_t w = load64(base + (bit >> 3)) >> (bit & 7);
        if ((bit & 7) == 0 && (w & 0xE0FFFFFFu) == 0x00088B1Fu && (bit >> 3) + 18 < size) {
          // a gzip member header?  it must parse, and its first block must decode
          f->err.clear();
          f->ipos = (size_t)(bit >> 3);
          f->bitbuf = 0;
          f->bitcnt = 0;
          f->n_members = 1;
          f->state = Impl::MEMBER_HEADER;
          if (f->member_header() && f->state == Impl::BLOCK_HEADER && f->block_header()) {
            bool good = true;
            if (f->state == Impl::HUFFMAN) {
              uint16_t* out = scratch.data() + WINDOW;
              good = f->huffman<uint16_t>(out, scratch.data() + scratch.size(), scratch.data()) && f->err.empty();
              if (good && f->state == Impl::BLOCK_HEADER) good = f->block_header();
            }
            if (good) {
              found = at_member = true;
              found_bit = bit;
              break;
            }
          }
        }
        if ((w & 7) != 4) continue;  // BFINAL = 0, BTYPE = 2
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        const int hclen = (int)((w >> 13) & 15) + 4;
        const uint64_t w2 = load64(base + ((bit + 17) >> 3)) >> ((bit + 17) & 7);  // (57 bits: 19 lengths of 3)
        uint32_t kraft = 0;
        for (int i = 0; i < hclen; i++) {
          const uint32_t l = (uint32_t)(w2 >> (3 * i)) & 7;
          if (l) kraft += 128u >> l;
        }
        if (kraft != 128) continue;  // the code-length code must be complete
        {
          std::lock_guard<std::mutex> g(mu);
          if (cancel) break;
        }
        f->err.clear();
        f->set_bit_position(bit);
        f->state = Impl::BLOCK_HEADER;
        if (!f->block_header() || f->state != Impl::HUFFMAN) continue;
        uint16_t* out = scratch.data() + WINDOW;
        if (!f->huffman<uint16_t>(out, scratch.data() + scratch.size(), scratch.data()) || !f->err.empty()) continue;
        if (f->state == Impl::BLOCK_HEADER) {  // the block ended: what follows must be a block header too
          if (!f->block_header()) continue;
        } else if (f->state != Impl::HUFFMAN) {
          continue;
        }  // (else: a million symbols without an error)
        found = true;
        found_bit = bit;
      }
    }
    {
      std::lock_guard<std::mutex> g(mu);
      c->found = found;
      c->at_member = at_member;
      c->start_bit = found_bit;
      c->find_done = true;
    }
    cv.notify_all();
  }

  // ---- decode ----
  void decode_task(Chunk* c) {
    std::unique_ptr<Impl> d(new Impl());
    d->open_memory(base, size, path);
    d->n_members = 1;
    if (c->at_member) {
      d->ipos = (size_t)(c->start_bit >> 3);
      d->state = Impl::MEMBER_HEADER;
    } else {
      d->set_bit_position(c->start_bit);
      d->state = Impl::BLOCK_HEADER;
    }
    std::vector<uint16_t>& sym = c->sym;
    g_sym_cache.take(sym);
    g_byte_cache.take(c->bytes);
    // (a segment of FASTQ inflates to 3 - 6 times its compressed length)
    {
      const size_t want = WINDOW + std::max<size_t>((size_t)((c->stop_bit ? c->stop_bit - c->start_bit : (uint64_t)chunk_bytes * 8) / 8) * 5, 1u << 20);
      if (sym.size() < want) sym.resize(want);
    }
    for (size_t i = 0; i < WINDOW; i++) sym[i] = (uint16_t)(MARKER + i);
    size_t n = 0;
    bool ok = true, at_end = false, reached = false;
    auto room = [&](size_t want) {
      if (sym.size() - (WINDOW + n) < want) sym.resize(sym.size() + sym.size() / 2 + want);
    };
    while (ok && !reached && !at_end) {
      switch (d->state) {
        case Impl::BLOCK_HEADER: {
          const uint64_t bp = d->bit_position();
          if (c->stop_bit && bp >= c->stop_bit) {
            ok = bp == c->stop_bit;
            reached = true;
            break;
          }
          ok = d->block_header();
          break;
        }
        case Impl::HUFFMAN: {
          room(1u << 18);
          uint16_t* out = sym.data() + WINDOW + n;
          ok = d->huffman<uint16_t>(out, sym.data() + sym.size(), sym.data()) && d->err.empty();
          n = (size_t)(out - (sym.data() + WINDOW));
          break;
        }
        case Impl::STORED: {
          const size_t len = d->stored_left;
          if (d->iend - d->ipos < len) {
            ok = d->fail("truncated stream");
            break;
          }
          room(len + 1024);
          uint16_t* out = sym.data() + WINDOW + n;
          for (size_t i = 0; i < len; i++) out[i] = d->ib[d->ipos + i];
          n += len;
          d->ipos += len;
          d->stored_left = 0;
          d->state = d->last_block ? Impl::TRAILER : Impl::BLOCK_HEADER;
          break;
        }
        case Impl::TRAILER: {
          d->align_to_byte();
          if (d->ipos + 8 > d->iend) {
            ok = d->fail("truncated stream");
            break;
          }
          const uint8_t* t = d->ib + d->ipos;
          MemberEnd e;
          e.at = n;
          e.crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
          e.isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
          c->members.push_back(e);
          d->ipos += 8;
          d->state = Impl::MEMBER_HEADER;
          break;
        }
        case Impl::MEMBER_HEADER:
          if (c->stop_bit && (uint64_t)d->ipos * 8 >= c->stop_bit && n + c->members.size() > 0) {
            ok = (uint64_t)d->ipos * 8 == c->stop_bit;
            reached = true;
            break;
          }
          ok = d->member_header();
          break;
        case Impl::END:
          at_end = true;
          break;
      }
      if (ok && (n & 0xFFFFF) < 0x800) {  // now and then
        std::lock_guard<std::mutex> g(mu);
        if (cancel) ok = false;
      }
      // A stream without block starts to find (stored or fixed-Huffman blocks only, or one endless block) would
      // become ONE segment of symbols, two bytes per output byte: beyond a bound the segment is given up and the
      // serial decoder takes the rest of the file, window known, in constant memory.
      if (n > max_segment) ok = d->fail("segment too long for the parallel decoder");
    }
    if (at_end && c->stop_bit) ok = false;  // the stream ended before the start the next segment was given
    {
      std::lock_guard<std::mutex> g(mu);
      c->n = n;
      c->ok = ok;
      c->at_end = at_end;
      c->end_bit = d->state == Impl::MEMBER_HEADER ? (uint64_t)d->ipos * 8 : d->bit_position();
      c->err = d->err;
      c->dec_done = true;
    }
    cv.notify_all();
  }

  static inline uint8_t resolve1(uint16_t v, const uint8_t* win) { return v < 256 ? (uint8_t)v : win[v & (WINDOW - 1)]; }

  // ---- resolve ----
  void resolve_task(Chunk* c) {
    const uint16_t* sy = c->sym.data() + WINDOW;
    // symbol -> byte through one table (literals map to themselves, markers to the window): a load per symbol, no branch
    std::vector<uint8_t> lut(65536, 0);
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    memcpy(lut.data() + MARKER, c->window_in.data(), WINDOW);
    if (c->bytes.size() < c->n + 64) c->bytes.resize(c->n + 64);
    uint8_t* b = c->bytes.data();
    const uint8_t* t = lut.data();
    size_t i = 0;
    for (; i + 8 <= c->n; i += 8) {
      b[i] = t[sy[i]];
      b[i + 1] = t[sy[i + 1]];
      b[i + 2] = t[sy[i + 2]];
      b[i + 3] = t[sy[i + 3]];
      b[i + 4] = t[sy[i + 4]];
      b[i + 5] = t[sy[i + 5]];
      b[i + 6] = t[sy[i + 6]];
      b[i + 7] = t[sy[i + 7]];
    }
    for (; i < c->n; i++) b[i] = t[sy[i]];
    size_t p = 0;
    for (const MemberEnd& e : c->members) {
      c->piece_crc.push_back(crc32_fast(0, b + p, e.at - p));
      p = e.at;
    }
    c->piece_crc.push_back(crc32_fast(0, b + p, c->n - p));
    g_sym_cache.give(std::move(c->sym));  // (kept at its size: the next segment does not touch fresh pages)
    c->sym = std::vector<uint16_t>();
    {
      std::lock_guard<std::mutex> g(mu);
      c->res_done = true;
    }
    cv.notify_all();
  }

  // mu held: start whatever can start.  false: the segment at `to_resolve` cannot be trusted
  bool advance(std::unique_lock<std::mutex>& g) {
    // finds, a bounded distance ahead of the consumer
    while (next_find < n_chunks && next_find < head + lookahead) {
      Chunk* c = get(next_find++);
      pool->submit([this, c] { find_task(c); });
    }
    // decodes: a chunk with a start, once the next start behind it is known (or the file has no further chunk)
    for (;;) {
      if (to_dispatch >= n_chunks || inflight >= max_inflight) break;
      Chunk* c = get(to_dispatch);
      uint64_t j = to_dispatch + 1;
      bool known = true;
      while (j < n_chunks) {
        auto it = chunks.find(j);
        if (it == chunks.end() || !it->second->find_done) {
          // (stretches without a start -- stored blocks, one huge block -- are searched on, beyond the look-ahead)
          if (it == chunks.end() && j == next_find) {
            Chunk* f = get(next_find++);
            pool->submit([this, f] { find_task(f); });
          }
          known = false;
          break;
        }
        if (it->second->found) break;
        j++;
      }
      if (!known) break;
      c->stop_bit = j < n_chunks ? chunks[j]->start_bit : 0;
      c->dispatched = true;
      inflight++;
      pool->submit([this, c] { decode_task(c); });
      to_dispatch = j;
    }
    // resolves, in stream order: the window of the next segment comes out of this one
    while (to_resolve < n_chunks) {
      auto it = chunks.find(to_resolve);
      if (it == chunks.end() || !it->second->dispatched || !it->second->dec_done) break;
      Chunk* c = it->second.get();
      if (!c->ok) return false;
      c->window_in = window;
      // the window behind this segment: its last 32 KiB (or the old window shifted by what it produced)
      if (c->n >= WINDOW) {
        const uint16_t* sy = c->sym.data() + WINDOW + c->n - WINDOW;
        std::vector<uint8_t> w(WINDOW);
        for (size_t i = 0; i < WINDOW; i++) w[i] = resolve1(sy[i], c->window_in.data());
        window.swap(w);
      } else {
        std::vector<uint8_t> w(WINDOW);
        memcpy(w.data(), c->window_in.data() + c->n, WINDOW - c->n);
        const uint16_t* sy = c->sym.data() + WINDOW;
        for (size_t i = 0; i < c->n; i++) w[WINDOW - c->n + i] = resolve1(sy[i], c->window_in.data());
        window.swap(w);
      }
      c->res_started = true;
      pool->submit([this, c] { resolve_task(c); });
      // the next segment in stream order: the next chunk with a start
      uint64_t j = to_resolve + 1;
      while (j < n_chunks) {
        auto jt = chunks.find(j);
        if (jt != chunks.end() && jt->second->find_done && jt->second->found) break;
        j++;  // (its find is done: this segment was dispatched with the start behind it known)
      }
      to_resolve = c->at_end ? n_chunks : j;
    }
    (void)g;
    return true;
  }
};

GzInflater::~GzInflater() {
  delete par_;  // (joins its threads, then unmaps the file; Par is complete here)
  if (p_->fd >= 0) close(p_->fd);
  delete p_;
}

bool GzInflater::open_parallel(int fd, const std::string& path, unsigned n_threads) {
  struct stat st;
  // compressed bytes per chunk; THM_INFLATE_CHUNK_KB lets the tests cut small files into many chunks
  size_t CHUNK = 1u << 20;
  if (const char* e = getenv("THM_INFLATE_CHUNK_KB")) CHUNK = (size_t)std::max(16L, atol(e)) << 10;
  if (n_threads < 2 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < 4 * CHUNK) return false;
  const size_t size = (size_t)st.st_size, page = (size_t)sysconf(_SC_PAGESIZE);
  // the file, and a page of zeros behind it: the decoders read a few bytes past the end of their input
  const size_t map_len = (size + page - 1) / page * page + page;
  void* m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return false;
  if (mmap(m, size, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0) == MAP_FAILED) {
    munmap(m, map_len);
    return false;
  }
  (void)madvise(m, size, MADV_SEQUENTIAL);
  std::unique_ptr<Par> par(new Par());
  par->map = m;
  par->map_len = map_len;
  par->base = (const uint8_t*)m;
  par->size = size;
  par->path = path;
  par->chunk_bytes = CHUNK;
  par->n_chunks = (size + CHUNK - 1) / CHUNK;
  par->lookahead = 4 * n_threads + 4;  // starts are searched this many chunks ahead (a search holds no buffer)
  par->max_inflight = n_threads + 2;   // segments being decoded, resolved or waiting for the consumer (they hold the buffers)
  par->window.assign(WINDOW, 0);
  if (const char* e = getenv("THM_INFLATE_MAX_SEGMENT_MB")) par->max_segment = (size_t)std::max(1L, atol(e)) << 20;
  // the first member's header: the first segment starts at its first block
  Impl h;
  h.open_memory(par->base, size, path);
  if (!h.member_header() || h.state != Impl::BLOCK_HEADER) return false;  // (the serial reader reports what is wrong)
  par->pool.reset(new Pool(n_threads));
  {
    std::unique_lock<std::mutex> g(par->mu);
    Par::Chunk* c0 = par->get(0);
    c0->find_done = c0->found = true;
    c0->start_bit = (uint64_t)h.ipos * 8;
    par->advance(g);
  }
  par_ = par.release();
  return true;
}

void GzInflater::open(int fd, const std::string& path, unsigned n_threads) {
  p_->fd = fd;
  p_->path = path;
  if (open_parallel(fd, path, n_threads)) return;
  p_->inbuf.resize(Impl::IN_CAP + Impl::IN_PAD);
  p_->ib = p_->inbuf.data();
}

long GzInflater::par_read(uint8_t* dst, size_t cap) {
  Par& P = *par_;
  Impl& s = *p_;  // (only its error string is used here)
  if (!s.err.empty()) return -1;
  if (cap == 0) return 0;
  for (;;) {
    if (P.tail) {  // serial from here on
      if (P.tail_off == P.tail_have) {
        // keep the last 32 KiB in front of the buffer, inflate behind them
        const size_t have = P.tail_hist + P.tail_have, keep = std::min(have, WINDOW);
        memmove(P.tail_buf.data() + WINDOW - keep, P.tail_buf.data() + WINDOW + P.tail_have - keep, keep);
        P.tail_hist = keep;
        const long n = serial_read(*P.tail, P.tail_buf.data() + WINDOW, P.tail_buf.size() - WINDOW, P.tail_hist);
        if (n < 0) {
          s.err = P.tail->err;
          return -1;
        }
        if (n == 0) return 0;
        P.tail_have = (size_t)n;
        P.tail_off = 0;
        if (!P.tail->err.empty()) s.err = P.tail->err;  // (bytes first, the error with the next call)
      }
      const size_t k = std::min(cap, P.tail_have - P.tail_off);
      memcpy(dst, P.tail_buf.data() + WINDOW + P.tail_off, k);
      P.tail_off += k;
      return (long)k;
    }
    if (P.finished) return 0;
    std::unique_lock<std::mutex> g(P.mu);
    bool trusted = P.advance(g);
    Par::Chunk* c = nullptr;
    if (trusted) {
      auto it = P.chunks.find(P.head);
      c = it == P.chunks.end() ? nullptr : it->second.get();
      if (!c || !c->res_done) {
        // wait for the head segment (or for the news that it cannot be trusted)
        P.cv.wait(g);
        continue;
      }
    }
    if (!trusted) {
      // Serial from the start of the segment at `to_resolve`: everything before it has been resolved and is handed out
      // first (head catches up with to_resolve), then the tail decoder takes over with the window known.
      if (P.head != P.to_resolve) {
        auto it = P.chunks.find(P.head);
        c = it == P.chunks.end() ? nullptr : it->second.get();
        if (!c || !c->res_done) {
          P.cv.wait(g);
          continue;
        }
      } else {
        Par::Chunk* bad = P.chunks[P.to_resolve].get();
        const uint64_t from = bad->start_bit;
        const bool from_member = bad->at_member || P.to_resolve == 0;
        P.cancel = true;
        g.unlock();
        P.pool.reset();  // (joins: nothing refers to the chunks any more)
        g.lock();
        P.chunks.clear();
        P.tail.reset(new Impl());
        P.tail->open_memory(P.base, P.size, P.path);
        P.tail->n_members = 1;
        if (from_member) {
          P.tail->ipos = P.to_resolve == 0 ? 0 : (size_t)(from >> 3);
          P.tail->n_members = P.to_resolve == 0 ? 0 : 1;
          P.tail->state = Impl::MEMBER_HEADER;
        } else {
          P.tail->set_bit_position(from);
          P.tail->state = Impl::BLOCK_HEADER;
        }
        P.tail->crc = P.crc;
        P.tail->member_out = P.member_out;
        P.tail_buf.resize(WINDOW + (4u << 20));
        memcpy(P.tail_buf.data(), P.window.data(), WINDOW);
        P.tail_hist = WINDOW;
        P.tail_have = P.tail_off = 0;
        continue;
      }
    }
    // hand out bytes of the head segment; at its end check the members that ended in it
    if (P.head_off < c->n) {
      const size_t k = std::min(cap, c->n - P.head_off);
      g.unlock();
      memcpy(dst, c->bytes.data() + P.head_off, k);
      P.head_off += k;
      return (long)k;
    }
    size_t p = 0, piece = 0;
    for (const Par::MemberEnd& e : c->members) {
      const size_t len = e.at - p;
      P.crc = (uint32_t)crc32_combine(P.crc, c->piece_crc[piece++], (z_off_t)len);
      P.member_out += len;
      if (P.crc != e.crc) s.err = "gzip read error in " + P.path + ": incorrect data check";
      else if ((uint32_t)P.member_out != e.isize) s.err = "gzip read error in " + P.path + ": incorrect length check";
      if (!s.err.empty()) return -1;
      P.crc = 0;
      P.member_out = 0;
      p = e.at;
    }
    P.crc = (uint32_t)crc32_combine(P.crc, c->piece_crc[piece], (z_off_t)(c->n - p));
    P.member_out += c->n - p;
    const bool at_end = c->at_end;
    // the next segment: the next chunk with a start
    uint64_t j = P.head + 1;
    while (j < P.n_chunks) {
      auto jt = P.chunks.find(j);
      if (jt != P.chunks.end() && jt->second->found) break;
      j++;
    }
    g_byte_cache.give(std::move(c->bytes));
    P.inflight--;
    for (uint64_t k = P.head; k < j; k++) P.chunks.erase(k);
    P.head = j;
    P.head_off = 0;
    if (at_end || P.head >= P.n_chunks) {
      // a stream that does not end in a member trailer is reported by its decode task (not ok -> serial tail)
      P.finished = true;
    }
  }
}

}  // namespace thm

// test hook: the whole gzip file through GzInflater in calls of `chunk` bytes (so that matches, stored blocks and
// members straddle calls); *n_out bytes land in out[0, cap).  THM_ERR_IO with the inflater's message on a bad stream.
extern "C" int32_t thm_debug_gunzip_mt(const char* path, uint64_t chunk, uint32_t n_threads, uint8_t* out, uint64_t cap, uint64_t* n_out);
extern "C" int32_t thm_debug_gunzip(const char* path, uint64_t chunk, uint8_t* out, uint64_t cap, uint64_t* n_out) {
  return thm_debug_gunzip_mt(path, chunk, 1, out, cap, n_out);
}
// ... with n_threads worker threads (the chunk-parallel decoder for files of four chunks or more)
extern "C" int32_t thm_debug_gunzip_mt(const char* path, uint64_t chunk, uint32_t n_threads, uint8_t* out, uint64_t cap, uint64_t* n_out) {
  if (!path || !out || !n_out || chunk < 1024) return THM_ERR_INVALID_ARG;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) {
    thm::set_global_error(std::string("cannot open ") + path);
    return THM_ERR_IO;
  }
  thm::GzInflater z;
  z.open(fd, path, n_threads);
  constexpr size_t W = 32768;
  std::vector<uint8_t> buf(W + chunk);
  size_t hist = 0;
  uint64_t total = 0;
  for (;;) {
    const long n = z.read(buf.data() + W, chunk, hist);
    if (n < 0) {
      *n_out = total;
      thm::set_global_error(z.error());
      return THM_ERR_IO;
    }
    if (n == 0) break;
    if (total + (uint64_t)n > cap) {
      thm::set_global_error("thm_debug_gunzip: output buffer too small");
      return THM_ERR_INVALID_ARG;
    }
    memcpy(out + total, buf.data() + W, (size_t)n);
    total += (uint64_t)n;
    const size_t have = hist + (size_t)n, keep = std::min(have, W);
    memmove(buf.data() + W - keep, buf.data() + W + (size_t)n - keep, keep);
    hist = keep;
  }
  *n_out = total;
  return THM_OK;
}
