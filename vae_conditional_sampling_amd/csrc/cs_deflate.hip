// Host code: a deflate (RFC 1951) encoder for the arrays the samplers store.
//
// The reference writes its artefacts with np.savez_compressed (test_step_vae_sampling.py:785-791, 2039-2045): zip
// members holding raw deflate streams.  The big member is `imputations` [T,B,K,D] float32, 1-2 GB per batch at the
// published settings, in which the observed half of every row repeats the row before it (the same example's next
// chain, or its next iteration) and the other half is sampled values whose low bytes do not compress.  A general
// deflate encoder spends its time searching hash chains for matches that can only be found one row back; this one
// looks only there (distance = the row stride) and codes what it finds with a Huffman code fitted to each block of
// 256 KiB (the high bytes of floats are few distinct values), so it runs ten times faster per thread and still
// writes smaller files.  Any inflate implementation reads the result.
//
// A piece is a run of Huffman-coded blocks closed by an empty stored block (what zlib's Z_SYNC_FLUSH emits), so it
// ends on a byte boundary and pieces made by different threads concatenate into one valid stream; matches may
// reach back into the bytes of the preceding piece, which the decoder has in its window.
#include "cs_common.h"

#include <cstring>
#include <vector>

namespace {

struct BitWriter {
  uint8_t* p;
  uint8_t* end;
  uint64_t buf = 0;
  int n = 0;
  bool overflow = false;
  inline void put(uint32_t bits, int len) {  // bits already in stream order (LSB first)
    buf |= (uint64_t)bits << n;
    n += len;
    if (n >= 32) {
      if (end - p < 4) { overflow = true; n -= 32; buf >>= 32; return; }
      const uint32_t w = (uint32_t)buf;
      std::memcpy(p, &w, 4);   // little-endian host
      p += 4;
      buf >>= 32;
      n -= 32;
    }
  }
  inline void align_byte() {
    if (n & 7) put(0, 8 - (n & 7));
  }
  inline void flush_bytes() {  // after align_byte
    while (n >= 8) {
      if (p >= end) { overflow = true; return; }
      *p++ = (uint8_t)buf;
      buf >>= 8;
      n -= 8;
    }
  }
};

inline uint32_t reverse_bits(uint32_t v, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) r |= ((v >> i) & 1u) << (len - 1 - i);
  return r;
}

struct Tables {
  uint16_t lit_code[256];
  uint8_t lit_len[256];
  // match length 3..258 -> code bits (Huffman code of the length symbol followed by its extra bits) and their count
  uint32_t len_code[259];
  uint8_t len_len[259];
  Tables() {
    for (int s = 0; s < 256; ++s) {   // RFC 1951, 3.2.6
      if (s < 144) { lit_code[s] = (uint16_t)reverse_bits(0x30 + s, 8); lit_len[s] = 8; }
      else { lit_code[s] = (uint16_t)reverse_bits(0x190 + (s - 144), 9); lit_len[s] = 9; }
    }
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                 131, 163, 195, 227, 258};
    static const int extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int L = 3; L <= 258; ++L) {
      int s = 28;
      while (base[s] > L) --s;
      if (L == 258) s = 28;
      else if (s == 28) s = 27;   // 227..257 belong to symbol 284; 285 is exactly 258
      const int sym = 257 + s;
      uint32_t code; int len;
      if (sym < 280) { code = reverse_bits(sym - 256, 7); len = 7; }
      else { code = reverse_bits(0xC0 + (sym - 280), 8); len = 8; }
      code |= (uint32_t)(L - base[s]) << len;   // extra bits: value, LSB first
      len_code[L] = code;
      len_len[L] = (uint8_t)(len + extra[s]);
    }
  }
};

const Tables& tables() {
  static const Tables t;
  return t;
}

// distance -> 5-bit code + extra bits
inline void dist_bits(int dist, uint32_t& code, int& len) {
  static const int base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049,
                               3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const int extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  int s = 29;
  while (base[s] > dist) --s;
  code = reverse_bits((uint32_t)s, 5) | ((uint32_t)(dist - base[s]) << 5);
  len = 5 + extra[s];
}

// number of leading bytes on which a[0..max) equals b[0..max)
inline int64_t equal_run(const uint8_t* a, const uint8_t* b, int64_t max) {
  int64_t k = 0;
  while (k + 8 <= max) {
    uint64_t x, y;
    std::memcpy(&x, a + k, 8);
    std::memcpy(&y, b + k, 8);
    const uint64_t d = x ^ y;
    if (d) return k + (__builtin_ctzll(d) >> 3);
    k += 8;
  }
  while (k < max && a[k] == b[k]) ++k;
  return k;
}

// Code lengths (<= max_len) of a Huffman code for freq[0..n): symbols with freq 0 get length 0.  The tree is built
// by repeatedly joining the two lightest nodes; lengths beyond max_len are folded back by moving codes between
// lengths until the Kraft sum is exactly one, the rarest symbols taking the longest codes.
void huffman_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len_out) {
  struct Node { uint64_t w; int left, right; };
  int used[288];
  int nu = 0;
  for (int i = 0; i < n; ++i) {
    len_out[i] = 0;
    if (freq[i]) used[nu++] = i;
  }
  if (nu == 0) return;
  if (nu == 1) { len_out[used[0]] = 1; return; }
  // sort used symbols by (freq, index)
  for (int i = 1; i < nu; ++i) {
    const int v = used[i];
    int j = i - 1;
    while (j >= 0 && (freq[used[j]] > freq[v] || (freq[used[j]] == freq[v] && used[j] > v))) { used[j + 1] = used[j]; --j; }
    used[j + 1] = v;
  }
  Node nodes[2 * 288];
  for (int i = 0; i < nu; ++i) nodes[i] = Node{freq[used[i]], -1, -1};
  // two queues: leaves [0, nu) in order, internal nodes [nu, ...) created in non-decreasing weight
  int leaf = 0, inner = nu, next = nu;
  auto take = [&]() {
    if (leaf < nu && (inner >= next || nodes[leaf].w <= nodes[inner].w)) return leaf++;
    return inner++;
  };
  while ((nu - leaf) + (next - inner) > 1) {
    const int a = take(), b = take();
    nodes[next] = Node{nodes[a].w + nodes[b].w, a, b};
    ++next;
  }
  // depths: the root is the last node; children come before their parents
  int depth[2 * 288];
  depth[next - 1] = 0;
  for (int i = next - 1; i >= nu; --i) {
    depth[nodes[i].left] = depth[i] + 1;
    depth[nodes[i].right] = depth[i] + 1;
  }
  int count[64] = {0};
  for (int i = 0; i < nu; ++i) count[depth[i] > 63 ? 63 : depth[i]]++;
  for (int d = 63; d > max_len; --d) { count[max_len] += count[d]; count[d] = 0; }
  uint64_t total = 0;
  for (int d = 1; d <= max_len; ++d) total += (uint64_t)count[d] << (max_len - d);
  while (total > (1ull << max_len)) {
    count[max_len]--;
    for (int d = max_len - 1; d >= 1; --d)
      if (count[d]) { count[d]--; count[d + 1] += 2; break; }
    total--;
  }
  // rarest symbols first -> longest codes first
  int i = 0;
  for (int d = max_len; d >= 1; --d)
    for (int c = 0; c < count[d]; ++c) len_out[used[i++]] = (uint8_t)d;
}

// canonical codes (RFC 1951, 3.2.2) in stream order (bit-reversed)
void canonical_codes(const uint8_t* len, int n, uint16_t* code) {
  int bl_count[16] = {0};
  for (int i = 0; i < n; ++i) bl_count[len[i]]++;
  bl_count[0] = 0;
  int next_code[16];
  int c = 0;
  for (int b = 1; b <= 15; ++b) {
    c = (c + bl_count[b - 1]) << 1;
    next_code[b] = c;
  }
  for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)reverse_bits((uint32_t)next_code[len[i]]++, len[i]) : 0;
}

constexpr int LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                              131, 163, 195, 227, 258};
constexpr int LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};

struct LenSym {
  uint8_t sym[259];   // match length -> index of its length symbol (0..28)
  LenSym() {
    for (int L = 3; L <= 258; ++L) {
      int s = 28;
      while (LEN_BASE[s] > L) --s;
      sym[L] = (uint8_t)s;
    }
  }
};

// One block with a Huffman code of its own (BTYPE = 10) for the tokens tok[0..nt): values < 256 are literals,
// 256 + (L - 3) a match of length L at the piece's one distance.
void write_dynamic_block(BitWriter& w, const uint16_t* tok, int64_t nt, int dist_sym, uint32_t dist_extra, int dist_extra_len) {
  static const LenSym LS;
  uint32_t freq[286] = {0};
  bool any_match = false;
  for (int64_t i = 0; i < nt; ++i) {
    const int t = tok[i];
    if (t < 256) freq[t]++;
    else { freq[257 + LS.sym[t - 256 + 3]]++; any_match = true; }
  }
  (void)any_match;
  freq[256] = 1;
  uint8_t ll_len[286];
  huffman_lengths(freq, 286, 15, ll_len);
  uint16_t ll_code[286];
  canonical_codes(ll_len, 286, ll_code);
  int hlit = 286;
  while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
  const int hdist = dist_sym + 1;   // the one distance symbol gets a 1-bit code (code 0), all others none
  // the sequence of code lengths, with runs of zeros as symbols 17 (3-10) / 18 (11-138)
  uint8_t seq[286 + 30];
  int ns = 0;
  for (int i = 0; i < hlit; ++i) seq[ns++] = ll_len[i];
  for (int i = 0; i < hdist; ++i) seq[ns++] = (i == dist_sym) ? 1 : 0;
  struct Item { uint8_t sym, extra; };
  Item items[286 + 30];
  int ni = 0;
  for (int i = 0; i < ns;) {
    if (seq[i] == 0) {
      int r = 1;
      while (i + r < ns && seq[i + r] == 0 && r < 138) ++r;
      if (r >= 11) items[ni++] = Item{18, (uint8_t)(r - 11)};
      else if (r >= 3) items[ni++] = Item{17, (uint8_t)(r - 3)};
      else { for (int k = 0; k < r; ++k) items[ni++] = Item{0, 0}; }
      i += r;
    } else {
      items[ni++] = Item{seq[i], 0};
      ++i;
    }
  }
  uint32_t cl_freq[19] = {0};
  for (int i = 0; i < ni; ++i) cl_freq[items[i].sym]++;
  uint8_t cl_len[19];
  huffman_lengths(cl_freq, 19, 7, cl_len);
  uint16_t cl_code[19];
  canonical_codes(cl_len, 19, cl_code);
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cl_len[order[hclen - 1]] == 0) --hclen;
  w.put(0x4, 3);                 // BFINAL = 0, BTYPE = 10: bits 0, 0, 1
  w.put((uint32_t)(hlit - 257), 5);
  w.put((uint32_t)(hdist - 1), 5);
  w.put((uint32_t)(hclen - 4), 4);
  for (int i = 0; i < hclen; ++i) w.put(cl_len[order[i]], 3);
  for (int i = 0; i < ni; ++i) {
    w.put(cl_code[items[i].sym], cl_len[items[i].sym]);
    if (items[i].sym == 17) w.put(items[i].extra, 3);
    else if (items[i].sym == 18) w.put(items[i].extra, 7);
  }
  // match length -> bits of the length symbol followed by its extra bits, then the distance bit and extra bits
  uint32_t m_code[259];
  uint8_t m_len[259];
  for (int L = 3; L <= 258; ++L) {
    const int s = LS.sym[L];
    m_code[L] = (uint32_t)ll_code[257 + s] | ((uint32_t)(L - LEN_BASE[s]) << ll_len[257 + s]);
    m_len[L] = (uint8_t)(ll_len[257 + s] + LEN_EXTRA[s]);
  }
  const uint32_t d_bits = dist_extra << 1;       // 1-bit code 0, then the extra bits
  const int d_len = 1 + dist_extra_len;
  for (int64_t i = 0; i < nt; ++i) {
    const int t = tok[i];
    if (t < 256) {
      w.put(ll_code[t], ll_len[t]);
    } else {
      const int L = t - 256 + 3;
      w.put(m_code[L], m_len[L]);   // <= 15 + 5 bits
      w.put(d_bits, d_len);         // <= 1 + 13 bits
    }
  }
  w.put(ll_code[256], ll_len[256]);
}

}  // namespace

// One piece of a raw deflate stream for src[0..n): literals and matches at distance `stride` only.  `lookback`:
// how many bytes before src belong to the same stream and are addressable at src - lookback (0 at the start of the
// stream's data); matches never reach further back.  Returns the number of bytes written to dst, or -1 when
// dst_cap is too small (n + n/8 + 64 always suffices).
extern "C" int64_t cs_deflate_rows(const uint8_t* src, int64_t n, int32_t stride, int64_t lookback, uint8_t* dst,
                                   int64_t dst_cap) {
  if (!src || !dst || n < 0 || stride < 1 || stride > 32768 || lookback < 0 || dst_cap < 16) {
    cs_fail(CS_EINVAL, "cs_deflate_rows: bad argument");
    return -1;
  }
  const Tables& T = tables();
  BitWriter w{dst, dst + dst_cap};
  uint32_t dcode; int dlen;
  dist_bits(stride, dcode, dlen);
  if (n < 2048) {
    // too short to pay for a code table of its own: the fixed code
    w.put(0x2, 3);   // BFINAL = 0, BTYPE = 01 (fixed Huffman): bits 0, 1, 0 in stream order
    int64_t i = 0;
    while (i < n) {
      int64_t run = 0;
      if (i + lookback >= stride) {
        const int64_t max = (n - i) < 258 ? (n - i) : 258;
        run = equal_run(src + i, src + i - stride, max);
      }
      if (run >= 3) {
        w.put(T.len_code[run], T.len_len[run]);
        w.put(dcode, dlen);
        i += run;
      } else {
        w.put(T.lit_code[src[i]], T.lit_len[src[i]]);
        ++i;
      }
    }
    w.put(0, 7);     // end of block (symbol 256)
  } else {
    static const int dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                  2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    int dsym = 29;
    while (dbase[dsym] > stride) --dsym;
    const uint32_t dextra = (uint32_t)(stride - dbase[dsym]);
    const int dextra_len = dlen - 5;
    constexpr int64_t BLOCK = 256 << 10;
    std::vector<uint16_t> tok((size_t)(n < BLOCK ? n : BLOCK));
    for (int64_t b0 = 0; b0 < n; b0 += BLOCK) {
      const int64_t b1 = (n - b0) < BLOCK ? n : b0 + BLOCK;
      int64_t nt = 0, i = b0;
      while (i < b1) {
        int64_t run = 0;
        if (i + lookback >= stride) {
          const int64_t max = (b1 - i) < 258 ? (b1 - i) : 258;
          run = equal_run(src + i, src + i - stride, max);
        }
        if (run >= 3) {
          tok[nt++] = (uint16_t)(256 + run - 3);
          i += run;
        } else {
          tok[nt++] = src[i];
          ++i;
        }
      }
      write_dynamic_block(w, tok.data(), nt, dsym, dextra, dextra_len);
    }
  }
  w.put(0, 3);     // BFINAL = 0, BTYPE = 00: an empty stored block ...
  w.align_byte();
  w.flush_bytes();
  w.put(0xFFFF0000u, 32);   // ... LEN = 0, NLEN = 0xFFFF
  w.flush_bytes();
  if (w.overflow) {
    cs_fail(CS_EINVAL, "cs_deflate_rows: output buffer too small");
    return -1;
  }
  return (int64_t)(w.p - dst);
}

// ----------------------------------------------------------------------------------------
// CRC-32 (the zip container's checksum of a member's uncompressed bytes; polynomial 0xEDB88320, as zlib's crc32) of
// a piece, and the combination of the checksums of two adjacent pieces: the writer's threads checksum the pieces
// they deflate, and the member's checksum is folded together from theirs instead of one thread reading the whole
// array again.
// ----------------------------------------------------------------------------------------
namespace {
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 0xFF] ^ (t[k - 1][i] >> 8);
  }
};
// a 32 x 32 matrix over GF(2) (one column per word) times a vector
inline uint32_t gf2_times(const uint32_t* mat, uint32_t vec) {
  uint32_t sum = 0;
  for (; vec; vec >>= 1, ++mat)
    if (vec & 1) sum ^= *mat;
  return sum;
}
inline void gf2_square(uint32_t* square, const uint32_t* mat) {
  for (int n = 0; n < 32; ++n) square[n] = gf2_times(mat, mat[n]);
}
}  // namespace

extern "C" uint32_t cs_crc32(const uint8_t* p, int64_t n, uint32_t crc) {
  static const CrcTables T;
  uint32_t c = ~crc;
  while (n > 0 && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {   // eight bytes per step (little-endian host)
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
        T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

// crc of A followed by B from crc(A), crc(B) and the length of B: appending len2 zero bytes to A is a linear map of
// its checksum register, applied by repeated squaring of the one-zero-bit operator.
extern "C" uint32_t cs_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2) {
  if (len2 <= 0) return crc1;
  uint32_t even[32], odd[32];
  odd[0] = 0xEDB88320u;   // the operator of one zero bit
  uint32_t row = 1;
  for (int n = 1; n < 32; ++n) { odd[n] = row; row <<= 1; }
  gf2_square(even, odd);  // two zero bits
  gf2_square(odd, even);  // four
  do {                    // the first squaring below gives the operator of one zero byte
    gf2_square(even, odd);
    if (len2 & 1) crc1 = gf2_times(even, crc1);
    len2 >>= 1;
    if (!len2) break;
    gf2_square(odd, even);
    if (len2 & 1) crc1 = gf2_times(odd, crc1);
    len2 >>= 1;
  } while (len2);
  return crc1 ^ crc2;
}
