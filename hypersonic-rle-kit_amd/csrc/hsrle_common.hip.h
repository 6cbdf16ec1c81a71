// hsrle_common.hip.h -- codec traits and per-lane memory helpers shared by the gfx950 kernels.
//
// Execution model of every codec kernel in this library: ONE LANE PER BLOCK.  A block is an independent reference
// stream (its own header and terminator), so a 64-wide wavefront walks 64 packet chains at once; the sequential
// packet-to-packet dependency of the format (reference: src/rleX_extreme_cpu_decode.h:129-162 -- the next header's
// position is known only after the previous packet's literal length) becomes 64-way parallel instead of leaving 63
// lanes idle behind one header walk.  Memory is moved in 16-byte per-lane vectors (unaligned global / LDS accesses are
// native on gfx950), output is staged in LDS rows and flushed as whole 128-byte lines.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsrle {

enum Family : int { PLAIN = 0, PACKED = 1, LUT3 = 2, LUT7 = 3, SINGLE = 4, PACKED_SINGLE = 5,
                    SHORT0 = 6, SHORT1 = 7, SHORT3 = 8, SHORT7 = 9,     // Short family: 0 / 1 / 3 / 7 symbol LUT, one-byte packed headers
                    SHORT_SINGLE = 10 };                                // rle8_single_short: one symbol per stream, none in the packets

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Entry record (hsrle_index.hip.h writes them, k_decode_blocks starts from them): the decoder state at output position b * B of a
// reference stream.  dwords: [0..1] stream position relative to `payload`, [2] literal bytes left, [3] run bytes left, [4] pattern
// phase | flags, [5] bytes from that position to the stream's end, [6..9] current symbol, [10..] move-to-front list (K x SW dwords)
constexpr uint32_t kEntryRecDwords = 24;
constexpr uint32_t REC_LAST = 0x100u, REC_SINGLE = 0x200u;

// Thresholds and field encodings per codec (reference: src/rle8_extreme_cpu.h:5-23,
// src/rleX_extreme_cpu.h:1-16, src/rle24_extreme_cpu.h:10-11, src/rle128_extreme_cpu.h:10-11, src/rleX_Xsl.h:1-17).
template <int FAM, int S, int AL>
struct Traits
{
  static constexpr bool kLut = (FAM == LUT3 || FAM == LUT7);
  static constexpr bool kPacked = (FAM == PACKED);
  static constexpr bool kSingle = (FAM == SINGLE || FAM == PACKED_SINGLE);
  // Short family (reference: src/rleX_Xsl_short.h:1-43): [lut index | count | range] in one byte, or the 3-byte
  // form with a 9 bit count and a SRB bit range; both fields carry the value + 2
  static constexpr bool kShort = (FAM >= SHORT0 && FAM <= SHORT_SINGLE);
  static constexpr bool kShortSingle = (FAM == SHORT_SINGLE);          // header parameters of the 0-symbol codec, thresholds of the LUT ones (rleX_Xsl_short.h:1-11)
  static constexpr int K = (FAM == LUT3 || FAM == SHORT3) ? 3 : ((FAM == LUT7 || FAM == SHORT7) ? 7 : (FAM == SHORT1 ? 1 : 0));
  static constexpr uint32_t SLB = (FAM == SHORT3) ? 2u : (FAM == SHORT1 ? 1u : (FAM == SHORT7 ? 3u : 0u));   // lut index bits
  static constexpr uint32_t SCB = (FAM == SHORT3 || FAM == SHORT1) ? 3u : (FAM == SHORT7 ? 2u : 4u);          // packed count bits
  static constexpr uint32_t SRBP = 8u - SLB - SCB;                                                              // packed range bits
  static constexpr uint32_t SRB = (FAM == SHORT7) ? 24u - SLB - SRBP - 9u : 24u - SLB - SCB - 9u;               // long form range bits (9..11)
  static constexpr uint32_t SCINV = (1u << SCB) - 1u;                 // packed count value that selects the 3-byte form
  static constexpr uint32_t SMAXPR = (1u << SRBP) - 1u, SMAXPC = (1u << SCB) - 2u, SMAXTR = (1u << SRB) - 1u, SMAXTC = 511u;
  static constexpr uint32_t SMINS = (K != 0 || kShortSingle) ? 2u : (uint32_t)S + 2u;                                         // shortest run that can be stored
  static constexpr uint32_t SMINL = kShortSingle ? 11u : ((K != 0) ? (uint32_t)S + 11u : (uint32_t)S + 12u);                         // runs this long are always stored
  static constexpr uint32_t STB = (FAM == SHORT3) ? 4u : (FAM == SHORT7 ? 2u : 8u);                            // second byte of the terminators
  static constexpr int RB = (FAM == LUT3) ? 7 : 6; // LUT range bits
  static constexpr bool kAligned = (S > 1) && (AL != 0);
  // Packed byte-aligned (and 8 bit Packed) use the 7-bit-or-4-byte range field; sym-aligned Packed is the hybrid (A.5 q10).
  static constexpr bool kRange7 = kPacked && !kAligned;
  static constexpr uint32_t SHORT = kLut ? 3u : (S == 1 ? (kPacked ? 3u : 6u) : (kPacked ? 3u : (uint32_t)S + 4u));
  static constexpr uint32_t MEDIUM = (S == 1) ? 4u : (uint32_t)S + 3u;                                  // Packed only
  static constexpr uint32_t LONG = kLut ? (uint32_t)S + 10u
                                        : (S == 1 ? (kPacked ? 11u : 9u) : (kPacked ? (kRange7 ? (uint32_t)S + 11u : (uint32_t)S + 10u) : (uint32_t)S + 11u));
  static constexpr uint32_t MAXRANGE = kLut ? ((1u << RB) - 1u) : (kRange7 ? 127u : 255u);
  static constexpr uint32_t kHeaderSize = (S == 1 && !kLut && !kShort) ? 9u : 8u;
  static constexpr bool kMtf = kLut || (kShort && K > 0);               // keeps a move-to-front list of K symbols
  static constexpr int SW = (S + 3) / 4; // dwords per symbol
};

// ------------------------------------------------------------------------------------------------------------------
// unaligned little-endian accesses (global or LDS; the address space is inferred after inlining)

__device__ __forceinline__ uint32_t ld8(const uint8_t *p) { return *p; }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u32x4 ld128(const uint8_t *p) { u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st8(uint8_t *p, uint32_t v) { *p = (uint8_t)v; }
__device__ __forceinline__ void st16(uint8_t *p, uint32_t v) { uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }
__device__ __forceinline__ void st32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void st128(uint8_t *p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// 16-byte LDS accesses at DWORD aligned addresses.  gfx950 executes ds_read_b128 / ds_write_b128 at any dword aligned address
// at full speed (tools/ubench/lds_widths.hip), but the compiler only emits them when it believes the address is 16-byte
// aligned (otherwise it splits into ds_read2_b32 pairs); the alignment claim below is what selects the single instruction.
__device__ __forceinline__ u32x4 lds_ld128(const uint8_t *p) { return *(const u32x4 *)__builtin_assume_aligned(p, 16); }
__device__ __forceinline__ void lds_st128(uint8_t *p, u32x4 v) { *(u32x4 *)__builtin_assume_aligned(p, 16) = v; }
__device__ __forceinline__ uint32_t lds_ld32(const uint8_t *p) { return *(const uint32_t *)__builtin_assume_aligned(p, 4); }

// bytes [1..4] of the 8-byte value hi:lo, i.e. (hi:lo) >> (8*n), n in 0..3 (v_alignbyte_b32)
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }

// exact per-byte zero detector: 0x80 in every byte of z that is zero
__device__ __forceinline__ uint32_t zero_bytes(uint32_t z)
{
  const uint32_t t = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
  return ~t & 0x80808080u;
}

// movemask of the zero bytes of a 16-byte value (bit k: byte k of t0..t3 is zero): the exact SWAR test leaves 0x80 in every NONZERO byte,
// v_dot4_u32_u8 with the weights 1 2 4 8 / 16 32 64 128 adds them up to 128 * (8 bit mask) per dword pair -- 6 VALU per dword where
// shift + multiply + shift + mask per dword took 11 (round 3; the form of cmpeq + movemask in every run detector here)
__device__ __forceinline__ uint32_t zero_mask16(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3)
{
  const uint32_t n0 = (((t0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t0) & 0x80808080u, n1 = (((t1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t1) & 0x80808080u;
  const uint32_t n2 = (((t2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t2) & 0x80808080u, n3 = (((t3 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t3) & 0x80808080u;
  const uint32_t a01 = __builtin_amdgcn_udot4(n1, 0x80402010u, __builtin_amdgcn_udot4(n0, 0x08040201u, 0u, false), false);
  const uint32_t a23 = __builtin_amdgcn_udot4(n3, 0x80402010u, __builtin_amdgcn_udot4(n2, 0x08040201u, 0u, false), false);
  return ~(((a23 << 8) | a01) >> 7) & 0xFFFFu;
}

// Zeroing of small tables on a stream by a KERNEL.  hipMemsetAsync is correct in eager mode, but captured into a HIP graph its node was seen to act on
// the first replay only (ROCm 7.2, MI355X: experiments/r04/graph_single_repro2.py -- the second replay of a captured compress found its counters
// not zeroed); a kernel node has no such surprise.  p: 4-byte aligned, bytes: rounded up to whole words.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_zero_words(uint32_t *__restrict__ p, uint64_t words)
{
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256u) p[i] = 0u;
}
// ... two tables in one launch (a launch is ~5 us of a call of 270)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_zero_words2(uint32_t *__restrict__ p, uint64_t words, uint32_t *__restrict__ q, uint64_t qwords)
{
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < words + qwords; i += (uint64_t)gridDim.x * 256u)
    if (i < words) p[i] = 0u; else q[i - words] = 0u;
}
inline hipError_t zero2_async(void *p, uint64_t bytes, void *q, uint64_t qbytes, hipStream_t st)
{
  const uint64_t words = (bytes + 3u) / 4u, qwords = (qbytes + 3u) / 4u;
  const uint64_t wg = (words + qwords + 255u) / 256u;
  if (wg == 0u) return hipSuccess;
  hipLaunchKernelGGL((k_zero_words2<0>), dim3((uint32_t)(wg < 2048u ? wg : 2048u)), dim3(256), 0, st, (uint32_t *)p, words, (uint32_t *)q, qwords);
  return hipGetLastError();
}
inline hipError_t zero_async(void *p, uint64_t bytes, hipStream_t st)
{
  const uint64_t words = (bytes + 3u) / 4u;
  if (words == 0u) return hipSuccess;
  const uint64_t wg = (words + 255u) / 256u;
  hipLaunchKernelGGL((k_zero_words<0>), dim3((uint32_t)(wg < 2048u ? wg : 2048u)), dim3(256), 0, st, (uint32_t *)p, words);
  return hipGetLastError();
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs of the MI355X (workgroup i runs on XCD i % 8), each with
// its own L2 and TLBs.  Mapping workgroup i to tile (i % 8) * (n / 8) + i / 8 gives every XCD one contiguous eighth of the tiles
// instead of every eighth tile, so the waves resident on an XCD work on one compact region of their buffers: 8x fewer pages and
// lines per XCD (measured on the decoder: 3.51 -> 3.43 ms, and its placement-dependent slow state 3.75 -> 3.57 ms).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t workgroup, uint32_t workgroups)
{
  const uint32_t perXcd = workgroups / 8u;
  return (workgroup < perXcd * 8u) ? (workgroup % 8u) * perXcd + workgroup / 8u : workgroup;
}

__device__ __forceinline__ uint32_t first_set_byte(uint32_t m) { return (uint32_t)__builtin_ctz(m) >> 3; }

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Copy n bytes with 16-byte vectors; may write up to 15 bytes past dst + n (the caller guarantees that slack) but never
// reads at or beyond srcEnd.
__device__ __forceinline__ void copy_over(uint8_t *dst, const uint8_t *src, uint32_t n, const uint8_t *srcEnd)
{
  // four loads in flight before the first store: a load-store pair per trip pays one memory latency per 16 bytes
  uint32_t k = 0;
  while (k < n && src + k + 16 <= srcEnd)
  {
    u32x4 v[4];
    bool has[4];
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++)
    {
      has[q] = k + 16u * q < n && src + k + 16u * q + 16 <= srcEnd;
      if (has[q]) v[q] = ld128(src + k + 16u * q);
    }
    uint32_t done = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++)
      if (has[q]) { st128(dst + k + 16u * q, v[q]); done += 16u; }
    k += done;
  }
  for (; k < n; k++)
    dst[k] = src[k];
}

// Copy exactly n bytes (no over-write, no over-read).
__device__ __forceinline__ void copy_exact(uint8_t *dst, const uint8_t *src, uint32_t n)
{
  uint32_t k = 0;
  for (; k + 16 <= n; k += 16)
    st128(dst + k, ld128(src + k));
  if (k + 8 <= n) { st64(dst + k, ld64(src + k)); k += 8; }
  if (k + 4 <= n) { st32(dst + k, ld32(src + k)); k += 4; }
  if (k + 2 <= n) { st16(dst + k, ld16(src + k)); k += 2; }
  if (k < n) dst[k] = src[k];
}

// 16 input bytes at block position p of the block that starts at `blockAt` (p may reach below the block or beyond the input: those
// bytes are never used and read as zero).  Kept out of line on purpose: it is the rare source of the ring encoders' literal copy.
// the 16 bytes at in + g of an input of U bytes, zeros where g + k lies outside [0, U): the END-OF-INPUT path of every 16-byte input read.
// A loop that is NOT unrolled on purpose: unrolled it is ~130 instructions at every one of its ~20 inline sites in the run list encoders
// (20 KB of their 51 KB of code, against an instruction cache of 64 KB), and it runs for the last bytes of the last block only.
__device__ __forceinline__ u32x4 load16_edge(const uint8_t *in, int64_t g, uint64_t U)
{
  uint64_t a = 0, b = 0;
#pragma unroll 1
  for (uint32_t k = 0; k < 16u; k++)
  {
    const int64_t gk = g + (int64_t)k;
    const uint64_t v = (gk >= 0 && (uint64_t)gk < U) ? (uint64_t)in[gk] : 0ull;
    if (k < 8u) a |= v << (8u * k); else b |= v << (8u * (k - 8u));
  }
  return u32x4{ (uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32) };
}

__device__ __forceinline__ u32x4 global_window16(const uint8_t *in, uint64_t blockAt, uint64_t U, uint32_t p)
{
  const int64_t g = (int64_t)blockAt + (int64_t)(int32_t)p;
  if (g >= 0 && (uint64_t)g + 16u <= U)
    return ld128(in + g);
  return load16_edge(in, g, U);
}

// Ring encoders: a literal stretch that has left the ring is copied by the whole wave (hsrle_encode8.hip.h: coop_flush) if it is at least
// this long, else fetched by its lane.  A wave-wide copy costs every lane of the wave a memory latency, so it only pays for stretches
// a lane would need many loads for (with 48, and noting between stored runs too, the 16 - 64 bit codecs lost 30 % on run data).
constexpr uint32_t kNotedLiteralMin = 256;

// position-parallel encoder (hsrle_encode8p.hip.h): largest block; records (one per stored run) a block may leave between its two launches
constexpr uint32_t kPpMaxBlock = 4096u;
constexpr uint32_t kPpRecords = 256u;

constexpr uint32_t kLaneRingStride = 132;  // bytes per lane in the input ring: 128 + one dword, so that the lanes' rows start in different banks

// Per-lane input ring in LDS (128 bytes per lane) with a wave-synchronous top-up.  The lanes of a wave drift apart in their streams,
// so with a per-lane register window some lane waits for global memory in nearly every loop trip, and the wave with it (measured:
// rle8m encode 17 ms per GiB).  Here ALL lanes top their rings up in the same trip (the caller decides when: every so many trips, or
// when any lane runs low), each with up to NQ 16-byte loads that are in flight together: one memory latency per top-up.  After
// topup<NQ>(cursor) the bytes [cursor & ~15, (cursor & ~15) + 16 NQ) are in the ring (those below the limit); NQ <= 8.
struct LaneRing
{
  const uint8_t *base;
  uint32_t limit;                           // readable bytes behind base
  uint8_t *row;
  uint32_t loadedEnd;                       // multiple of 16: everything below is in the ring (or behind the limit)
  template <uint32_t NQ>
  __device__ __forceinline__ void topup(uint32_t cursor)
  {
    static_assert(NQ >= 1u && NQ <= 8u, "the ring holds 8 chunks");
    const uint32_t want = (cursor & ~15u) + 16u * NQ;
    uint64_t ca[NQ], cb[NQ];
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++)
    {
      const uint32_t pos = loadedEnd + 16u * q;
      ca[q] = 0; cb[q] = 0;
      if (pos < want && pos < limit)
      {
        if (pos + 16u <= limit) { ca[q] = ld64(base + pos); cb[q] = ld64(base + pos + 8); }
        else if (limit >= 16u)
        {
          const uint32_t sh = 8u * (pos - (limit - 16u));                 // 8 .. 120: the last, partial chunk is read at limit - 16
          const uint64_t a = ld64(base + limit - 16u), b = ld64(base + limit - 8u);
          ca[q] = (sh < 64u) ? ((a >> sh) | (b << (64u - sh))) : (b >> (sh - 64u));
          cb[q] = (sh < 64u) ? (b >> sh) : 0ull;
        }
        else
          for (uint32_t x = 0; pos + x < limit; x++)
          {
            if (x < 8u) ca[q] |= (uint64_t)base[pos + x] << (8u * x); else cb[q] |= (uint64_t)base[pos + x] << (8u * (x - 8u));
          }
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++)
    {
      const uint32_t pos = loadedEnd + 16u * q;
      if (pos < want && pos < limit)
      {
        uint32_t *const r = reinterpret_cast<uint32_t *>(row + (pos & 127u));
        r[0] = (uint32_t)ca[q]; r[1] = (uint32_t)(ca[q] >> 32); r[2] = (uint32_t)cb[q]; r[3] = (uint32_t)(cb[q] >> 32);
      }
    }
    if (want > loadedEnd) loadedEnd = want;
  }
  __device__ __forceinline__ uint32_t get(uint32_t pos) const { return row[pos & 127u]; }
  // the 8 bytes at pos (any alignment); reads the three dwords around them
  __device__ __forceinline__ void get64(uint32_t pos, uint32_t &v0, uint32_t &v1) const
  {
    const uint32_t *const w = reinterpret_cast<const uint32_t *>(row);
    const uint32_t k = (pos & 127u) >> 2, sh = pos & 3u;
    const uint32_t d0 = w[k], d1 = w[(k + 1u) & 31u], d2 = w[(k + 2u) & 31u];
    v0 = alignbyte(d1, d0, sh); v1 = alignbyte(d2, d1, sh);
  }
  __device__ __forceinline__ uint32_t get32(uint32_t pos) const
  {
    const uint32_t *const w = reinterpret_cast<const uint32_t *>(row);
    const uint32_t k = (pos & 127u) >> 2;
    return alignbyte(w[(k + 1u) & 31u], w[k], pos & 3u);
  }
};

// ---- move-to-front lists as transformers (hsrle_mono_encode.hip.h: what a chunk does to the list in front of it is "these d symbols, most
//      recent first, now lead"): lists are 8 words per chunk, entry k in word k, word 7 = d (results) / "encode me" (guesses) ----
__device__ __forceinline__ uint64_t mono_default_entry(uint32_t k, uint32_t S)
{
  // 0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE in every symbol byte (rleX_Xsl.h: the initial list)
  const uint64_t b = (0xFE807E01FF7F00ull >> (8u * k)) & 0xFFull;
  const uint64_t all = b * 0x0101010101010101ull;
  return (S >= 8u) ? all : (all & ((1ull << (8u * S)) - 1ull));
}

// newest-first accumulation of at most K distinct symbols
struct MonoListAcc
{
  uint64_t e[7];
  uint32_t n;
  __device__ __forceinline__ void add_one(uint64_t v, uint32_t K)
  {
    bool have = false;
#pragma unroll
    for (int k = 0; k < 7; k++) have = have || ((uint32_t)k < n && e[k] == v);
    if (!have && n < K)
    {
#pragma unroll
      for (int k = 0; k < 7; k++) if ((uint32_t)k == n) e[k] = v;
      n++;
    }
  }
  __device__ __forceinline__ void add(const uint64_t *__restrict__ t, uint32_t K)
  {
    const uint32_t d = (uint32_t)t[7];
#pragma unroll
    for (int j = 0; j < 7; j++)
      if ((uint32_t)j < d && n < K) add_one(t[j], K);
  }
};

// a word another lane of this wave may have rewritten since this lane last read its line: past the L1
__device__ __forceinline__ uint64_t ld_fresh64(const uint64_t *p) { return __hip_atomic_load(const_cast<uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

} // namespace hsrle
