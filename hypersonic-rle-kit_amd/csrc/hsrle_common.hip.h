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
  uint32_t k = 0;
  for (; k < n && src + k + 16 <= srcEnd; k += 16)
    st128(dst + k, ld128(src + k));
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
__device__ __forceinline__ u32x4 global_window16(const uint8_t *in, uint64_t blockAt, uint64_t U, uint32_t p)
{
  const int64_t g = (int64_t)blockAt + (int64_t)(int32_t)p;
  if (g >= 0 && (uint64_t)g + 16u <= U)
    return ld128(in + g);
  uint32_t t[4] = { 0, 0, 0, 0 };
  for (uint32_t k = 0; k < 16u; k++)
  {
    const int64_t gk = g + (int64_t)k;
    if (gk >= 0 && (uint64_t)gk < U) t[k >> 2] |= (uint32_t)in[gk] << (8u * (k & 3u));
  }
  return u32x4{ t[0], t[1], t[2], t[3] };
}

} // namespace hsrle
