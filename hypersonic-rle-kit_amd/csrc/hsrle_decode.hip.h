// hsrle_decode.hip.h -- block-parallel decoder for every rleX_extreme packet grammar.
//
// Replaces the reference's decode bodies:
//   8 bit multi/single      src/rle8_extreme_cpu.h:702-764 (dispatch), :1546-2006, :2008-2434
//   16/32/64 bit            src/rleX_extreme_cpu.h:82-111, src/rleX_extreme_cpu_decode.h:27-164
//   24/48 bit               src/rle{24,48}_extreme_cpu_decode.h
//   128 bit                 src/rle128_extreme_cpu.h:499-802
//   3/7 symbol LUT          src/rleX_Xsl.h:530-1881
//   MEMCPY_* / MEMSET_*     src/rleX_extreme_common.h:32-312  (-> the literal / run parts of the packet loop below)
//
// One lane decodes one block (= one complete reference stream); a 64-lane workgroup (one wavefront) owns 64 consecutive blocks,
// i.e. one contiguous 64 * blockSize slice of the output.  Data path per workgroup and step of Q = T = 128 output bytes per lane:
//
//   HBM --(top-up: 8 adjacent lanes hold the next 8 16-byte chunks of ONE stream in registers; a chunk moves into the ring when it
//          fits and its lane then requests the chunk 128 bytes further on: every chunk is requested exactly once)--> LDS ring [64][R]
//   LDS ring --(per-lane packet walk: header fields, 16-byte literal / run chunks)--> LDS tile [64][T]
//   LDS tile --(flush: 8 adjacent lanes write the 128 contiguous bytes of one row, non-temporal)--> HBM
//
// so the packet-to-packet dependency chain (the next header's position is known only after the previous packet's literal
// length, reference: src/rleX_extreme_cpu_decode.h:129-162) only ever waits on LDS, never on HBM, and it is walked by
// 64 lanes at once.  Measured facts of gfx950 that shape the code (tools/ubench/):
//   * 16-byte loads by 64 lanes at 64 different streams run at ~1 TB/s chip-wide; grouped loads read whole lines; rows must be
//     written as whole aligned 128-byte lines (misaligned rows: 0.45x).
//   * an LDS access that is not NATURALLY aligned (16/8/4-byte access at a multiple of its size) is executed one lane at a time
//     (64 cycles per wave instruction instead of 11-22).  So every LDS access here is naturally aligned and the byte granularity
//     of the format is restored in registers (v_alignbyte / v_bfi / v_cndmask: funnel16, funnel24, merge_low_m).
//   * a wave on its own needs ~13.5 k cycles per 128-byte step (a serial chain of ~1 450 instructions), so throughput scales with
//     the resident waves: ring and tile are unpadded (XOR-swizzled rows), everything else lives in registers: 16.9 KB LDS,
//     <= 168 VGPRs, 9 waves per CU (LDS is allocated in 1 280-byte granules: 10 waves would need <= 15 360 bytes).
//   * under full occupancy the vector-memory instructions of a step stall for ~10 k cycles (alone: 1.5 k): the L1's outstanding
//     requests are the bound (TCP_PENDING_STALL 60 %, ~1 000 cycles per read request), hence the exactly-once top-up.
// Nothing outside [0, uncompressedSize) is written (the reference scribbles up to 128 bytes past the end, A.5 q7).
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_parse.hip.h"

namespace hsrle {

typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));   // `out` is a caller pointer: no alignment promise

enum DecodeError : uint32_t
{
  DEC_ERR_HEADER = 1u,   // block stream header does not match the container table
  DEC_ERR_STREAM = 2u,   // packet chain leaves the stream / ends early
  DEC_ERR_MODE = 4u      // unknown 8 bit mode byte (reference: rle8_extreme_cpu.h:759-760)
};

// Per-lane fill pattern of a run, in registers (S > 1).  `v` holds the symbol in its low S bytes.
//   S in {2,4,8,16}: the symbol replicated to 16 bytes (the pattern has period 16)
//   S in {3,6}:      three dwords d0 d1 d2 = one 12-byte period (lcm(S, 4) == 12); w is unused
template <int S>
__device__ __forceinline__ u32x4 make_pattern(u32x4 v)
{
  if constexpr (S == 2) { const uint32_t h = v.x & 0xFFFFu; const uint32_t d = h | (h << 16); return u32x4{ d, d, d, d }; }
  else if constexpr (S == 4) return u32x4{ v.x, v.x, v.x, v.x };
  else if constexpr (S == 8) return u32x4{ v.x, v.y, v.x, v.y };
  else if constexpr (S == 16) return v;
  else if constexpr (S == 3)
  {
    const uint32_t t = v.x & 0xFFFFFFu;
    return u32x4{ t | (t << 24), (t >> 8) | (t << 16), (t >> 16) | (t << 8), 0 };
  }
  else // S == 6
  {
    const uint32_t lo = v.x, hi = v.y & 0xFFFFu;
    return u32x4{ lo, hi | (lo << 16), (lo >> 16) | (hi << 16), 0 };
  }
}

template <int S>
__device__ __forceinline__ u32x4 mask_symbol(u32x4 v)
{
  if constexpr (S == 1) return u32x4{ v.x & 0xFFu, 0, 0, 0 };
  else if constexpr (S == 2) return u32x4{ v.x & 0xFFFFu, 0, 0, 0 };
  else if constexpr (S == 3) return u32x4{ v.x & 0xFFFFFFu, 0, 0, 0 };
  else if constexpr (S == 4) return u32x4{ v.x, 0, 0, 0 };
  else if constexpr (S == 6) return u32x4{ v.x, v.y & 0xFFFFu, 0, 0 };
  else if constexpr (S == 8) return u32x4{ v.x, v.y, 0, 0 };
  else return v;
}

// 16 bytes starting at byte offset `sh` (0..15) of the 32-byte value [x | y]: dword select + v_alignbyte
__device__ __forceinline__ u32x4 funnel16(u32x4 x, u32x4 y, uint32_t sh)
{
  const bool d2 = (sh & 8u) != 0u, d1 = (sh & 4u) != 0u;
  const uint32_t t0 = d2 ? x.z : x.x, t1 = d2 ? x.w : x.y, t2 = d2 ? y.x : x.z, t3 = d2 ? y.y : x.w, t4 = d2 ? y.z : y.x, t5 = d2 ? y.w : y.y;
  const uint32_t z0 = d1 ? t1 : t0, z1 = d1 ? t2 : t1, z2 = d1 ? t3 : t2, z3 = d1 ? t4 : t3, z4 = d1 ? t5 : t4;
  const uint32_t b = sh & 3u;
  return u32x4{ alignbyte(z1, z0, b), alignbyte(z2, z1, b), alignbyte(z3, z2, b), alignbyte(z4, z3, b) };
}

// 16 bytes starting at byte `p` of an LDS buffer whose base is 16-byte aligned, using two NATURALLY ALIGNED 16-byte reads
// (gfx950 executes an LDS access that is not naturally aligned one lane at a time: 64 cycles instead of ~15, measured in
// tools/ubench/lds_align.hip) and a byte funnel in registers.
__device__ __forceinline__ u32x4 lds_read16(const uint8_t *base, uint32_t p)
{
  const uint8_t *src = base + (p & ~15u);
  return funnel16(lds_ld128(src), lds_ld128(src + 16), p & 15u);
}

// 16 bytes of the periodic run pattern starting at pattern phase ph (0 <= ph < S), S in {2,4,8,16}
__device__ __forceinline__ u32x4 pattern_chunk16(u32x4 pv, uint32_t ph) { return funnel16(pv, pv, ph); }
// the same with the period known: 2 and 4 byte symbols are one dword rotated (1 instruction instead of the 14 of the general funnel),
// 8 byte symbols a dword pair
template <int S>
__device__ __forceinline__ u32x4 pattern_chunk16_s(u32x4 pv, uint32_t ph)
{
  if constexpr (S == 2 || S == 4)
  {
    const uint32_t d = alignbyte(pv.x, pv.x, ph);
    return u32x4{ d, d, d, d };
  }
  else if constexpr (S == 8)
  {
    const bool sw = (ph & 4u) != 0u;
    const uint32_t a = sw ? pv.y : pv.x, b = sw ? pv.x : pv.y;
    const uint32_t lo = alignbyte(b, a, ph & 3u), hi = alignbyte(a, b, ph & 3u);
    return u32x4{ lo, hi, lo, hi };
  }
  else
    return funnel16(pv, pv, ph);
}

// S in {3,6}: the three distinct dwords e0 e1 e2 of the pattern stream that starts at phase ph; 16-byte chunk number k of
// that stream is { e[k%3], e[(k+1)%3], e[(k+2)%3], e[k%3] } (16 bytes ahead == one dword further in the 12-byte period)
__device__ __forceinline__ void pattern_dwords12(u32x4 pv, uint32_t ph, uint32_t &e0, uint32_t &e1, uint32_t &e2)
{
  const bool j = ph >= 4u;                                           // ph <= 5
  const uint32_t r0 = j ? pv.y : pv.x, r1 = j ? pv.z : pv.y, r2 = j ? pv.x : pv.z;
  const uint32_t b = ph & 3u;
  e0 = alignbyte(r1, r0, b);
  e1 = alignbyte(r2, r1, b);
  e2 = alignbyte(r0, r2, b);
}

// Same result from three NATURALLY ALIGNED 8-byte reads (24-byte window at p & ~7): one dword-select stage instead of two.
// `base` is 16-byte aligned and readable up to (p & ~7) + 24.
__device__ __forceinline__ uint64_t lds_ld64(const uint8_t *p) { return *(const uint64_t *)__builtin_assume_aligned(p, 8); }

__device__ __forceinline__ u32x4 funnel24(uint64_t a, uint64_t b, uint64_t c, uint32_t sh)
{
  const bool d1 = (sh & 4u) != 0u;
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32), c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
  const uint32_t z0 = d1 ? a1 : a0, z1 = d1 ? b0 : a1, z2 = d1 ? b1 : b0, z3 = d1 ? c0 : b1, z4 = d1 ? c1 : c0;
  const uint32_t n = sh & 3u;
  return u32x4{ alignbyte(z1, z0, n), alignbyte(z2, z1, n), alignbyte(z3, z2, n), alignbyte(z4, z3, n) };
}

__device__ __forceinline__ u32x4 lds_read16_w8(const uint8_t *base, uint32_t p)
{
  const uint8_t *src = base + (p & ~7u);
  return funnel24(lds_ld64(src), lds_ld64(src + 8), lds_ld64(src + 16), p & 7u);
}

// low c bytes (c in 0..15) from `keep`, the rest from `fresh` (the encoders' accumulators; the decoder takes its mask from a LUT)
__device__ __forceinline__ u32x4 merge_low(u32x4 keep, u32x4 fresh, uint32_t c)
{
  // one 64-bit mask of the low (c & 7) bytes; it is the low half's mask for c < 8 and the high half's for c >= 8
  const uint64_t part = ~(~0ull << (8u * (c & 7u)));
  const bool hiHalf = c >= 8u;
  const uint32_t p0 = (uint32_t)part, p1 = (uint32_t)(part >> 32);
  const uint32_t m0 = hiHalf ? ~0u : p0, m1 = hiHalf ? ~0u : p1, m2 = hiHalf ? p0 : 0u, m3 = hiHalf ? p1 : 0u;
  return u32x4{ (keep.x & m0) | (fresh.x & ~m0), (keep.y & m1) | (fresh.y & ~m1), (keep.z & m2) | (fresh.z & ~m2), (keep.w & m3) | (fresh.w & ~m3) };
}

// the same with the mask given (bytes set in m come from `keep`)
__device__ __forceinline__ u32x4 merge_low_m(u32x4 keep, u32x4 fresh, u32x4 m)
{
  return u32x4{ (keep.x & m.x) | (fresh.x & ~m.x), (keep.y & m.y) | (fresh.y & ~m.y), (keep.z & m.z) | (fresh.z & ~m.z), (keep.w & m.w) | (fresh.w & ~m.w) };
}

// 32 bits at byte offset pos (0..12) of the 16-byte little-endian value hi:lo
__device__ __forceinline__ uint32_t ex32(uint64_t lo, uint64_t hi, uint32_t pos)
{
  const uint32_t sh = pos * 8u;
  const uint64_t a = (lo >> (sh & 63u)) | ((hi << 1) << (63u - (sh & 63u)));
  const uint64_t b = hi >> (sh & 63u);
  return (uint32_t)(sh < 64u ? a : b);
}

// constant-position fields of a 16-byte header window (the common, short packet forms sit at compile-time positions: no second ring read,
// no variable funnel): the 32 bits at byte P (P <= 12), the S-byte symbol at byte P (P + S <= 16; dwords beyond the symbol zero)
template <int P>
__device__ __forceinline__ uint32_t win_u32(u32x4 v)
{
  static_assert(P >= 0 && P <= 12, "inside the window");
  if constexpr ((P & 3) == 0) return v[P >> 2];
  else return alignbyte(v[(P >> 2) + 1], v[P >> 2], (uint32_t)(P & 3));
}
template <int S, int P>
__device__ __forceinline__ u32x4 win_symbol(u32x4 v)
{
  static_assert(S <= 8 && P + S <= 16 && P + (S > 4 ? 8 : 4) <= 16, "the symbol lies inside the window");
  u32x4 r = u32x4{ win_u32<P>(v), 0, 0, 0 };
  if constexpr (S > 4) r.y = win_u32<P + 4>(v);
  return mask_symbol<S>(r);
}

// The workgroup IS one wavefront (64 threads): LDS instructions of one wave execute in order, so lanes see each other's
// LDS writes without an s_barrier and without the `s_waitcnt vmcnt(0)` that __syncthreads() implies (which would stall
// every round until the round's global stores have completed).  This only stops the compiler from moving LDS accesses
// across the point.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// (mask & a) | (~mask & b)  -> v_bfi_b32
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }

// FAM in {PLAIN, PACKED, LUT3, LUT7}; the 8 bit PLAIN / PACKED kernels also decode the Single modes (mode byte 1),
// exactly like rle8_decompress / rle8_packed_decompress do.
//   T = output bytes per lane and flush (LDS tile row), Q = output bytes per lane and decode/top-up step (T % Q == 0),
//   R = per-lane stream ring size in LDS (power of two).
//   SGL = false compiles the Single mode out of the 8 bit PLAIN / PACKED kernels (-2 % kernel time: the mode selects fold away): the
//   launchers use it for the codec ids whose encoders only write mode 0 (rle8_multi, rle8_packed_multi); a mode-1 block in such a
//   container is reported as DEC_ERR_MODE.  The Single codec ids and the host-pointer drop-in functions (which look at the mode
//   byte first, like rle8_decompress does) use the general kernel.
template <int FAM, int S, int AL, int T, int R, int Q = T, bool SGL = true, bool ENT = true>
__global__ __launch_bounds__(64) void k_decode_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets,
                                                      const uint8_t *__restrict__ payloadEnd, uint8_t *__restrict__ out, uint64_t U,
                                                      uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ entries, uint32_t entryBase, const uint32_t *__restrict__ gate)
{
  // gate != nullptr (monolithic streams, the decode enqueued behind the index passes without the host in between): the resolve pass left regions to repair or
  // found the chain malformed -- the entry records were not written, so there is nothing to decode: dOut and the status word stay untouched
  if (gate != nullptr && (gate[0] | gate[1]) != 0u) return;
  // entries != nullptr: lane b starts from entry record (b - entryBase) -- the decoder state at output position b * B of a stream that
  // lies somewhere in `payload` (ONE monolithic reference stream, or sub-block b of a container block; hsrle_index.hip.h writes the
  // records) -- instead of from the header of block stream b; everything behind the prologue is the same: a lane still produces the
  // B output bytes [b * B, (b + 1) * B)
#ifdef HSRLE_NO_ENTRIES  // A/B builds only: the kernel without the entry-record prologue
  entries = nullptr;
#endif
  // ENT = false: the instantiation without the entry-record prologue.  For 8 .. 64 bit symbols the prologue is free (same-box A/B); with
  // 16-byte symbols it costs 10 % (8 GiB rle128_sym: 3 120 against 3 460 GiB/s run-distributed, 2 011 against 2 260 video-shaped), so the
  // 128 bit codecs launch this one for plain containers (inst_w128.hip).
  if constexpr (!ENT) entries = nullptr;
  using TR = Traits<FAM, S, AL>;
#ifndef HSRLE_DEC_STARVED_MIN
#define HSRLE_DEC_STARVED_MIN 32u        // capped rounds: lanes starved of literal bytes that make the round go on without a flush (below)
#endif
#ifdef HSRLE_DEC_TRIPS   // A/B builds: one cap for every instantiation
  constexpr uint32_t CAP = HSRLE_DEC_TRIPS;
#else
  // packets per lane and round, 0 = a round lasts until every lane has its T bytes (see HS_ROWPOS below)
  constexpr uint32_t CAP = (S >= 2 && S <= 8) ? 4u : ((S == 1 && (FAM == SHORT3 || FAM == SHORT7)) ? 8u : 0u);
#endif
  constexpr int TS = T;                      // tile row stride: no pad -- the 16-byte chunks of a row are XOR-swizzled by the row index instead (TSW)
  constexpr int RS = R;                      // ring row stride: no pad, no mirror -- chunks are XOR-swizzled by the row index (rowx), every 8-byte piece is addressed on its own
  constexpr int CPR = T / 16;                // 16-byte chunks per tile row == lanes that serve one row in top-up / flush
  constexpr int RPI = 64 / CPR;              // rows covered by one flush instruction
#ifndef HSRLE_FLUSH_GROUP
#define HSRLE_FLUSH_GROUP 4
#endif
  constexpr int FH = CPR < HSRLE_FLUSH_GROUP ? CPR : HSRLE_FLUSH_GROUP; // flush instructions in flight together
  constexpr int LPR = Q / 16;                // lanes that serve one row in a top-up (Q contiguous stream bytes)
  constexpr int RPL = 64 / LPR;              // rows covered by one top-up instruction
  constexpr uint32_t RMASK = (uint32_t)R - 1u;
  constexpr uint32_t MAXHDR = 1u + 4u + (uint32_t)S + 4u + 2u; // longest packet header of any family (+ slack)
  static_assert((R & (R - 1)) == 0 && R >= 64 && (R % Q == 0 || Q % R == 0) && T % Q == 0 && (Q == 32 || Q == 64 || Q == 128), "ring size must be a power of two and a multiple (or a divisor: streams that shrink a lot) of Q");
  static_assert(T == 64 || T == 128, "tile rows are flushed as whole 64/128-byte pieces");
  static_assert(TS % 16 == 0 && RS % 16 == 0, "rows are 16-byte aligned");

  __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TS];
  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * RS];
  __shared__ __attribute__((aligned(16))) uint32_t rinfo[64];           // per-row scalars for the serving lanes (publish())
  // mlut[c] = 16-byte mask with the low c bytes set: the merge of a partial chunk takes its mask from here (one ds_read_b128; the 16
  // entries cover all 64 banks exactly once, so lanes either share an address or hit different banks) instead of ~9 VALU
  __shared__ __attribute__((aligned(16))) uint8_t mlut[16 * 16];
  if (threadIdx.x < 16u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (uint32_t)part, p1 = (uint32_t)(part >> 32);
    lds_st128(mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }
#define HS_MERGE_LOW(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))
#ifdef HSRLE_LDS_BALLAST  // occupancy experiment only: extra LDS so that fewer waves fit on a CU
  __shared__ uint8_t ballast[HSRLE_LDS_BALLAST];
  if (U == 0x7FFFFFFFFFFFFFFFull) ballast[threadIdx.x] = 1;
#endif

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = firstBlock + xcd_tile(blockIdx.x, gridDim.x) * 64u;   // XCD-aware tile order (hsrle_common.hip.h)
  const uint32_t lastBlockExcl = firstBlock + blockCount;
  const uint32_t b = wgFirst + lane;
  const bool active = b < lastBlockExcl;

  uint8_t *const row = tile + lane * TS;
  // Tile swizzle: chunk k of row r lives at chunk slot k ^ f(r), so that 8 neighbouring rows touch 8 different bank groups
  // when they access the same chunk index (what the pad did before, without its 1 KB per wave).  T = 128: f = r & 7;
  // T = 64: f = (r >> 1) & 3 (two rows share a 128-byte bank line).
  auto tsw_of = [](uint32_t r) -> uint32_t { return (T == 128 ? (r & 7u) : ((r >> 1) & 3u)) << 4; };
  const uint32_t tsw = tsw_of(lane);
  // Ring swizzle, same idea as the tile's: ring byte x (0 <= x < R) of row r lives at ring[(r * R) ^ rsw_of(r) ^ x]; the XOR only
  // touches the chunk-index bits, so bytes inside a 16-byte chunk (and 8-byte halves) stay in place.
  auto rsw_of = [](uint32_t r) -> uint32_t { return (R >= 128 ? (r & 7u) : ((r >> 1) & 3u)) << 4; };
  const uint32_t rowx = (lane * (uint32_t)RS) ^ rsw_of(lane);
  // 16 stream bytes at virtual position p: three naturally aligned 8-byte reads (each wraps / swizzles on its own) + byte funnel
  auto ring_win16 = [&](uint32_t p) -> u32x4 {
    const uint32_t a = p & ~7u;
    return funnel24(lds_ld64(ring + (rowx ^ (a & RMASK))), lds_ld64(ring + (rowx ^ ((a + 8u) & RMASK))), lds_ld64(ring + (rowx ^ ((a + 16u) & RMASK))), p & 7u);
  };

  // 24 stream bytes at virtual position p as six dwords: ONE round trip of four aligned 8-byte reads + a byte funnel (the packet headers of
  // the 2 .. 8 byte symbols: every field then sits at a compile-time position, hsrle_parse.hip.h)
  auto ring_win24 = [&](uint32_t p, uint32_t (&d)[6]) {
    const uint32_t a = p & ~7u;
    const uint64_t w0 = lds_ld64(ring + (rowx ^ (a & RMASK))), w1 = lds_ld64(ring + (rowx ^ ((a + 8u) & RMASK)));
    const uint64_t w2 = lds_ld64(ring + (rowx ^ ((a + 16u) & RMASK))), w3 = lds_ld64(ring + (rowx ^ ((a + 24u) & RMASK)));
    const bool d1 = (p & 4u) != 0u;
    const uint32_t z0 = (uint32_t)w0, z1 = (uint32_t)(w0 >> 32), z2 = (uint32_t)w1, z3 = (uint32_t)(w1 >> 32), z4 = (uint32_t)w2, z5 = (uint32_t)(w2 >> 32), z6 = (uint32_t)w3, z7 = (uint32_t)(w3 >> 32);
    const uint32_t y0 = d1 ? z1 : z0, y1 = d1 ? z2 : z1, y2 = d1 ? z3 : z2, y3 = d1 ? z4 : z3, y4 = d1 ? z5 : z4, y5 = d1 ? z6 : z5, y6 = d1 ? z7 : z6;
    const uint32_t nb = p & 3u;
    d[0] = alignbyte(y1, y0, nb); d[1] = alignbyte(y2, y1, nb); d[2] = alignbyte(y3, y2, nb); d[3] = alignbyte(y4, y3, nb); d[4] = alignbyte(y5, y4, nb); d[5] = alignbyte(y6, y5, nb);
  };

  // ---- per-lane stream state ----
  uint32_t slen = 0, blen = 0;
  uint32_t sp = 0;        // read position in the stream
  uint32_t E = 0;         // stream bytes [.., E) are in the ring (multiple of 16)
  uint32_t lim = 0;       // E never exceeds lim (loadable bytes of this stream incl. the payload tail pad)
  uint32_t lit = 0;       // literal bytes of the current packet still to copy
  uint32_t run = 0;       // run bytes of the current packet still to write
  uint32_t phase = 0;     // S > 1: pattern phase of the next run byte
  uint32_t o = 0;         // bytes of this block produced so far
  uint32_t sym4 = 0;      // S == 1: current symbol, byte-broadcast
  [[maybe_unused]] u32x4 patv = u32x4{ 0, 0, 0, 0 }; // S > 1: run pattern (make_pattern)
  u32x4 acc = u32x4{ 0, 0, 0, 0 }; // the 16-byte chunk of the tile row that contains the write position (its low (q & 15) bytes are valid)
  bool last = false;      // the stream ends after the current packet's literals
  bool done = true;
  [[maybe_unused]] bool singleVar = false;
#define single (SGL && singleVar)
  uint32_t err = 0;
  [[maybe_unused]] uint32_t lut[TR::kMtf ? TR::K : 1][TR::SW];

  // Stream positions (sp, E, slen, lim) are VIRTUAL: position 0 is the Q-byte aligned global address at or below the
  // stream start, the stream itself begins at virtual position g0.  Top-up loads then fetch Q-byte aligned pieces, so
  // every 64/128-byte sector of the container is requested exactly once by this workgroup.
  uint32_t g0 = 0;
  uint64_t myBase0 = 0;                                             // this row's (virtual) stream start, relative to `payload`
  {
    uint64_t base0 = 0;
    if (active)
    {
      uint64_t off0, off1;
      if (entries != nullptr)
      {
        const uint32_t *const er = entries + (uint64_t)(b - entryBase) * kEntryRecDwords;
        off0 = (uint64_t)er[0] | ((uint64_t)er[1] << 32);
        off1 = off0 + er[5];
      }
      else { off0 = offsets[b]; off1 = offsets[b + 1]; }
      // the offset table is data too: an entry outside the payload (or a negative / oversized stream length) must end as an error
      // bit, never as a wild read -- the lane then sees an empty stream at offset 0, which fails the header check
      const uint64_t payloadBytes = (uint64_t)(payloadEnd - payload);
      if (off0 > off1 || off1 > payloadBytes || off1 - off0 > 0xFFFFFF00ull) { off0 = 0; off1 = 0; }   // -> DEC_ERR_HEADER below
      // Round 4: position 0 is EXACTLY the Q-byte (128-byte) aligned address below the stream start -- until round 3 it was that address plus
      // (start & 15), which put every window of 8 chunks across two memory lines: the serving lanes' loads touched two lines per row and
      // instruction.  The ring starts at floor16(g0) (E0 below; the first fill reaches over one line boundary, once), so nothing in front
      // of the stream's first chunk is ever requested and g0 needs neither be a multiple of 16 nor lie below R.
      g0 = (uint32_t)((uintptr_t)(payload + off0) & (uintptr_t)(Q - 1));
      base0 = off0 - g0;                                               // may be "negative" for the first blocks: wraps, added to `payload` again below
      slen = (uint32_t)(off1 - off0) + g0;
      const uint64_t start = (uint64_t)b * B;
      blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
      const uint64_t room = (uint64_t)(payloadEnd - payload) - base0;
      lim = (uint32_t)(room > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : room) & ~15u;
      lim = umin(lim, (slen + 15u) & ~15u);                            // nothing behind the stream's last chunk is ever requested
      E = g0 & ~15u;                                                   // the ring's first fill starts at the stream's first chunk
      sp = g0;
      done = false;
    }
    myBase0 = base0;
  }

  auto set_sym = [&](u32x4 v) {
    if constexpr (S == 1) sym4 = (v.x & 0xFFu) * 0x01010101u;
    else patv = make_pattern<S>(v);
  };

  // ---- ring top-up.  issue(): LPR loads; in load q, lanes LPR*g .. LPR*g+LPR-1 read Q contiguous bytes of row RPL*q+g ----
  u32x4 pf[LPR];
  uint32_t pfPos[LPR];                                              // stream position of the chunk held in / on its way to pf[q] (~0: none)
#pragma unroll
  for (int q = 0; q < LPR; q++) { pf[q] = u32x4{ 0, 0, 0, 0 }; pfPos[q] = 0xFFFFFFFFu; }
  uint64_t myBase[LPR];                                             // stream starts of the LPR rows this lane helps to load
  // whether a stream has a chunk left for a lane: from the row's lim kept in a register per served row, or -- where registers are
  // what stands between 8 and 9 waves per CU (the LUT decoders of the wide symbols) -- from a 4-bit count the owner packs below E
  constexpr bool kPackLim = TR::kMtf && TR::K > 1 && S > 1;
  [[maybe_unused]] uint32_t limq[LPR];                              // lim of those rows

  // Per-row scalars travel between the row's owner lane and the lanes that serve the row through a 64-dword LDS array:
  // the owner of row r writes slot (r % G) * P + r / G (P = lanes per row, G = 64 / P rows per instruction), so the P rows a
  // lane serves (rows q * G + lane / P, q = 0 .. P-1) are P consecutive dwords: one or two broadcast ds_read_b128 instead of
  // P (or 2 P) ds_bpermute.
  auto publish = [&](uint32_t v, int P) { rinfo[(lane % (64u / (uint32_t)P)) * (uint32_t)P + lane / (64u / (uint32_t)P)] = v; };
#define HS_EXCHANGE(dst, v, P) { publish(v, P); wave_sync(); _Pragma("unroll") for (int q_ = 0; q_ < (P); q_++) dst[q_] = rinfo[(lane / (P)) * (P) + q_]; }

#ifdef HSRLE_STAMPS
  unsigned long long tx0 = 0, tx1 = 0, tx2 = 0, tx3 = 0, tx4 = 0, tq0, tq1;
#define HS_XSTAMP(acc) { tq1 = __builtin_readcyclecounter(); acc += tq1 - tq0; tq0 = tq1; }
#else
#define HS_XSTAMP(acc)
#endif
  // topup(): one exchange per step.  The LPR lanes that serve a row keep the next LPR chunks behind E in flight / in registers,
  // lane c the chunk whose index is c modulo LPR.  The row owner accounts for what LANDS now that this step's decode has freed
  // ring space (the ring may hold [floor16(sp), floor16(sp) + R); whether a chunk fits is decided when it lands, so the ring only
  // has to hold one step of consumption, not two) and publishes the new E; a serving lane whose chunk lies below it moves the
  // chunk into the ring and requests its next one, Q bytes further on; a chunk that does not fit yet simply stays in its register.
  // Every chunk of the container is requested exactly once -- no re-requests, no dummy loads of predicated-off lanes: the L1's
  // outstanding-request slots are what bounds this kernel under full occupancy (TCP_PENDING_STALL 60 %), and the
  // previous policy (request Q bytes behind E every step, drop what does not fit) spent 1.8x the compressed bytes on them.
#ifdef HSRLE_REQ_STATS  // diagnostic build (tools/fetch_calib.sh): how the top-up's load instructions cover the 128-byte lines
  unsigned long long nFullLine = 0, nHalfLine = 0, nBothHalves = 0, nOneHalf = 0;
  // one row's LPR serving lanes sit side by side: a line is asked for as a whole iff all of them load in the same instruction
  auto count_requests = [&](bool loads) {
    const unsigned long long m = __ballot(loads);
    for (uint32_t grp = 0; grp < 64u / LPR; grp++)
    {
      const uint32_t bits = (uint32_t)(m >> (grp * LPR)) & ((1u << LPR) - 1u);
      if (bits == (1u << LPR) - 1u && LPR == 8) nFullLine++;
      else
      {
        if (bits & 0x0Fu) nHalfLine++;
        if (bits & 0xF0u) nHalfLine++;
        if ((bits & 0x0Fu) && (bits & 0xF0u)) nBothHalves++; else if (bits) nOneHalf++;
      }
    }
  };
#endif
  static_assert(RPL % 8 == 0, "the ring swizzle of row q * RPL + g must not depend on q");
  const uint32_t serveBase = ((lane / LPR) * (uint32_t)RS) ^ rsw_of(lane / LPR);   // ring row (swizzled) of the first row this lane serves
  auto topup = [&]() {
#ifdef HSRLE_STAMPS
    tq0 = __builtin_readcyclecounter();
#endif
    if (!done) E += umin(umin((uint32_t)Q, lim - E), (uint32_t)R - (E - (sp & ~15u)));
    uint32_t ri[LPR];
    HS_EXCHANGE(ri, kPackLim ? (E | umin(15u, (lim - E) >> 4)) : E, LPR)      // E is a multiple of 16
    wave_sync();
    HS_XSTAMP(tx0)
    bool landed[LPR];
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      landed[q] = pfPos[q] < (ri[q] & ~15u);
      if (landed[q])                                                    // row q * RPL + lane / LPR: the q term is the instruction's offset field
        lds_st128(ring + ((serveBase ^ (pfPos[q] & RMASK)) + (uint32_t)q * RPL * RS), pf[q]);
    }
    HS_XSTAMP(tx1)
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
#ifdef HSRLE_REQ_STATS
      count_requests(landed[q] && (kPackLim ? (((pfPos[q] + (uint32_t)Q - (ri[q] & ~15u)) >> 4) < (ri[q] & 15u)) : (limq[q] - pfPos[q] > (uint32_t)Q)));
#endif
      if (landed[q])
      {
        // the chunk Q bytes further on exists iff it starts below lim
        if (kPackLim ? (((pfPos[q] + (uint32_t)Q - (ri[q] & ~15u)) >> 4) < (ri[q] & 15u)) : (limq[q] - pfPos[q] > (uint32_t)Q))
        {
          pfPos[q] += (uint32_t)Q;
          pf[q] = ld128(payload + myBase[q] + pfPos[q]);
        }
        else
          pfPos[q] = 0xFFFFFFFFu;                                         // this lane's part of the stream is complete
      }
    }
    HS_XSTAMP(tx2)
  };
  // prologue: fill the ring, then read the stream header from it
  wave_sync();
#pragma unroll
  for (int q = 0; q < LPR; q++)
  {
    const int r = (int)((uint32_t)q * RPL + lane / LPR);
    const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)myBase0, r, 64), hi32 = (uint32_t)__shfl((int)(uint32_t)(myBase0 >> 32), r, 64);
    myBase[q] = ((uint64_t)hi32 << 32) | lo32;
  }
#pragma unroll
  for (int q = 0; q < LPR; q++)
  {
    const int r = (int)((uint32_t)q * RPL + lane / LPR);
    const uint32_t rowLim = (uint32_t)__shfl((int)lim, r, 64);
    const uint32_t rowE0 = (uint32_t)__shfl((int)E, r, 64);             // (floor16(g0) of that row)
    if constexpr (!kPackLim) limq[q] = rowLim;
    pfPos[q] = (lane % LPR) * 16u;                                    // the first Q bytes of every stream from its first chunk on ...
    if (pfPos[q] < rowE0) pfPos[q] += (uint32_t)Q;                     // (this lane's chunk of the first line lies in front of the stream: the next line's)
#ifdef HSRLE_REQ_STATS
    count_requests(pfPos[q] < rowLim);
#endif
    if (pfPos[q] < rowLim) pf[q] = ld128(payload + myBase[q] + pfPos[q]);
    else pfPos[q] = 0xFFFFFFFFu;
  }
  for (int k = 0; k < (R / Q > 0 ? R / Q : 1); k++)
  {
    topup();                                                          // ... land (sp is still 0); the next piece is requested and
    wave_sync();                                                      //     flies during the first step's decode
  }

  const uint32_t *const rec = (entries != nullptr && active) ? entries + (uint64_t)(b - entryBase) * kEntryRecDwords : nullptr;
  if (rec != nullptr)
  {
    const uint32_t rf = rec[4];
    sp = g0; lit = rec[2]; run = rec[3];
    phase = rf & 0xFFu;
    last = (rf & REC_LAST) != 0u;
    if constexpr (SGL) singleVar = (rf & REC_SINGLE) != 0u;
    if (lit > slen - sp) { err |= DEC_ERR_STREAM; done = true; }
  }
  else if (active)
  {
    sp = g0 + TR::kHeaderSize;
    const u32x4 hdv = ring_win16(g0);                                  // the stream header: 16 bytes at the stream's first byte
    const uint32_t hd8 = hdv.z & 0xFFu, hd9 = (hdv.z >> 8) & 0xFFu;    // bytes 8 and 9: mode / the Single symbol

    if (slen < g0 + TR::kHeaderSize + 2u || hdv.x != blen || hdv.y != slen - g0)
    {
      err |= DEC_ERR_HEADER;
      done = true;
    }
    else if constexpr (TR::kShortSingle)
    {
      set_sym(u32x4{ hd8, 0, 0, 0 });                                    // the stream's one symbol sits behind the header (rleX_Xsl_short.h:1211-1216)
      sp = g0 + 9;
    }
    else if constexpr (S == 1 && !TR::kLut && !TR::kShort)
    {
      const uint32_t mode = hd8;
      if (mode == 1u)
      {
        if constexpr (SGL) { singleVar = true; set_sym(u32x4{ hd9, 0, 0, 0 }); sp = g0 + 10; }
        else { err |= DEC_ERR_MODE; done = true; }
      }
      else if (mode != 0u) { err |= DEC_ERR_MODE; done = true; }
    }
  }

  if constexpr (TR::kMtf)
  {
    // initial move-to-front list (reference: rleX_Xsl.h:533-543, rleX_Xsl_short.h:1218-1229; the 1-symbol list starts with 0)
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
    for (int k = 0; k < TR::K; k++)
    {
      const u32x4 m = mask_symbol<S>(u32x4{ init[k] * 0x01010101u, init[k] * 0x01010101u, init[k] * 0x01010101u, init[k] * 0x01010101u });
#pragma unroll
      for (int w = 0; w < TR::SW; w++)
        lut[k][w] = m[w];
    }
    set_sym(u32x4{ 0, 0, 0, 0 });
  }
  else if (!single && !TR::kShortSingle)
  {
    set_sym(u32x4{ 0, 0, 0, 0 }); // Packed decoders start with symbol 0 (rleX_extreme_cpu_decode.h:31-37, q5)
  }
  if (rec != nullptr)
  {
    set_sym(u32x4{ rec[6], rec[7], rec[8], rec[9] });
    if constexpr (TR::kMtf)
    {
#pragma unroll
      for (int k = 0; k < TR::K; k++)
#pragma unroll
        for (int w = 0; w < TR::SW; w++)
          lut[k][w] = rec[10 + k * TR::SW + w];
    }
  }

  // CAPPED ROUNDS (round 5).  A round used to last until every lane had its T output bytes, i.e. as long as the lane with the most packets: on video-shaped
  // data (many short packets, unevenly spread) half the lane trips were idle.  With CAP != 0 a round is at most CAP packets per lane: the tile row is a RING of T
  // bytes addressed by the output position mod T, a lane may run up to T bytes ahead of what it has flushed (target = base + T), the flush writes the 64-byte
  // halves that are complete (whole write requests, as before) and a lane that waits for stream bytes goes on in the next round -- the sum of the per-round
  // maxima becomes (nearly) the maximum of the sums.  Measured per family, 4 GiB, same box (LAB_NOTEBOOK.md, round 5 calls 48, 52, 53): 2 .. 8 byte symbols CAP = 4:
  // video-shaped +8 ... +64 % (rle64_7symlut_byte_short_greedy streams 886 -> 1 422 GiB/s), run data -5 ... +10 %, random bytes +-1 %; the 8 bit Short codecs with a
  // 3 / 7 symbol list CAP = 8: +13 ... +22 % / -5 ... +2 %; the other 8 bit codecs and 128 bit: -7 ... +6 % with any CAP (few, large packets: what a round costs
  // besides its trips is what the cap multiplies) -> 0.
#define HS_ROWPOS(pos) (CAP != 0u ? ((pos) & ((uint32_t)T - 1u)) : ((pos) - base))
#define HS_ROWWRAP(x) (CAP != 0u ? ((x) & ((uint32_t)T - 1u)) : (x))
  uint32_t base = 0;      // block offset of the first byte of the tile row (multiple of 16); the row holds [base, o)

  // every spin is bounded: a malformed stream (or a bug) ends as DEC_ERR_STREAM, never as a hang
  // output rounds + worst-case starved rounds (>= 16 stream bytes each); capped rounds: a round is at least one packet (>= 1 output byte) of every lane that is not starved
  uint32_t roundsLeft = (CAP != 0u ? B : B / (uint32_t)T) + B / 16u + 64u;

#ifdef HSRLE_STAMPS
  unsigned long long tIssue = 0, tDecode = 0, tFlush = 0, tLand = 0, nRounds = 0, nIter = 0, t0, t1;
#define HS_STAMP(acc) { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; }
#else
#define HS_STAMP(acc)
#endif

  [[maybe_unused]] bool skippedFlush = false;                          // capped rounds: the last trip went round again without a flush
  while (__ballot(!done && o < blen) != 0ull)
  {
    if (roundsLeft-- == 0u) { err |= DEC_ERR_STREAM; break; }
#ifdef HSRLE_STAMPS
    t0 = __builtin_readcyclecounter(); nRounds++;
#endif

    [[maybe_unused]] bool starved = false;                              // capped rounds: this lane stopped for LITERAL bytes that were not resident (not for its cap, not for a header)
    static_assert(CAP == 0u || (T == Q && T == 128), "capped rounds: one step per round");
    const uint32_t flushTarget = CAP != 0u ? umin(base + (uint32_t)T, blen) : umin((o / (uint32_t)T + 1u) * (uint32_t)T, blen);

#pragma unroll 1
    for (int step = 0; step < T / Q; step++)
    {
    const uint32_t target = CAP != 0u ? flushTarget : umin((o / (uint32_t)Q + 1u) * (uint32_t)Q, flushTarget);
    [[maybe_unused]] uint32_t tripsLeft = CAP;
    [[maybe_unused]] uint32_t itersLeft = 2u * (uint32_t)Q + 16u;

    // Pass 0 decodes with the ring as it stands.  A lane whose packet needs bytes that were not resident (the ring holds
    // < R bytes behind a misaligned sp; literal-heavy stretches consume more than that per step) stops early; the top-up that
    // lands after pass 0 holds what it waits for, so pass 1 lets exactly those lanes finish the step.  No lane ever falls
    // a step behind, and every row stays T-aligned in the output.
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++)
    {
    const uint32_t avail0 = E;                                         // bytes [.., avail0) are readable during this pass
    if constexpr (CAP != 0u) { starved = false; tripsLeft = CAP; }

    if constexpr (S == 1)
    {
      // ================= 8 bit: one packet per loop trip; lane flags live in one VGPR =================
      constexpr uint32_t SHORT_SINGLE = TR::kPacked ? 2u : 4u;
      constexpr uint32_t F_DONE = 1u, F_STALL = 2u, F_LAST = 4u;        // F_STALL: waits for the next pass's ring bytes
      uint32_t fl = (done ? F_DONE : 0u) | (last ? F_LAST : 0u);
      // a header may be parsed at sp iff sp <= spOK: MAXHDR bytes resident (or the stream completely loaded) and >= 2 bytes left
      const uint32_t spOK = umin((avail0 < lim) ? avail0 - MAXHDR : 0xFFFFFFFFu, slen - 2u);

      for (;;)
      {
        const bool act = (fl & (F_DONE | F_STALL)) == 0u && o < target;
        if (__ballot(act) == 0ull) break;
        if constexpr (CAP != 0u) { if (tripsLeft-- == 0u) break; }
        // every trip of an active lane either consumes stream bytes (a header is >= 2 bytes), produces output bytes, or
        // deactivates the lane (done / stall), so the loop is bounded by the resident stream bytes plus the step's output

        if (act && (lit | run) == 0u)
        {
          if ((fl & F_LAST) != 0u || sp > spOK)
          {
            // rare: end of the stream, header not resident yet, or a truncated stream
            if ((fl & F_LAST) != 0u) fl |= F_DONE;
            else if (sp + 2u > slen) { err |= DEC_ERR_STREAM; fl |= F_DONE; }
            else fl |= F_STALL;
          }
          else
          {
            // ---------------- packet header: 16 stream bytes at sp, fields picked with shifts ----------------
            const u32x4 hv = ring_win16(sp);
            [[maybe_unused]] const uint64_t lo = (uint64_t)hv.x | ((uint64_t)hv.y << 32), hi = (uint64_t)hv.z | ((uint64_t)hv.w << 32);
            uint32_t cnt, pos, nsym = sym4, range, used, endNow = 0, hbad = 0;

            if constexpr (TR::kShort)
            {
              // Short family (rleX_Xsl_short.h:1255-1310): [idx | count3 | range] in one byte, or -- count3 all ones -- three bytes with a
              // 9 bit count and an SRB bit range, each extended by 2 / 4 bytes when it reads 1 / 0; both fields carry value + 2
              const uint32_t p1 = hv.x & 0xFFu, p2 = (hv.x >> 8) & 0xFFu, p3 = (hv.x >> 16) & 0xFFu;
              [[maybe_unused]] const uint32_t idx = (TR::K > 0) ? p1 >> (TR::SCB + TR::SRBP) : 0u;
              const uint32_t c3 = (p1 >> TR::SRBP) & TR::SCINV;
              const bool longf = c3 == TR::SCINV;
              uint32_t lc = (p2 >> (TR::SRB - 8u)) | ((p1 & TR::SMAXPR) << (16u - TR::SRB));
              uint32_t lr = p3 | ((p2 & ((1u << (TR::SRB - 8u)) - 1u)) << 8);
              pos = longf ? 3u : 1u;
              const uint32_t cext = alignbyte(hv.y, hv.x, 3u);
              const bool c32 = longf && lc == 0u, c16 = longf && lc == 1u;
              lc = c32 ? cext : (c16 ? (cext & 0xFFFFu) : lc);
              pos += c32 ? 4u : (c16 ? 2u : 0u);
              const uint32_t rext = ex32(lo, hi, pos);                    // pos <= 7
              const bool r32 = longf && lr == 0u, r16 = longf && lr == 1u;
              lr = r32 ? rext : (r16 ? (rext & 0xFFFFu) : lr);
              pos += r32 ? 4u : (r16 ? 2u : 0u);
              endNow = (r16 && lr == 0u) ? 1u : 0u;
              cnt = longf ? lc : c3 + 2u;
              range = longf ? lr : (p1 & TR::SMAXPR) + 2u;
              uint32_t sb = ex32(lo, hi, pos) & 0xFFu;                    // pos <= 11: the symbol byte of a packet that carries one
              if constexpr (TR::kShortSingle)
                sb = sym4 & 0xFFu;                                       // no symbols in the packets
              else if constexpr (TR::K == 0)
                pos += 1u;
              else
              {
                const bool isNew = idx == (uint32_t)TR::K;
                pos += isNew ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < TR::K; k++)
                  if (idx == (uint32_t)k) sb = lut[k][0];
                const uint32_t limit = isNew ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
                for (int k = TR::K - 1; k >= 1; k--)
                  if ((uint32_t)k <= limit) lut[k][0] = lut[k - 1][0];
                lut[0][0] = sb;
              }
              nsym = sb * 0x01010101u;
              used = pos;
              hbad = (!endNow && range < 2u) ? 1u : 0u;
              range = (range >= 2u) ? range - 1u : 0u;                    // literal count + 1, like the other families
            }
            else if constexpr (TR::kLut)
            {
              const uint32_t w16 = hv.x & 0xFFFFu;
              const uint32_t idx = w16 >> (FAM == LUT3 ? 14 : 13);
              const uint32_t c7 = (w16 >> TR::RB) & 0x7Fu;
              const uint32_t r7 = w16 & ((1u << TR::RB) - 1u);
              const bool isNew = idx == (uint32_t)TR::K;
              pos = isNew ? 3u : 2u;
              // the move-to-front list lives in lut[k][0], one byte per entry
              uint32_t sb = (hv.x >> 16) & 0xFFu;
#pragma unroll
              for (int k = 0; k < TR::K; k++)
                if (idx == (uint32_t)k) sb = lut[k][0];
              const uint32_t limit = isNew ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
              for (int k = TR::K - 1; k >= 1; k--)
                if ((uint32_t)k <= limit) lut[k][0] = lut[k - 1][0];
              lut[0][0] = sb;
              nsym = sb * 0x01010101u;
              const uint32_t cext = ex32(lo, hi, pos);
              cnt = (c7 == 0u) ? cext : (c7 == 1u ? (cext & 0xFFFFu) : c7);
              pos += (c7 == 0u) ? 4u : (c7 == 1u ? 2u : 0u);
              const uint32_t rext = ex32(lo, hi, pos);
              range = (r7 == 0u) ? rext : (r7 == 1u ? (rext & 0xFFFFu) : r7);
              used = pos + ((r7 == 0u) ? 4u : (r7 == 1u ? 2u : 0u));
              endNow = (r7 == 1u && range == 0u) ? 1u : 0u;
              hbad = (!endNow && range < 2u) ? 1u : 0u;
              range = (range >= 2u) ? range - 1u : 0u;                    // literal count + 1, like the other families
            }
            else
            {
              const uint32_t b0 = hv.x & 0xFFu;

              if constexpr (!TR::kPacked)
              {
                cnt = single ? b0 : ((hv.x >> 8) & 0xFFu);                // multi: sym, cnt ...   single: cnt ...
                pos = single ? 1u : 2u;
                nsym = single ? sym4 : __builtin_amdgcn_perm(hv.x, hv.x, 0u); // byte 0 broadcast
                const uint32_t c32 = ex32(lo, hi, pos);
                const bool longc = cnt == 0u;
                cnt = longc ? c32 : cnt;
                pos += longc ? 4u : 0u;

                const uint32_t w = ex32(lo, hi, pos);
                const uint32_t r0 = w & 0xFFu;
                const bool longr = r0 == 0u;
                const uint32_t r32 = ex32(lo, hi, pos + 1u);
                range = longr ? r32 : r0;
                used = pos + (longr ? 5u : 1u);
                endNow = (longr && range == 0u) ? 1u : 0u;
              }
              else
              {
                // Packed: [cnt7 | same-flag] [cnt32 if cnt7 == 0] [symbol unless same] [range7 | long-flag, or range32]
                // everything sits in the 8 bytes A:Bd that start at byte 0 (short count) or byte 4 (long count)
                const uint32_t c7 = single ? b0 : (b0 & 0x7Fu);
                const bool longc = c7 == 0u;
                cnt = longc ? alignbyte(hv.y, hv.x, 1u) : c7;
                const uint32_t A = longc ? hv.y : hv.x, Bd = longc ? hv.z : hv.y;
                const bool newSym = !single && !(b0 & 0x80u);
                nsym = newSym ? __builtin_amdgcn_perm(A, A, 0x01010101u) : sym4;   // byte 1 of A, broadcast
                const uint32_t posr = newSym ? 2u : 1u;                   // the range field, relative to A
                pos = (longc ? 4u : 0u) + posr;
                const uint32_t w = alignbyte(Bd, A, posr);
                const uint32_t r0 = w & 0xFFu;

                if (!single)
                {
                  const bool longr = (r0 & 1u) != 0u;                     // 7-bit-or-4-byte range (rle8_extreme_cpu.h:1883-1899)
                  range = longr ? (w >> 1) : (r0 >> 1);
                  used = pos + (longr ? 4u : 1u);
                  endNow = (longr && range == 0u) ? 1u : 0u;
                }
                else
                {
                  const bool longr = r0 == 0u;
                  const uint32_t r32 = alignbyte(Bd, A, posr + 1u);
                  range = longr ? r32 : r0;
                  used = pos + (longr ? 5u : 1u);
                  endNow = (longr && range == 0u) ? 1u : 0u;
                }
              }
            }

            const uint32_t shortv = single ? SHORT_SINGLE : TR::SHORT;
            const bool lastNow = endNow || cnt == 0u;
            sym4 = nsym;
            lit = (range == 0u || endNow) ? 0u : range - 1u;             // a 7 bit range byte of 0x00 carries no literals (q11)
            run = lastNow ? 0u : (TR::kShort ? cnt + TR::SMINS - 2u : cnt + shortv - (TR::kLut ? 2u : 1u));
            if (lastNow) fl |= F_LAST;
            sp += used;
            if (hbad || sp > slen || lit > slen - sp || (lit == 0u && run == 0u && !lastNow)) { err |= DEC_ERR_STREAM; fl |= F_DONE; }
          }
        }

        if (act && (fl & (F_DONE | F_STALL)) == 0u)
        {
          // ---- literals: tile chunks at the aligned positions A + 16k receive ring bytes [sp - c + 16k, +16) ----
          if (lit != 0u)
          {
            const uint32_t resident = (avail0 > sp) ? avail0 - sp : 0u;
            const uint32_t n = umin(umin(lit, target - o), resident);
            if (n == 0u) { fl |= F_STALL; starved = true; }                // literals not resident yet: continue next pass
            else
            {
              const uint32_t q = HS_ROWPOS(o), c = q & 15u;
              const uint32_t d0 = q & ~15u;
              const uint32_t srcp = sp - c;
              const uint32_t total = c + n;
              const uint32_t s8 = srcp & ~7u, sh = srcp & 7u;
              uint64_t xa = lds_ld64(ring + (rowx ^ (s8 & RMASK))), xb = lds_ld64(ring + (rowx ^ ((s8 + 8u) & RMASK))), xc = lds_ld64(ring + (rowx ^ ((s8 + 16u) & RMASK)));
              u32x4 w = HS_MERGE_LOW(acc, funnel24(xa, xb, xc, sh), c);     // keep the c valid bytes of the straddled chunk
              lds_st128(row + (d0 ^ tsw), w);
              for (uint32_t k = 16; k < total; k += 16)
              {
                xa = xc;
                xb = lds_ld64(ring + (rowx ^ ((s8 + k + 8u) & RMASK)));
                xc = lds_ld64(ring + (rowx ^ ((s8 + k + 16u) & RMASK)));
                w = funnel24(xa, xb, xc, sh);
                lds_st128(row + (HS_ROWWRAP(d0 + k) ^ tsw), w);
              }
              acc = w;
              sp += n;
              lit -= n;
              o += n;
            }
          }

          // ---- run: the same aligned chunks filled with the byte-broadcast symbol ----
          if (lit == 0u && run != 0u && o < target)
          {
            const uint32_t m = umin(run, target - o);
            const uint32_t q = HS_ROWPOS(o), c = q & 15u;
            const uint32_t d0 = q & ~15u;
            const uint32_t total = c + m;
            const u32x4 v = u32x4{ sym4, sym4, sym4, sym4 };
            const u32x4 w = HS_MERGE_LOW(acc, v, c);
            lds_st128(row + (d0 ^ tsw), w);
            for (uint32_t k = 16; k < total; k += 16)
              lds_st128(row + (HS_ROWWRAP(d0 + k) ^ tsw), v);
            acc = (total <= 16u) ? w : v;
            run -= m;
            o += m;
          }
        }
      }

      done = (fl & F_DONE) != 0u;
      last = (fl & F_LAST) != 0u;
    }
    else
    {
    // ================= wider symbols: per-lane loop; lane flags (done, last) live in one VGPR as in the 8 bit loop =================
    constexpr uint32_t F_DONE = 1u, F_LAST = 4u;
    uint32_t fl = (done ? F_DONE : 0u) | (last ? F_LAST : 0u);
    while ((fl & F_DONE) == 0u && o < target)
    {
      if constexpr (CAP != 0u) { if (tripsLeft-- == 0u) break; }
      if (itersLeft-- == 0u) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
#ifdef HSRLE_STAMPS
      nIter++;
#endif

      if (lit == 0 && run == 0)
      {
        if ((fl & F_LAST) != 0u) { fl |= F_DONE; break; }
        if (sp + 2u > slen) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
        if (avail0 - sp < MAXHDR && avail0 < lim) break;               // header not resident yet: continue next round

        // ---------------- packet header ----------------
        [[maybe_unused]] uint32_t cnt = 0, range = 0, used = 0;   // (every path that reads them assigns them first; the zeros only silence -Wsometimes-uninitialized: they are dead stores the compiler removes)
        bool endNow = false;

        if constexpr (S <= 8 && !TR::kLut && !TR::kShort)
        {
          // Round 4, plain and Packed packets of 2 .. 8 byte symbols: the header in ONE ring read of 32 bytes, every field at a compile-time
          // position of the window (shifted by one dword behind a 32-bit count) -- the code below reads the ring up to three times (first
          // window, fields behind the count, the symbol).  rleX_extreme_cpu_decode.h:129-162, rle8_extreme_cpu.h:1849-1899 (Packed).
          uint32_t d[6];
          ring_win24(sp, d);
          if constexpr (!TR::kPacked)
          {
            // [symbol S] [count 8, or 0 + count 32] [range 8, or 0 + range 32]
            set_sym(mask_symbol<S>(u32x4{ d[0], d[1], 0u, 0u }));
            const uint32_t cb = u32c<S>(d) & 0xFFu;
            const bool longc = cb == 0u;
            cnt = longc ? u32c<S + 1>(d) : cb;
            const uint32_t w = longc ? u32c<S + 5>(d) : u32c<S + 1>(d);
            const uint32_t r0 = w & 0xFFu;
            const bool longr = r0 == 0u;
            const uint32_t r32 = longc ? u32c<S + 6>(d) : u32c<S + 2>(d);
            range = longr ? r32 : r0;
            endNow = longr && range == 0u;
            used = (uint32_t)S + (longc ? 5u : 1u) + (longr ? 5u : 1u);
          }
          else
          {
            // [count 7 | same-symbol flag] [count 32 if count 7 == 0] [symbol S unless same] [range: 7 bits | long flag or 32 bits (byte-aligned),
            //  8 bits or 0 + 32 bits (symbol-aligned)]
            const uint32_t b0 = d[0] & 0xFFu, c7 = b0 & 0x7Fu;
            const bool longc = c7 == 0u;
            cnt = longc ? u32c<1>(d) : c7;
            const uint32_t e[6] = { longc ? d[1] : d[0], longc ? d[2] : d[1], longc ? d[3] : d[2], longc ? d[4] : d[3], longc ? d[5] : d[4], d[5] };
            const bool newSym = (b0 & 0x80u) == 0u;
            if (newSym) set_sym(mask_symbol<S>(u32x4{ u32c<1>(e), u32c<5>(e), 0u, 0u }));
            const uint32_t w = newSym ? u32c<1 + S>(e) : u32c<1>(e);
            const uint32_t r0 = w & 0xFFu;
            uint32_t rl;
            if constexpr (TR::kRange7)
            {
              const bool longr = (r0 & 1u) != 0u;
              range = longr ? (w >> 1) : (r0 >> 1);
              rl = longr ? 4u : 1u;
              endNow = longr && range == 0u;
            }
            else
            {
              const bool longr = r0 == 0u;
              const uint32_t r32 = newSym ? u32c<2 + S>(e) : u32c<2>(e);
              range = longr ? r32 : r0;
              rl = longr ? 5u : 1u;
              endNow = longr && range == 0u;
            }
            used = 1u + (longc ? 4u : 0u) + (newSym ? (uint32_t)S : 0u) + rl;
          }
          lit = (range == 0u) ? 0u : range - 1u;
          run = (cnt == 0u) ? 0u : (TR::kAligned ? (cnt + TR::SHORT / (uint32_t)S - 1u) * (uint32_t)S : cnt + TR::SHORT - 1u);
        }
        else if constexpr (S <= 8 && TR::kLut)
        {
          // Round 4, the 3 / 7 symbol LUT codecs: ONE ring read of 32 bytes and the header grammar of hsrle_parse.hip.h (every field extracted
          // at the positions it can have, then selected -- no second / third ring read for the fields behind the count or the symbol, no
          // nested branches that a wave executes on both sides as soon as one lane differs).  The same function the index walks use.
          // Same-box A/B at 4 GiB (experiments/r04/call39.sh), run data / video-shaped: rle32_7symlut_byte +9.5 % / +3.0 %, rle24_3symlut_byte
          // +4.2 / +4.1, rle16_7symlut_byte +2.8 / +5.3, rle64_3symlut_byte +1.7 / +2.5, rle16_3symlut_byte -4.2 / +5.3 (it had the
          // one-read parse of the unextended packets below).  For plain / Packed / Short packets the same parse LOSES 2 ... 8 %: their
          // general parse branches rarely diverge, and the selects are paid by every packet -- they keep the code below.
          uint32_t dw[6];
          ring_win24(sp, dw);
          const uint64_t lo = (uint64_t)dw[0] | ((uint64_t)dw[1] << 32), hi = (uint64_t)dw[2] | ((uint64_t)dw[3] << 32), ex = (uint64_t)dw[4] | ((uint64_t)dw[5] << 32);
          struct NoReader { __device__ __forceinline__ uint32_t load32(uint32_t) const { return 0u; } };
          const Pkt k = parse_window<FAM, S, AL>(NoReader{}, lo, hi, ex, sp, slen, false);
          if (k.bad) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
          // the S symbol bytes at window offset k.symAt - sp (<= 11: behind at most a three-byte header with two 32-bit fields)
          const uint32_t so = k.symAt - sp;
          u32x4 nv = u32x4{ ex32x(lo, hi, ex, so), 0u, 0u, 0u };
          if constexpr (S > 4) nv.y = ex32x(lo, hi, ex, so + 4u);
          nv = mask_symbol<S>(nv);
          if constexpr (TR::kMtf)
          {
            // move-to-front list (rleX_Xsl.h:597-650): op < K takes slot op to the front, op == K pushes the packet's symbol
            if (!(TR::kShort && k.last && k.run == 0u && k.lit == 0u))     // (a Short END terminator carries no list operation)
            {
              const uint32_t idx = k.op;
              uint32_t tmp[TR::SW];
#pragma unroll
              for (int w = 0; w < TR::SW; w++) tmp[w] = nv[w];
#pragma unroll
              for (int kk = 0; kk < TR::K; kk++)
                if (idx == (uint32_t)kk)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) tmp[w] = lut[kk][w];
                }
              const uint32_t limit = (idx >= (uint32_t)TR::K) ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
              for (int kk = TR::K - 1; kk >= 1; kk--)
                if ((uint32_t)kk <= limit)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) lut[kk][w] = lut[kk - 1][w];
                }
              u32x4 pv = u32x4{ 0, 0, 0, 0 };
#pragma unroll
              for (int w = 0; w < TR::SW; w++) { lut[0][w] = tmp[w]; pv[w] = tmp[w]; }
              set_sym(pv);
            }
          }
          else
          {
            if (k.hasSym) set_sym(nv);
          }
          used = k.used; lit = k.lit; run = k.run;
          cnt = k.last ? 0u : 1u;                                         // (only feeds the F_LAST test below)
        }
        else
        {
#define HS_RD16(off) ring_win16(sp + (off))
          // S > 1: the first 16 header bytes in one read; the (rare) fields behind them with a second read
          const u32x4 hv = HS_RD16(0);
          const uint64_t lo = (uint64_t)hv.x | ((uint64_t)hv.y << 32), hi = (uint64_t)hv.z | ((uint64_t)hv.w << 32);

          // ---- fast path of the 3 / 7 symbol LUT codecs with 2 and 4 byte symbols: a packet whose count and range are not extended has
          //      every field at a compile-time position of the first window -- one ring read instead of two (the new symbol) or three,
          //      no 64 bit shifts, no nested branches.  Everything else (extended fields, the terminators) takes the general parse below;
          //      same results, field for field.  Same-box A/B, 8 GiB: rle16_3symlut_byte +11 % / +4 % (run data / video-shaped),
          //      rle32_3symlut_sym +9 % / +13 %, rle16_7symlut_sym +2 % / +12 %.  The same idea for the plain / Packed / Short packets and for
          //      3, 6, 8 byte symbols was built and measured too: -1 ... -5 % (their general parse has no divergent second read to save,
          //      and a wave in which ANY lane needs the general parse executes both) -- not shipped ----
          bool fastDone = false;
          if constexpr (TR::kLut && (S == 2 || S == 4))
          {
            const uint32_t w16 = hv.x & 0xFFFFu;
            const uint32_t idx = w16 >> (FAM == LUT3 ? 14 : 13);
            const uint32_t c7 = (w16 >> TR::RB) & 0x7Fu, r7 = w16 & ((1u << TR::RB) - 1u);
            if (c7 >= 2u && r7 >= 2u)                                       // neither field extended (rleX_Xsl.h:723-748)
            {
              const bool isNew = idx == (uint32_t)TR::K;
              const u32x4 nv = win_symbol<S, 2>(hv);
              uint32_t tmp[TR::SW];
#pragma unroll
              for (int w = 0; w < TR::SW; w++) tmp[w] = nv[w];
#pragma unroll
              for (int k = 0; k < TR::K; k++)
                if (idx == (uint32_t)k)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) tmp[w] = lut[k][w];
                }
              const uint32_t limit = isNew ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
              for (int k = TR::K - 1; k >= 1; k--)
                if ((uint32_t)k <= limit)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) lut[k][w] = lut[k - 1][w];
                }
              u32x4 pv = u32x4{ 0, 0, 0, 0 };
#pragma unroll
              for (int w = 0; w < TR::SW; w++) { lut[0][w] = tmp[w]; pv[w] = tmp[w]; }
              set_sym(pv);
              cnt = c7; range = r7;
              used = 2u + (isNew ? (uint32_t)S : 0u);
              lit = range - 2u;
              run = TR::kAligned ? (cnt + 3u / (uint32_t)S - 2u) * (uint32_t)S : cnt + 1u;
              fastDone = true;
            }
          }

          if (!fastDone)
          {
          if constexpr (TR::kShort)
          {
            // Short family header, see the 8 bit loop
            const uint32_t p1 = hv.x & 0xFFu;
            [[maybe_unused]] const uint32_t idx = (TR::K > 0) ? p1 >> (TR::SCB + TR::SRBP) : 0u;
            const uint32_t c3 = (p1 >> TR::SRBP) & TR::SCINV;
            if (c3 != TR::SCINV)
            {
              cnt = c3 + 2u;
              range = (p1 & TR::SMAXPR) + 2u;
              used = 1;
            }
            else
            {
              const uint32_t p2 = (hv.x >> 8) & 0xFFu, p3 = (hv.x >> 16) & 0xFFu;
              cnt = (p2 >> (TR::SRB - 8u)) | ((p1 & TR::SMAXPR) << (16u - TR::SRB));
              range = p3 | ((p2 & ((1u << (TR::SRB - 8u)) - 1u)) << 8);
              used = 3;
              if (cnt == 0u) { cnt = ex32(lo, hi, 3u); used = 7; }
              else if (cnt == 1u) { cnt = ex32(lo, hi, 3u) & 0xFFFFu; used = 5; }
              const uint32_t rext = ex32(lo, hi, used);
              if (range == 0u) { range = rext; used += 4; }
              else if (range == 1u) { range = rext & 0xFFFFu; used += 2; endNow = (range == 0u); }
            }

            if (!endNow)
            {
              if constexpr (TR::K == 0)
              {
                set_sym(mask_symbol<S>(HS_RD16(used)));                  // every packet carries its symbol
                used += S;
              }
              else if (idx != 0u)
              {
                uint32_t tmp[TR::SW];
                if (idx == (uint32_t)TR::K)
                {
                  const u32x4 nv = mask_symbol<S>(HS_RD16(used));
                  used += S;
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) tmp[w] = nv[w];
                }
                else
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) tmp[w] = lut[0][w];
#pragma unroll
                  for (int k = 1; k < TR::K; k++)
                    if (idx == (uint32_t)k)
                    {
#pragma unroll
                      for (int w = 0; w < TR::SW; w++) tmp[w] = lut[k][w];
                    }
                }
                const uint32_t limit = (idx == (uint32_t)TR::K) ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
                for (int k = TR::K - 1; k >= 1; k--)
                  if ((uint32_t)k <= limit)
                  {
#pragma unroll
                    for (int w = 0; w < TR::SW; w++) lut[k][w] = lut[k - 1][w];
                  }
                u32x4 pv = u32x4{ 0, 0, 0, 0 };
#pragma unroll
                for (int w = 0; w < TR::SW; w++) { lut[0][w] = tmp[w]; pv[w] = tmp[w]; }
                set_sym(pv);
              }
            }

            if (!endNow && range < 2u) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
            lit = endNow ? 0u : range - 2u;
            run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + TR::SMINS / (uint32_t)S - 2u) * (uint32_t)S : cnt + TR::SMINS - 2u);
          }
          else if constexpr (TR::kLut)
          {
            const uint32_t w16 = hv.x & 0xFFFFu;
            used = 2;
            const uint32_t idx = w16 >> (FAM == LUT3 ? 14 : 13);
            cnt = (w16 >> TR::RB) & 0x7Fu;
            range = w16 & ((1u << TR::RB) - 1u);

            if (idx != 0)
            {
              uint32_t tmp[TR::SW];
              if (idx == (uint32_t)TR::K)
              {
                const u32x4 nv = mask_symbol<S>(HS_RD16(2));
                used += S;
#pragma unroll
                for (int w = 0; w < TR::SW; w++) tmp[w] = nv[w];
              }
              else
              {
#pragma unroll
                for (int w = 0; w < TR::SW; w++) tmp[w] = lut[0][w];
#pragma unroll
                for (int k = 1; k < TR::K; k++)
                  if (idx == (uint32_t)k)
                  {
#pragma unroll
                    for (int w = 0; w < TR::SW; w++) tmp[w] = lut[k][w];
                  }
              }
              const uint32_t limit = (idx == (uint32_t)TR::K) ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
              for (int k = TR::K - 1; k >= 1; k--)
                if ((uint32_t)k <= limit)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) lut[k][w] = lut[k - 1][w];
                }
              u32x4 pv = u32x4{ 0, 0, 0, 0 };
#pragma unroll
              for (int w = 0; w < TR::SW; w++) { lut[0][w] = tmp[w]; pv[w] = tmp[w]; }
              set_sym(pv);
            }

            // the extension fields sit at most at byte 2 + 8 + 4: re-read behind the symbol
            const u32x4 ev = (used > 2u) ? HS_RD16(used) : u32x4{ (uint32_t)(lo >> 16), (uint32_t)(lo >> 48) | ((uint32_t)hi << 16), (uint32_t)(hi >> 16), (uint32_t)(hi >> 48) };
            const uint64_t elo = (uint64_t)ev.x | ((uint64_t)ev.y << 32);
            uint32_t ep = 0;
            if (cnt == 0) { cnt = (uint32_t)elo; ep = 4; }
            else if (cnt == 1) { cnt = (uint32_t)elo & 0xFFFFu; ep = 2; }
            const uint32_t rext32 = (ep == 0u) ? (uint32_t)elo : ((ep == 2u) ? (uint32_t)(elo >> 16) : (uint32_t)(elo >> 32));
            if (range == 0) { range = rext32; ep += 4; }
            else if (range == 1) { range = rext32 & 0xFFFFu; ep += 2; endNow = (range == 0); }
            used += ep;

            if (!endNow && range < 2u) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
            lit = endNow ? 0u : range - 2u;
            run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + 3u / (uint32_t)S - 2u) * (uint32_t)S : cnt + 1u);
          }
          else
          {
            uint32_t x = 0;
            if constexpr (!TR::kPacked)
            {
              set_sym(hv);
              used = S;
            }
            else
            {
              x = hv.x & 0xFFu;
              used = 1;
            }

            // count byte (+ u32), optional symbol, range: everything behind `used` comes from one more read
            const u32x4 tv = HS_RD16(used);
            const uint64_t tlo = (uint64_t)tv.x | ((uint64_t)tv.y << 32), thi = (uint64_t)tv.z | ((uint64_t)tv.w << 32);
            uint32_t tp;

            if constexpr (!TR::kPacked)
            {
              cnt = tv.x & 0xFFu; tp = 1;
              if (cnt == 0) { cnt = (uint32_t)(tlo >> 8); tp = 5; }
            }
            else
            {
              cnt = x & 0x7Fu; tp = 0;
              if (cnt == 0) { cnt = tv.x; tp = 4; }
              if (!(x & 0x80u))
              {
                set_sym(ring_win16(sp + used + tp));
                tp += S;
              }
            }

            // range field: for S == 16 it can start beyond the 16 bytes of tv
            uint32_t w;
            if (tp <= 12u) w = ex32(tlo, thi, tp);
            else w = ring_win16(sp + used + tp).x;
            const uint32_t r0 = w & 0xFFu;

            if constexpr (TR::kRange7)
            {
              if (r0 & 1u) { range = w >> 1; tp += 4; endNow = (range == 0); }
              else { range = r0 >> 1; tp += 1; }
            }
            else
            {
              range = r0; tp += 1;
              if (r0 == 0)
              {
                range = (tp <= 12u) ? ex32(tlo, thi, tp) : ring_win16(sp + used + tp).x;
                tp += 4;
                endNow = (range == 0);
              }
            }
            used += tp;

            lit = (range == 0) ? 0u : range - 1u;
            run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + TR::SHORT / (uint32_t)S - 1u) * (uint32_t)S : cnt + TR::SHORT - 1u);
          }
          }
#undef HS_RD16
        }

        sp += used;
        phase = 0;
        if (endNow || cnt == 0u) fl |= F_LAST;
        if (endNow) { lit = 0; run = 0; }

        if (sp > slen || lit > slen - sp) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
        if (lit == 0 && run == 0 && (fl & F_LAST) == 0u) { err |= DEC_ERR_STREAM; fl |= F_DONE; break; }
      }

      // ---- literals: tile chunks at the dword aligned positions A + 16k receive ring bytes [sp - c + 16k, +16) ----
      if (lit != 0)
      {
        const uint32_t resident = (avail0 > sp) ? avail0 - sp : 0u;
        const uint32_t n = umin(umin(lit, target - o), resident);
        if (n == 0) { starved = true; break; }                         // literals not resident yet: continue next round

        const uint32_t q = HS_ROWPOS(o), c = q & 15u;
        const uint32_t d0 = q & ~15u;
        const uint32_t srcp = sp - c;
        const uint32_t total = c + n;                                  // bytes from dst that must end up valid
        const uint32_t s8 = srcp & ~7u, sh = srcp & 7u;
        uint64_t xa = lds_ld64(ring + (rowx ^ (s8 & RMASK))), xb = lds_ld64(ring + (rowx ^ ((s8 + 8u) & RMASK))), xc = lds_ld64(ring + (rowx ^ ((s8 + 16u) & RMASK)));
        u32x4 w = HS_MERGE_LOW(acc, funnel24(xa, xb, xc, sh), c);         // keep the c valid bytes of the straddled chunk
        lds_st128(row + (d0 ^ tsw), w);
        for (uint32_t k = 16; k < total; k += 16)
        {
          xa = xc;
          xb = lds_ld64(ring + (rowx ^ ((s8 + k + 8u) & RMASK)));
          xc = lds_ld64(ring + (rowx ^ ((s8 + k + 16u) & RMASK)));
          w = funnel24(xa, xb, xc, sh);
          lds_st128(row + (HS_ROWWRAP(d0 + k) ^ tsw), w);
        }
        acc = w;
        sp += n;
        lit -= n;
        o += n;
      }

      // ---- run: the same aligned chunks filled with the symbol pattern ----
      if (lit == 0 && run != 0 && o < target)
      {
        const uint32_t m = umin(run, target - o);
        const uint32_t q = HS_ROWPOS(o), c = q & 15u;
        const uint32_t d0 = q & ~15u;
        const uint32_t total = c + m;

        {
          // the pattern byte for tile byte (A + j) is sym[(phase - c + j) mod S]
          const uint32_t ph = (phase + 16u * (uint32_t)S - c) % (uint32_t)S;
          if constexpr (16 % S == 0)
          {
            const u32x4 v = pattern_chunk16_s<S>(patv, ph);               // every chunk of the run holds the same 16 bytes
            const u32x4 w = HS_MERGE_LOW(acc, v, c);
            lds_st128(row + (d0 ^ tsw), w);
            for (uint32_t k = 16; k < total; k += 16)
              lds_st128(row + (HS_ROWWRAP(d0 + k) ^ tsw), v);
            acc = (total <= 16u) ? w : v;
          }
          else
          {
            uint32_t e0, e1, e2;
            pattern_dwords12(patv, ph, e0, e1, e2);
            u32x4 w = HS_MERGE_LOW(acc, (u32x4{ e0, e1, e2, e0 }), c);
            lds_st128(row + (d0 ^ tsw), w);
            for (uint32_t k = 16; k < total; k += 16)
            {
              const uint32_t t = e0; e0 = e1; e1 = e2; e2 = t;
              w = u32x4{ e0, e1, e2, e0 };
              lds_st128(row + (HS_ROWWRAP(d0 + k) ^ tsw), w);
            }
            acc = w;
          }
          phase = (phase + m) % (uint32_t)S;
        }
        run -= m;
        o += m;
      }
    }
    done = (fl & F_DONE) != 0u;
    last = (fl & F_LAST) != 0u;
    }

    HS_STAMP(tDecode)
    wave_sync();                                                   // every lane is done reading the ring and writing its row
    if (pass != 0) break;
    if constexpr (CAP != 0u)
    {
      // capped rounds: lanes that have not reached their target are the rule, not a sign of starvation: one pass, then the top-up (see below for the lanes that
      // stopped in the middle of a literal stretch)
      topup();
      wave_sync();
      break;
    }

    // ---- top-up: the loads issued one step ago have had the whole decode to arrive; they move into the ring first (this wait
    //      never includes the flush stores below: vmcnt counts loads and stores in order), then the next piece is requested --
    //      BEFORE the flush stores, so that the prefetch registers are live across the flush (the flush data then cannot share
    //      registers with them, which would force a full `s_waitcnt vmcnt(0)` drain of the stores before every top-up) ----
    topup();
    HS_STAMP(tLand)
    wave_sync();
    if (__ballot(!done && o < target) == 0ull) break;                // nobody starved (the common case)
    }
    }

    // Capped rounds: a lane that stopped in the middle of a LITERAL stretch because the ring ran dry (literal-heavy data uses up the ring within a pass) gets
    // one more pass on the topped-up ring before the flush -- without it such lanes make half a row per round (random bytes: -27 %).  As a second trip of the
    // round loop, not as pass 1 of the loop above: that form keeps two decode passes' state alive across the top-up (rle48_7symlut: 166 -> 188 VGPRs, a wave
    // per SIMD less, -13 ... -18 %).  A lane that only waits for a header's bytes does not ask for it (64-byte rings do that all the time).
    if constexpr (CAP != 0u)
    {
#ifndef HSRLE_DEC_NO_SECOND_PASS   // (A/B builds)
      if (!skippedFlush && (uint32_t)__builtin_popcountll(__ballot(starved && !done)) >= HSRLE_DEC_STARVED_MIN) { skippedFlush = true; continue; }
      skippedFlush = false;
#endif
    }

    // ---- flush: whole 16-byte chunks only.  A row that ends inside a chunk (lane starved, or the block tail) keeps that
    //      chunk in `acc` and re-writes it at tile offset 0 in the next round, so there is no byte-granular path ----
    const uint32_t produced = o - base;
    const bool tailNow = active && o == blen && (produced & 15u) != 0u;   // only the last block of a buffer can have one
    // (capped rounds: whole 64-byte halves; everything at the block's end)
    const uint32_t chunks = !active ? 0u : ((CAP != 0u && o != blen) ? ((produced >> 6) << 2) : (produced >> 4));
#ifdef HSRLE_STAMPS
    if (active && produced != (uint32_t)T) tIssue += 1ull << 40;          // diagnostic: partial rows
#endif
    uint32_t fi[CPR];
#ifdef HSRLE_STAMPS
    tq0 = __builtin_readcyclecounter();
#endif
    HS_EXCHANGE(fi, chunks != 0u ? (base | chunks) : 0u, CPR)          // base is a multiple of 16 while chunks are pending (after the tail bytes it is not), chunks <= 8
    HS_XSTAMP(tx3)
#pragma unroll
    for (int h = 0; h < CPR; h += FH)                                  // FH row-groups at a time bounds the live registers
    {
      u32x4 fv[FH];
      uint32_t fAt[FH];
#pragma unroll
      for (int k = 0; k < FH; k++)
      {
        const int q = h + k;
        const uint32_t r = (uint32_t)q * RPI + lane / CPR, c = lane % CPR;
        if constexpr (CAP != 0u) fv[k] = lds_ld128(tile + r * TS + ((((fi[q] & ~15u) + c * 16u) & ((uint32_t)T - 1u)) ^ tsw_of(r)));
        else fv[k] = lds_ld128(tile + r * TS + ((c * 16u) ^ tsw_of(r)));
        fAt[k] = (c < (fi[q] & 15u)) ? (fi[q] & ~15u) + c * 16u : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < FH; k++)
      {
        const int q = h + k;
#if defined(HSRLE_ABLATE_STORES) && HSRLE_ABLATE_STORES == 2  // timing-only diagnostic build: no store instruction is ever executed (output is wrong)
        if (U == 0x7FFFFFFFFFFFFFF1ull)
#elif defined(HSRLE_ABLATE_STORES)  // timing-only diagnostic build: only one lane in 64 stores (output is wrong)
        if (lane == 0)
#endif
        if (fAt[k] != 0xFFFFFFFFu)
          // streaming store: the output is written once and never read here, keep it from evicting the compressed lines in L2
          __builtin_nontemporal_store(fv[k], (u32x4_unaligned *)(out + (uint64_t)(wgFirst + (uint32_t)q * RPI + lane / CPR) * B + fAt[k]));
      }
    }
    HS_XSTAMP(tx4)
    base += chunks << 4;
    if (tailNow)
    {
      const uint32_t at = CAP != 0u ? (base & ((uint32_t)T - 1u)) : (chunks << 4);   // (capped rounds: base is already behind the flushed chunks)
      for (uint32_t k = 0; k < (produced & 15u); k++)
        out[(uint64_t)b * B + base + k] = row[(at ^ tsw) + k];
      base = o;                                                        // written exactly once
    }

    wave_sync();
    HS_STAMP(tFlush)
  }

#ifdef HSRLE_STAMPS
  // diagnostic build only (never shipped): per-phase cycle sums of every workgroup's lane 0, appended behind the status word
  if (status != nullptr)
  {
    unsigned long long *dbg = (unsigned long long *)(status + 16);
    const unsigned long long iters = __builtin_amdgcn_readfirstlane((uint32_t)nIter);
    if (lane == 0)
    {
      atomicAdd(dbg + 0, tIssue); atomicAdd(dbg + 1, tDecode); atomicAdd(dbg + 2, tFlush); atomicAdd(dbg + 3, tLand);
      atomicAdd(dbg + 4, nRounds); atomicAdd(dbg + 5, iters); atomicAdd(dbg + 6, 1ull);
      atomicAdd(dbg + 12, tx0); atomicAdd(dbg + 13, tx1); atomicAdd(dbg + 14, tx2); atomicAdd(dbg + 15, tx3); atomicAdd(dbg + 16, tx4);
    }
  }
#endif

#ifdef HSRLE_REQ_STATS
  if (status != nullptr && lane == 0)
  {
    unsigned long long *dbg = (unsigned long long *)(status + 16);
    atomicAdd(dbg + 0, nFullLine); atomicAdd(dbg + 1, nHalfLine); atomicAdd(dbg + 2, nBothHalves); atomicAdd(dbg + 3, nOneHalf);
  }
#endif

  if (active && o != blen)
    err |= DEC_ERR_STREAM;

  if (err != 0 && status != nullptr)
    atomicOr(status, err);
#undef single
}

} // namespace hsrle
