// hsrle_parse.hip.h -- ONE statement of the packet header grammar of every codec (reference: src/rle8_extreme_cpu.h:1849-1899,
// src/rleX_extreme_cpu_decode.h:129-162, src/rleX_Xsl.h:580-760, src/rleX_Xsl_short.h:560-700, src/rle128_extreme_cpu.h:600-802), on a
// 24-byte window of the stream held in registers.  Used by the walks of the index / packet list (hsrle_index.hip.h) and -- round 4 -- by the
// per-lane loop of the block decoder for symbols of 2 .. 8 bytes (hsrle_decode.hip.h).
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

// symbol-state slots a decoder of this family carries from packet to packet
template <int FAM>
struct IndexState
{
  static constexpr int KE = (FAM == PACKED) ? 1 : Traits<FAM, 1, 0>::K;
};

struct Pkt
{
  uint32_t used;     // header bytes
  uint32_t lit;      // literal bytes that follow the header (they come BEFORE the run in the output)
  uint32_t run;      // run bytes
  uint32_t op;       // state slots: < KE move slot op to the front, == KE push the symbol at symAt
  uint32_t symAt;    // stream offset of the symbol this packet carries
  bool hasSym, last, bad;
};

// 32 bits at byte offset pos (0..20) of the 24-byte little-endian value ex:hi:lo
__device__ __forceinline__ uint32_t ex32x(uint64_t lo, uint64_t hi, uint64_t ex, uint32_t pos)
{
  const uint32_t sh = pos * 8u, s = sh & 63u;
  const uint64_t w = sh < 64u ? lo : (sh < 128u ? hi : ex), wn = sh < 64u ? hi : (sh < 128u ? ex : 0ull);
  return (uint32_t)((w >> s) | ((wn << 1) << (63u - s)));
}

// One packet header at stream offset p (the stream is readable up to C + 32).  The field rules are those of the decoder
// (hsrle_decode.hip.h): the two must agree on every stream, which tests/test_gpu_mono.py checks by decoding through
// the index what the block kernel decodes on its own.
// 32 bits at the compile-time byte offset P (<= 20) of a 24-byte window given as six dwords: one v_alignbyte at most.  The parse below
// extracts every field at EVERY position it can have (there are two to six) and selects: a handful of independent full-rate instructions
// instead of a chain of position arithmetic and 64-bit shifts -- a hop of a walk is one wave's dependent chain, its depth is its time.
template <int P>
__device__ __forceinline__ uint32_t u32c(const uint32_t (&d)[6])
{
  static_assert(P >= 0 && P <= 20, "inside the 24-byte window");
  if constexpr ((P & 3) == 0) return d[P >> 2];
  else return alignbyte(d[(P >> 2) + 1], d[P >> 2], (uint32_t)(P & 3));
}

template <int FAM, int S, int AL, typename READER>
__device__ __forceinline__ Pkt parse_window(const READER &rd, uint64_t lo, uint64_t hi, uint64_t ex, uint32_t p, uint32_t C, bool single);

// does parse_window read header bytes at or beyond byte 16 of its window?  (Short: never -- fields end at byte 10; LUT: extension fields up to
// 6 + S + 3; Packed: the range field up to 5 + S + 3, sym-aligned + 4; plain: up to S + 5 + 4 + 3)
template <int FAM, int S, int AL>
__host__ __device__ constexpr bool header_beyond_16()
{
  using TR = Traits<FAM, S, AL>;
  if (S == 16) return true;
  if (TR::kShort) return false;
  if (TR::kLut) return 6 + S + 3 >= 16;
  if (TR::kPacked) return (TR::kRange7 ? 5 + S + 3 : 5 + S + 7) >= 16;    // (sym-aligned Packed: the 32-bit range behind a zero range byte)
  return S + 5 + 4 + 3 >= 16;
}

// the same with the 24 header bytes given (walks that load the next packet's window while they book the current packet)
template <int FAM, int S, int AL, typename READER>
__device__ __forceinline__ Pkt parse_window(const READER &rd, uint64_t lo, uint64_t hi, uint64_t ex, uint32_t p, uint32_t C, bool single)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int KE = IndexState<FAM>::KE;
  Pkt k;
  k.used = 1; k.lit = 0; k.run = 0; k.op = 0; k.symAt = p; k.hasSym = false; k.last = false; k.bad = false;
  if (p + 2u > C) { k.bad = true; return k; }

  // 24 header bytes in registers: every field of every header form of symbols up to 8 bytes lies below byte 20, so the parse needs no
  // second read and -- written with selects -- no branch (a hop of the walks is a chain of ~200 dependent instructions in ONE wave per
  // SIMD: exec-mask round trips cost more than the instructions they skip).  128-bit symbols keep a reload for fields beyond byte 20.
  auto u32at = [&](uint32_t pos) -> uint32_t {
    if constexpr (S == 16) return pos <= 20u ? ex32x(lo, hi, ex, pos) : rd.load32(p + pos);
    else return ex32x(lo, hi, ex, pos);
  };

  uint32_t cnt, range, pos;
  bool endNow = false, hbad = false;

  if constexpr (TR::kShort)
  {
    const uint32_t p1 = (uint32_t)lo & 0xFFu, p2 = ((uint32_t)lo >> 8) & 0xFFu, p3 = ((uint32_t)lo >> 16) & 0xFFu;
    const uint32_t idx = (TR::K > 0) ? p1 >> (TR::SCB + TR::SRBP) : 0u;
    const uint32_t c3 = (p1 >> TR::SRBP) & TR::SCINV;
    const bool lf = c3 == TR::SCINV;                                     // the three-byte form (+ extensions)
    uint32_t cntL = (p2 >> (TR::SRB - 8u)) | ((p1 & TR::SMAXPR) << (16u - TR::SRB));
    uint32_t rngL = p3 | ((p2 & ((1u << (TR::SRB - 8u)) - 1u)) << 8);
    const uint32_t cw = u32at(3u);
    const bool c0 = lf && cntL == 0u, c1 = lf && cntL == 1u;
    cntL = c0 ? cw : (c1 ? (cw & 0xFFFFu) : cntL);
    uint32_t posL = c0 ? 7u : (c1 ? 5u : 3u);
    const uint32_t rext = u32at(posL);
    const bool r0 = lf && rngL == 0u, r1 = lf && rngL == 1u;
    rngL = r0 ? rext : (r1 ? (rext & 0xFFFFu) : rngL);
    posL += r0 ? 4u : (r1 ? 2u : 0u);
    endNow = r1 && rngL == 0u;
    cnt = lf ? cntL : c3 + 2u;
    range = lf ? rngL : (p1 & TR::SMAXPR) + 2u;
    pos = lf ? posL : 1u;
    if constexpr (TR::kShortSingle) { }
    else if constexpr (TR::K == 0)
    {
      // every packet carries its symbol; the END terminator carries ONE zero byte whatever the symbol width (rleX_Xsl_short.h:497-520)
      k.hasSym = true; k.symAt = p + pos; pos += endNow ? 1u : (uint32_t)S;
    }
    else
    {
      k.op = idx;
      const bool hs = idx == (uint32_t)TR::K;
      k.hasSym = hs; k.symAt = hs ? p + pos : p; pos += hs ? (uint32_t)S : 0u;
    }
    hbad = !endNow && range < 2u;
    k.last = endNow || cnt == 0u;
    k.lit = (endNow || range < 2u) ? 0u : range - 2u;
    k.run = k.last ? 0u : (TR::kAligned ? (cnt + TR::SMINS / (uint32_t)S - 2u) * (uint32_t)S : cnt + TR::SMINS - 2u);
  }
  else if constexpr (TR::kLut)
  {
    static_assert(S <= 8, "LUT codecs have symbols of up to 8 bytes");
    const uint32_t d[6] = { (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)ex, (uint32_t)(ex >> 32) };
    const uint32_t w16 = d[0] & 0xFFFFu;
    const uint32_t idx = w16 >> (FAM == LUT3 ? 14 : 13);
    const uint32_t c7 = (w16 >> TR::RB) & 0x7Fu, r7 = w16 & ((1u << TR::RB) - 1u);
    k.op = idx;
    const bool hs = idx == (uint32_t)TR::K;
    k.hasSym = hs; k.symAt = hs ? p + 2u : p;
    // extension fields sit at 2 (+ S with a symbol) (+ 2 / 4 behind a 16 / 32 bit count)
    const uint32_t e0 = hs ? u32c<2 + S>(d) : u32c<2>(d), e2 = hs ? u32c<4 + S>(d) : u32c<4>(d), e4 = hs ? u32c<6 + S>(d) : u32c<6>(d);
    const bool c0 = c7 == 0u, c1 = c7 == 1u;
    cnt = c0 ? e0 : (c1 ? (e0 & 0xFFFFu) : c7);
    const uint32_t rw = c0 ? e4 : (c1 ? e2 : e0);
    const bool r0 = r7 == 0u, r1 = r7 == 1u;
    range = r0 ? rw : (r1 ? (rw & 0xFFFFu) : r7);
    pos = (hs ? 2u + (uint32_t)S : 2u) + (c0 ? 4u : (c1 ? 2u : 0u)) + (r0 ? 4u : (r1 ? 2u : 0u));
    endNow = r1 && range == 0u;
    hbad = !endNow && range < 2u;
    k.last = endNow || cnt == 0u;
    k.lit = (endNow || range < 2u) ? 0u : range - 2u;
    k.run = k.last ? 0u : (TR::kAligned ? (cnt + 3u / (uint32_t)S - 2u) * (uint32_t)S : cnt + 1u);
  }
  else
  {
    const bool sgl = (S == 1) && single;
    [[maybe_unused]] uint32_t wPacked = 0;
    if constexpr (!TR::kPacked)
    {
      pos = sgl ? 0u : (uint32_t)S;
      k.hasSym = !sgl;
      const uint32_t cw = u32at(pos);                                  // count byte, and (if it is 0) the 32-bit count behind it at pos + 1
      const uint32_t cx = (S == 16) ? u32at(pos + 1u) : (uint32_t)((cw >> 8) | (u32at(pos + 4u) << 24));
      cnt = cw & 0xFFu;
      const bool c0 = cnt == 0u;
      cnt = c0 ? cx : cnt;
      pos += c0 ? 5u : 1u;
    }
    else
    {
      const uint32_t b0 = (uint32_t)lo & 0xFFu;
      cnt = sgl ? b0 : (b0 & 0x7Fu);
      const bool c0 = cnt == 0u;
      cnt = c0 ? u32at(1u) : cnt;
      pos = c0 ? 5u : 1u;
      const bool hs = !sgl && !(b0 & 0x80u);
      k.hasSym = hs; k.symAt = hs ? p + pos : p; k.op = hs ? 1u : 0u;
      pos += hs ? (uint32_t)S : 0u;
      if constexpr (S <= 8)
      {
        // the range field at 1 / 5 (+ S with a symbol): all four, then select
        const uint32_t d[6] = { (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)ex, (uint32_t)(ex >> 32) };
        wPacked = hs ? (c0 ? u32c<5 + S>(d) : u32c<1 + S>(d)) : (c0 ? u32c<5>(d) : u32c<1>(d));
      }
    }
    const uint32_t w = (TR::kPacked && S <= 8) ? wPacked : u32at(pos);
    const uint32_t r0 = w & 0xFFu;
    if (TR::kRange7 && !sgl)
    {
      const bool lng = (r0 & 1u) != 0u;
      range = lng ? (w >> 1) : (r0 >> 1);
      pos += lng ? 4u : 1u;
      endNow = lng && range == 0u;
    }
    else
    {
      const bool lng = r0 == 0u;
      const uint32_t rx = (S == 16) ? (lng ? u32at(pos + 1u) : 0u) : (uint32_t)((w >> 8) | (u32at(pos + 4u) << 24));
      range = lng ? rx : r0;
      pos += lng ? 5u : 1u;
      endNow = lng && range == 0u;
    }
    const uint32_t shortv = sgl ? (TR::kPacked ? 2u : 4u) : TR::SHORT;
    k.last = endNow || cnt == 0u;
    k.lit = (range == 0u || endNow) ? 0u : range - 1u;            // a 7 bit range byte of 0x00 carries no literals (A.5 q11)
    k.run = k.last ? 0u : (TR::kAligned ? (cnt + TR::SHORT / (uint32_t)S - 1u) * (uint32_t)S : cnt + shortv - 1u);
  }
  (void)KE;
  k.used = pos;
  const uint32_t sp = p + pos;
  k.bad = hbad || sp > C || k.lit > C - sp || (k.lit == 0u && k.run == 0u && !k.last);
  return k;
}

} // namespace hsrle
