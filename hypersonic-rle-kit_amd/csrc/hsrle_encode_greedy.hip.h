// hsrle_encode_greedy.hip.h -- the Greedy encoders of the byte-aligned 1/3/7 symbol LUT Short codecs
// (rle{16,24,32,48,64}_{1,3,7}symlut_byte_short_compress_greedy; reference: src/rle.h:398-416).
//
// Replaces: src/rleX_Xsl_short.h:746-1000 (the greedy scan) + :152-372 (process_symbol of the Short family).
//
// Besides repeats of the symbol at hand the greedy scan tries the symbols of the move-to-front list -- and, for symbols wider
// than 16 bit, their leading bytes -- as runs, so a 2-byte "run" of a listed symbol's first bytes can be stored in one header
// byte.  The scan is a byte-granular state machine whose next position depends on the list (which changes with every stored
// run), so it does not map onto the window-parallel run enumeration of k_encodeS_blocks; it runs here as what it is, one lane
// per block: symbols through the LaneRing (hsrle_common.hip.h), output through the per-lane Sink of hsrle_encode.hip.h.  The streams are
// Short streams: they decode with k_decode_blocks<SHORT1/3/7, S, 0> (src/codec_funcs.h:298-388 pairs them the same way).
// Bytes at or beyond the block end never match.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_encode.hip.h" // Sink, load_sym

namespace hsrle {

// MONO (round 4): chunks of ONE monolithic stream (hsrle_mono_encode.hip.h).  The greedy scan, too, stores a run of SMINL bytes or more whatever
// its state is, and it extends a run to the exact end of its periodic stretch; what it cannot know behind such a run is the LIST (which
// rotation of the symbol went to the front depends on where the scan entered the stretch) -- the lists in front of the chunks are guessed,
// checked and repaired like those of the other list codecs (monoSyms: 8 words per chunk, entry k = lut0 | lut1 << 32, word 7 = "encode this
// chunk in this pass"; monoListOut: the list behind the chunk, word 7 = how many of its entries the chunk itself determined; monoDry: no
// output, only the list).  A chunk ends with the packet of its boundary run (no terminator, no header: k_mono_finish writes the header);
// the look-ahead of the scan sees the true input behind the chunk.
template <int FAM, int S, bool MONO = false>
__global__ __launch_bounds__(64) void k_encode_greedy_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                             uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes,
                                                             const uint64_t *__restrict__ monoStarts, const uint64_t *monoSyms, const uint64_t *__restrict__ monoSlotOff,
                                                             uint32_t monoSteps, uint64_t *monoListOut, uint32_t monoDry, const uint32_t *__restrict__ ringSel)
{
  using TR = Traits<FAM, S, 0>;
  static_assert(TR::kShort && TR::K > 0 && S >= 2 && S <= 8, "1/3/7 symbol LUT Short codecs of the 16..64 bit symbols");
  constexpr int K = TR::K;
  constexpr uint32_t SU = (uint32_t)S;
  // MONO with B != 0 ("split encode" of a container with small blocks, hsrle_capi.hip compress_split): the chunks are pieces of the container's
  // BLOCKS -- a chunk that opens its block writes the block stream's header, the block's end is the end of the input, ringSel[0] = the
  // number of chunks, and a repair round (monoSteps >> 16) that has nothing to do returns at once (ringSel[7 + round]: how many chunks it has)
  const bool blocks = MONO && B != 0u;
  if constexpr (MONO) { if (blocks && ringSel != nullptr && (monoSteps >> 16) != 0u && ringSel[7u + (monoSteps >> 16)] == 0u) return; }

  __shared__ __attribute__((aligned(16))) uint8_t ringMem[64 * kLaneRingStride];
  const uint32_t b = MONO ? blockIdx.x * 64u + threadIdx.x : xcd_tile(blockIdx.x, gridDim.x) * 64u + threadIdx.x;
  // (no early return in MONO mode before the activity test: the ring's top-ups are wave-wide ballots, but lanes that are out simply do not vote)
  if (b >= nBlocks)
    return;
  if constexpr (MONO) { if (blocks && ringSel != nullptr && b >= ringSel[0]) return; }
  if constexpr (MONO) { if (ld_fresh64(monoSyms + 8ull * b + 7) == 0ull) return; }   // repair rounds switch most chunks off
#ifdef HSRLE_GREEDY_DRY   // timing-only diagnostic build (never shipped): the scan without its output stores (sizes are still written, the payload is not)
  const bool dry = true;
#else
  const bool dry = MONO && monoDry != 0u;
#endif

  const uint64_t start = MONO ? monoStarts[b] : (uint64_t)b * B;
  if constexpr (MONO)
  {
    if (blocks) { const uint64_t blockEnd = (start / B + 1ull) * B; if (blockEnd < U) U = blockEnd; }   // (U: from here on the end of this chunk's world)
  }
  // n: where this lane's scan ends (block / chunk length); nT: the input the scan may LOOK at (block: the same; chunk: up to the true end)
  const uint32_t n = MONO ? (uint32_t)(monoStarts[b + 1u] - start) : (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  const uint32_t nT = MONO ? (uint32_t)((U - start) < 0xFFFFFFFFull ? (U - start) : 0xFFFFFFFFull) : n;
  const bool lastChunk = !MONO || start + n >= U;
  const uint8_t *const d = in + start;
  // the scan reads its symbols through the LaneRing (topped up by all lanes together whenever one of them runs low); reading behind
  // the block (inside the input) is harmless, sym_at masks what lies beyond nT
  LaneRing ring{ d, (uint32_t)((U - start) < 0xFFFFFFFFull ? (U - start) : 0xFFFFFFFFull), ringMem + threadIdx.x * kLaneRingStride, 0u };
  auto ensure = [&](uint32_t i) {
    if (__builtin_amdgcn_ballot_w64(i + 64u > ring.loadedEnd) != 0ull) ring.template topup<8>(i);
  };
  // output: the block's slot, or the chunk's place in the staging area; a dry pass only counts
  struct Out
  {
    Sink s; bool dry;
    __device__ __forceinline__ void put8(uint32_t v) { if (!dry) s.put8(v); else s.at += 1; }
    __device__ __forceinline__ void put16(uint32_t v) { if (!dry) s.put16(v); else s.at += 2; }
    __device__ __forceinline__ void put32(uint32_t v) { if (!dry) s.put32(v); else s.at += 4; }
    __device__ __forceinline__ void putn(const uint8_t *src, uint32_t k) { if (!dry) s.putn(src, k); else s.at += k; }
    __device__ __forceinline__ void put_sym(u32x4 v) { if (!dry) s.template put_sym<S>(v); else s.at += (uint32_t)S; }
  };
  Out s{ Sink{ MONO ? slots + monoSlotOff[b] : slots + (uint64_t)b * slotStride, 0u, in + U }, dry };
  if constexpr (!MONO)
  {
    s.put32(n);
    s.put32(0);
  }
  else if (blocks && (start % B) == 0ull)
  {
    s.put32((uint32_t)(U - start));                                      // the block's uncompressed length; compressedLength: k_split_finish
    s.put32(0);
  }

  // move-to-front list, entry k = {lut0[k], lut1[k]} (low S bytes); rleX_Xsl_short.h:759-774
  uint32_t lut0[K], lut1[K];
  [[maybe_unused]] uint32_t mtfDepth = 0;
  if constexpr (MONO)
  {
#pragma unroll
    for (int k = 0; k < K; k++) { const uint64_t v = ld_fresh64(monoSyms + 8ull * b + k); lut0[k] = (uint32_t)v; lut1[k] = (uint32_t)(v >> 32); }
  }
  else
  {
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const uint32_t b4 = init[k] * 0x01010101u;
      lut0[k] = (S >= 4) ? b4 : (b4 & ((1u << (8 * (S & 3))) - 1u));
      lut1[k] = (S == 8) ? b4 : (S == 6 ? (b4 & 0xFFFFu) : 0u);
    }
  }

  uint32_t lastRLE = 0;

  // the symbol at position i, zero extended; bytes at or beyond nT read as zero
  auto sym_at = [&](uint32_t i, uint32_t &s0, uint32_t &s1) {
    if constexpr (S > 4) ring.get64(i, s0, s1); else { s0 = ring.get32(i); s1 = 0u; }
    const uint32_t have = (nT - i < SU) ? nT - i : SU;                   // i < nT
    const uint64_t keep = (have >= 8u) ? ~0ull : ~(~0ull << (8u * have));
    s0 &= (uint32_t)keep; s1 &= (uint32_t)(keep >> 32);
  };
  // number of equal leading bytes of two symbols (S when they are equal)
  auto prefix = [&](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) -> uint32_t {
    const uint64_t diff = (uint64_t)(a0 ^ b0) | ((uint64_t)(a1 ^ b1) << 32);
    return diff == 0ull ? SU : (uint32_t)__builtin_ctzll(diff) >> 3;
  };

  // process_symbol (rleX_Xsl_short.h:152-372) for the run [i - count, i) of {s0, s1}
  auto process = [&](uint32_t s0, uint32_t s1, uint32_t count, uint32_t i) -> bool {
    const uint32_t gap = i - lastRLE - count;
    const uint32_t range = gap + 2u;
    uint32_t m = (uint32_t)K;
#pragma unroll
    for (int k = K - 1; k >= 0; k--)
      if (lut0[k] == s0 && lut1[k] == s1) m = (uint32_t)k;
    const int32_t sc = (int32_t)count - (int32_t)TR::SMINS + 2;
    const bool pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
    uint32_t pen = (m == (uint32_t)K) ? SU : 0u;
    if (!pack1)
    {
      pen += 2u;
      if (!(sc <= (int32_t)TR::SMAXTC && range <= TR::SMAXTR))
        pen += ((range <= 0xFFFFFu) ? (range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
    }
    if (!(count >= TR::SMINL || count >= TR::SMINS + pen))
      return false;

    const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
    for (int k = K - 1; k >= 1; k--)
      if ((uint32_t)k <= limit) { lut0[k] = lut0[k - 1]; lut1[k] = lut1[k - 1]; }
    lut0[0] = s0; lut1[0] = s1;
    if (m >= mtfDepth && mtfDepth < (uint32_t)K) mtfDepth++;             // (as the ring encoders count it: hsrle_encodeS.hip.h)

    const uint32_t mi = m << (TR::SCB + TR::SRBP);
    if (pack1)
      s.put8(mi | ((uint32_t)(sc - 2) << TR::SRBP) | gap);
    else
    {
      const uint32_t scu = (uint32_t)sc;
      const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
      const uint32_t rx = (range <= TR::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
      s.put8((mi | (TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8)) & 0xFFu);
      s.put8(((scx << (TR::SRB - 8u)) | (rx >> 8)) & 0xFFu);
      s.put8(rx & 0xFFu);
      if (scx != scu) { if (scu <= 0xFFFFu) s.put16(scu); else s.put32(scu); }
      if (rx != range) { if (range <= 0xFFFFu) s.put16(range); else s.put32(range); }
    }
    if (m == (uint32_t)K) s.put_sym(u32x4{ s0, s1, 0u, 0u });
    s.putn(d + lastRLE, gap);
    lastRLE = i;
    return true;
  };

  // ---- the greedy scan (rleX_Xsl_short.h:783-974) ----
  uint32_t y0, y1;                                                     // state.symbol: starts as the complement of the first symbol
  ensure(0u);
  sym_at(0, y0, y1);
  y0 = ~y0; y1 = ~y1;
  if constexpr (S < 4) y0 &= (1u << (8 * S)) - 1u;
  if constexpr (S <= 4) y1 = 0u; else if constexpr (S == 6) y1 &= 0xFFFFu;
  uint32_t count = 0, i = 0;
  bool stopped = false;                                                // MONO: the boundary run's packet is out

  // (a chunk that is not the stream's last goes on until its boundary run's packet is out: the run may reach n in whole symbols, and the scan
  //  only sees that it is over when it looks at position n)
  while ((MONO && !lastChunk) ? !stopped : i < n)
  {
    ensure(i);
    // inside a run: W = 8 (6 for 3 / 6 byte symbols) bytes against the symbol's pattern per trip -- what W / S of the reference's steps do (each compares one
    // symbol and stops at the first that differs, taking its matching leading bytes along: the first differing BYTE says both; rleX_Xsl_short.h:800-860).  One
    // step per trip of the ONE loop, not a loop of its own: a phase that only the lanes inside a run execute makes the others wait (round 4's literal skip).
    constexpr uint32_t W = (S == 3 || S == 6) ? 6u : 8u;
    if (S < 8 && count != 0u && i + W <= nT)
    {
      uint32_t x0, x1, p0, p1;
      ring.get64(i, x0, x1);
      if constexpr (S == 2) { p0 = y0 * 0x00010001u; p1 = p0; }
      else if constexpr (S == 3) { p0 = y0 | (y0 << 24); p1 = y0 >> 8; }
      else if constexpr (S == 4) { p0 = y0; p1 = y0; }
      else { p0 = y0; p1 = y1; }                                         // S == 6: y1 holds bytes 4, 5
      const uint64_t diff = ((uint64_t)(x0 ^ p0) | ((uint64_t)(x1 ^ p1) << 32)) & (W == 8u ? ~0ull : 0xFFFFFFFFFFFFull);
      if (diff == 0ull) { count += W; i += W; continue; }
      const uint32_t e = (uint32_t)__builtin_ctzll(diff) >> 3;           // bytes that still belong to the run: whole symbols, then the leading bytes of the one that differs
      count += e; i += e;
    }
    else if (count != 0u && i + SU <= nT)
    {
      uint32_t x0, x1;
      sym_at(i, x0, x1);
      const uint32_t j = prefix(y0, y1, x0, x1);
      if (j == SU) { count += SU; i += SU; continue; }
      if constexpr (S == 2) { if (j != 0u) { count += 1u; i += 1u; } }
      else { count += j; i += j; }
    }

    for (;;)                                                           // label not_a_full_match_but_a_match (:861)
    {
      ensure(i);
      if (count >= TR::SMINS) process(y0, y1, count, i);   // a shorter "run" (0 bytes between two literal positions) is never stored and changes no state
      if (MONO && !lastChunk && i >= n) { stopped = true; break; }      // the chunk ends behind its boundary run: the next chunk takes the scan up at i
      sym_at(i, y0, y1);
      const bool fits = i + SU <= nT;

      if (fits && i + 2u * SU <= nT)
      {
        uint32_t z0, z1;
        sym_at(i + SU, z0, z1);
        if (z0 == y0 && z1 == y1) { count = 2u * SU; i += 2u * SU; break; }
      }
      if (!fits) { count = 0u; i += 1u; break; }

      uint32_t pc = 0u, idx = 0u;
      bool full = false;
      // a listed symbol counts from two matching leading bytes on (SMINS): where no lane of the wave has an entry whose first 16 bits are the data's, nobody
      // needs the lengths (literal stretches of data the list does not fit: 99 % of the trips, half of a trip's instructions)
      bool cand = false;
#pragma unroll
      for (int k = 0; k < K; k++) cand = cand || ((lut0[k] ^ y0) & 0xFFFFu) == 0u;
      if (__ballot(cand) != 0ull)
      {
#pragma unroll
      for (int k = 0; k < K; k++)
        if (!full)
        {
          const uint32_t c = prefix(lut0[k], lut1[k], y0, y1);
          if (c == SU) { idx = (uint32_t)k; pc = SU; full = true; }
          else if (S != 2 && c > pc) { idx = (uint32_t)k; pc = c; }
        }
      }

      if (S != 2 ? pc >= TR::SMINS : pc != 0u)
      {
        count = pc; i += pc;
#pragma unroll
        for (int k = 0; k < K; k++)
          if (idx == (uint32_t)k) { y0 = lut0[k]; y1 = lut1[k]; }
        if (S != 2 && count < SU) continue;                            // goto not_a_full_match_but_a_match
        break;
      }
      count = 0u; i += 1u;
      break;
    }
  }

  if (lastChunk)
  {
    // ---- remaining bytes (rleX_Xsl_short.h:976-1032) ----
    if (process(y0, y1, count, i))
    {
      s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(1); s.put16(0); s.put16(0);
    }
    else
    {
      const uint32_t kLit = n - lastRLE;
      s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(0); s.put16(0); s.put32(kLit + 2u);
      s.putn(d + lastRLE, kLit);
    }
  }
  if constexpr (!MONO) s.s.patch32(4, s.s.at);
  // (a chunk that is not its stream's last must end exactly on its boundary: a scan that ran past it would write bytes the next chunk writes again.
  //  The cut rule -- stretches of >= 4 S + 11 bytes -- makes that impossible as far as anyone has seen; size 0 is how a chunk says "not me", and the
  //  callers check it: k_mono_zero_sizes / k_split_check)
  if (!dry || !MONO) sizes[b] = (MONO && !lastChunk && i != n) ? 0u : s.s.at;
  if constexpr (MONO)
  {
#pragma unroll
    for (int k = 0; k < K; k++) monoListOut[8ull * b + k] = (uint64_t)lut0[k] | ((uint64_t)lut1[k] << 32);
    monoListOut[8ull * b + 7] = mtfDepth;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// rle8_single_short (reference: src/rle.h:223-224; rleX_Xsl_short.h with SINGLE: wrapper :380-523, body :1058-1120), restated
// step for step on the same per-lane data path: the symbol is picked by the estimator of the extreme Single codec
// (single_pick_symbol), the body's skip loop passes over 16-byte windows with fewer than two occurrences of the symbol (unless
// the last byte is one), and the position it stops at when the windows run out is never examined (the for loop's own i++).
template <int FAM>   // SHORT_SINGLE (a template so that the header can be included by every instantiation unit)
__global__ __launch_bounds__(64) void k_encode_single_short_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                                   uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes)
{
  using TR = Traits<FAM, 1, 0>;
  static_assert(FAM == SHORT_SINGLE, "rle8_single_short only");
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x) * 64u + threadIdx.x;
  if (b >= nBlocks)
    return;

  const uint64_t start = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  const uint8_t *const d = in + start;
  Sink s{ slots + (uint64_t)b * slotStride, 0u, in + U };

  uint32_t prob[256], pc[256]; // per-lane histograms (private segment)
  const uint32_t sym = single_pick_symbol(d, n, prob, pc);
  const uint32_t bs = sym * 0x01010101u;
  s.put32(n);
  s.put32(0);
  s.put8(sym);

  uint32_t lastRLE = 0;

  // process_symbol with SINGLE (rleX_Xsl_short.h:152-372): no list, no symbol in the packet
  auto process = [&](int32_t count, uint32_t i) -> bool {
    const uint32_t gap = i - lastRLE - (uint32_t)count;
    const uint32_t range = gap + 2u;
    const int32_t sc = count - (int32_t)TR::SMINS + 2;
    const bool pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
    uint32_t pen = 0u;
    if (!pack1)
    {
      pen = 2u;
      if (!(sc <= (int32_t)TR::SMAXTC && range <= TR::SMAXTR))
        pen += ((range <= 0xFFFFFu) ? (range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
    }
    if (!(count >= (int32_t)TR::SMINL || count >= (int32_t)(TR::SMINS + pen)))
      return false;
    if (pack1)
      s.put8(((uint32_t)(sc - 2) << TR::SRBP) | gap);
    else
    {
      const uint32_t scu = (uint32_t)sc;
      const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
      const uint32_t rx = (range <= TR::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
      s.put8(((TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8)) & 0xFFu);
      s.put8(((scx << (TR::SRB - 8u)) | (rx >> 8)) & 0xFFu);
      s.put8(rx & 0xFFu);
      if (scx != scu) { if (scu <= 0xFFFFu) s.put16(scu); else s.put32(scu); }
      if (rx != range) { if (range <= 0xFFFFu) s.put16(range); else s.put32(range); }
    }
    s.putn(d + lastRLE, gap);
    lastRLE = i;
    return true;
  };

  int32_t count = 0;
  int64_t i = 0;
  const int64_t end = (int64_t)n - 16;

  for (; i < end; i++)                                                  // compress_single_sse2 (:1058-1120)
  {
    const Cmp16 c(d + i, bs);
    if (c.all()) { count += 16; i += 15; continue; }
    if (c.any() || count > 1)
    {
      const uint32_t z = c.leading();
      count += (int32_t)z;
      i += z;
      process(count, (uint32_t)i);
    }
    count = 0;
    bool streak = false;                                                // the last window was skipped: four at a time (one dependent load per window otherwise)
    while (i < end)
    {
      if (streak && i + 48 < end)
      {
        const Cmp16 b0(d + i, bs), b1(d + i + 16, bs), b2(d + i + 32, bs), b3(d + i + 48, bs);
        const bool s0 = !b0.any() || (!b0.lastByte() && b0.pop() < 2u), s1 = !b1.any() || (!b1.lastByte() && b1.pop() < 2u);
        const bool s2 = !b2.any() || (!b2.lastByte() && b2.pop() < 2u), s3 = !b3.any() || (!b3.lastByte() && b3.pop() < 2u);
        if (s0 && s1 && s2 && s3) { i += 64; continue; }
      }
      const Cmp16 w(d + i, bs);
      if (!w.any() || (!w.lastByte() && w.pop() < 2u))
      {
        i += 16;
        streak = true;
      }
      else
      {
        i += w.first();
        count = 1;
        break;
      }
    }
  }

  for (; i < (int64_t)n; i++)                                           // scalar tail (:452-466)
  {
    if (d[i] == sym)
      count++;
    else
    {
      process(count, (uint32_t)i);
      count = 0;
    }
  }

  if (process(count, (uint32_t)i))
  {
    s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(1); s.put16(0); s.put16(0);
  }
  else
  {
    const uint32_t kLit = n - lastRLE;
    s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(0); s.put16(0); s.put32(kLit + 2u);
    s.putn(d + lastRLE, kLit);
  }
  s.patch32(4, s.at);
  sizes[b] = s.at;
}

// One CHUNK of a monolithic rle8_single_short stream (round 4; as encode_chunk_single of the extreme Single codecs, hsrle_encode.hip.h): behind any
// stored run the body's state is its position alone, and a long run of the symbol is stored whatever the state -- the input is cut behind runs
// of >= SMINL + 16 bytes of the symbol (the margin: the body counts a run from where its search found it), a chunk is the body started at
// its first byte (not the symbol: the first trip falls through to the search), `end` is the TRUE end - 16, the scalar tail and the terminator
// belong to the last chunk, no stream header; a chunk in front of the last ends with its boundary run's packet.  Returns the chunk's
// stream bytes, 0 if the chunk did not end on a stored run at its boundary (the cut was wrong: the caller falls back to one lane).
template <int FAM>
__device__ inline uint32_t encode_chunk_single_short(const uint8_t *d, uint32_t nChunk, uint32_t nTrue32, uint32_t sym, Sink &s, const CopyJobs &jobs)
{
  using TR = Traits<FAM, 1, 0>;
  const uint32_t bs = sym * 0x01010101u;
  const bool finalChunk = nChunk == nTrue32;
  uint32_t lastRLE = 0;
  // long literal stretches are NOTED for k_copy_jobs, not copied by this one lane (as encode_chunk_single, hsrle_encode.hip.h: data its symbol is rare in
  // is all literals -- 64 MiB run-distributed in 4 KiB blocks: 709 us with the lanes copying)
  auto put_literals = [&](uint32_t from, uint32_t len) {
    if (len >= kCopyJobMin && jobs.list != nullptr)
    {
      const uint32_t k = atomicAdd(jobs.count, 1u);
      if (k < jobs.cap)
      {
        jobs.list[3ull * k] = jobs.srcBase + from;
        jobs.list[3ull * k + 1] = jobs.dstBase + s.at;
        jobs.list[3ull * k + 2] = len;
        s.at += len;
        return;
      }
    }
    s.putn(d + from, len);
  };
  auto process = [&](int32_t count, uint32_t i) -> bool {
    const uint32_t gap = i - lastRLE - (uint32_t)count;
    const uint32_t range = gap + 2u;
    const int32_t sc = count - (int32_t)TR::SMINS + 2;
    const bool pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
    uint32_t pen = 0u;
    if (!pack1)
    {
      pen = 2u;
      if (!(sc <= (int32_t)TR::SMAXTC && range <= TR::SMAXTR))
        pen += ((range <= 0xFFFFFu) ? (range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
    }
    if (!(count >= (int32_t)TR::SMINL || count >= (int32_t)(TR::SMINS + pen)))
      return false;
    if (pack1)
      s.put8(((uint32_t)(sc - 2) << TR::SRBP) | gap);
    else
    {
      const uint32_t scu = (uint32_t)sc;
      const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
      const uint32_t rx = (range <= TR::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
      s.put8(((TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8)) & 0xFFu);
      s.put8(((scx << (TR::SRB - 8u)) | (rx >> 8)) & 0xFFu);
      s.put8(rx & 0xFFu);
      if (scx != scu) { if (scu <= 0xFFFFu) s.put16(scu); else s.put32(scu); }
      if (rx != range) { if (range <= 0xFFFFu) s.put16(range); else s.put32(range); }
    }
    put_literals(lastRLE, gap);
    lastRLE = i;
    return true;
  };

  int32_t count = 0;
  int64_t i = 0;
  const int64_t n = (int64_t)nTrue32, stopAt = (int64_t)nChunk;
  const int64_t end = n - 16;

  for (; i < end; i++)                                                  // compress_single_sse2 (rleX_Xsl_short.h:1058-1120)
  {
    const Cmp16 c(d + i, bs);
    if (c.all()) { count += 16; i += 15; continue; }
    if (c.any() || count > 1)
    {
      const uint32_t z = c.leading();
      count += (int32_t)z;
      i += z;
      const bool stored = process(count, (uint32_t)i);
      if (!finalChunk && i >= stopAt)
        return (stored && i == stopAt) ? s.at : 0u;                     // the boundary run's packet ends the chunk
    }
    count = 0;
    bool streak = false;                                                // the last window was skipped: four at a time (one dependent load per window otherwise)
    while (i < end)
    {
      if (streak && i + 48 < end)
      {
        const Cmp16 b0(d + i, bs), b1(d + i + 16, bs), b2(d + i + 32, bs), b3(d + i + 48, bs);
        const bool s0 = !b0.any() || (!b0.lastByte() && b0.pop() < 2u), s1 = !b1.any() || (!b1.lastByte() && b1.pop() < 2u);
        const bool s2 = !b2.any() || (!b2.lastByte() && b2.pop() < 2u), s3 = !b3.any() || (!b3.lastByte() && b3.pop() < 2u);
        if (s0 && s1 && s2 && s3) { i += 64; continue; }
      }
      const Cmp16 w(d + i, bs);
      if (!w.any() || (!w.lastByte() && w.pop() < 2u))
      {
        i += 16;
        streak = true;
      }
      else
      {
        i += w.first();
        count = 1;
        break;
      }
    }
  }
  if (!finalChunk)
    return 0u;

  for (; i < n; i++)                                                    // scalar tail (:452-466)
  {
    if (d[i] == sym)
      count++;
    else
    {
      process(count, (uint32_t)i);
      count = 0;
    }
  }
  if (process(count, (uint32_t)i))
  {
    s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(1); s.put16(0); s.put16(0);
  }
  else
  {
    const uint32_t kLit = (uint32_t)n - lastRLE;
    s.put8(TR::SCINV << TR::SRBP); s.put8(TR::STB); s.put8(0); s.put16(0); s.put32(kLit + 2u);
    put_literals(lastRLE, kLit);
  }
  return s.at;
}

// B != 0 (round 4): chunks of the blocks of a container (split encode of small containers, as k_encode_single_chunks: the block's end is the end of
// the input, a block's first chunk writes the stream header and the symbol byte, pick = a symbol byte per block, chunkCount[0] = how many chunks there are)
template <int FAM>
__global__ __launch_bounds__(64) void k_encode_single_short_chunks(const uint8_t *__restrict__ in, uint64_t U, uint32_t chunks, const uint64_t *__restrict__ starts,
                                                                   const uint64_t *__restrict__ slotOff, uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes,
                                                                   const uint32_t *__restrict__ pick, uint64_t *__restrict__ jobList, uint32_t *__restrict__ jobCount, uint32_t jobCap,
                                                                   uint32_t B, const uint32_t *__restrict__ chunkCount)
{
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (B != 0u && chunkCount != nullptr) chunks = umin(chunks, chunkCount[0]);
  if (c >= chunks) return;
  const uint64_t start = starts[c];
  Sink s{ slots + slotOff[c], 0u, in + U };
  const CopyJobs jobs{ jobList, jobCount, jobCap, start, slotOff[c] };
  if (B == 0u)
  {
    sizes[c] = encode_chunk_single_short<FAM>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(U - start), pick[0] & 0xFFu, s, jobs);
    return;
  }
  const uint64_t blk = start / B, blockEnd = ((blk + 1ull) * B < U) ? (blk + 1ull) * B : U;
  const uint32_t sym = ((const uint8_t *)pick)[blk];
  if (start == blk * B) { s.put32((uint32_t)(blockEnd - start)); s.put32(0); s.put8(sym); }     // (rleX_Xsl_short.h:1211-1216: 8 bytes, then the symbol)
  sizes[c] = encode_chunk_single_short<FAM>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(blockEnd - start), sym, s, jobs);
}

} // namespace hsrle
