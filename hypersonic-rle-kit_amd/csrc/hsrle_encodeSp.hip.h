// hsrle_encodeSp.hip.h -- POSITION-PARALLEL encoder for the plain and Packed codecs of 2, 3, 4, 6 and 8 byte symbols (rle16 / 24 / 32 / 48 / 64
// _sym, _sym_packed, _byte, _byte_packed: 20 codecs) and the 3 symbol LUT codecs of 3, 4, 6 and 8 byte symbols (rle24 / 32 / 48 / 64 _3symlut_sym /
// _3symlut_byte: 8 codecs, src/rleX_Xsl.h:114-346 -- every run of theirs is stored, only the symbol's list index depends on what came before), blocks
// of at most 4 KiB.  The 8 bit kernel (hsrle_encode8p.hip.h) with the run discovery,
// the emit rule and the packet forms of the wide codecs; everything that file's header says about the division of labour holds here.
//
// Replaces: src/rleX_extreme_cpu_encode.h:14-609 (run discovery :315-371, extension :79-163, emit rule :174-311, terminators :384-603; the 24 / 48 bit
//           forms in src/rle24_extreme_cpu_encode.h, src/rle48_extreme_cpu_encode.h) -- and, in this library, the ring / run list encoders + staging
//           slots + k_compact for containers of these codecs (hsrle_encodeS.hip.h; the run list encoders of these widths were removed in round 6).
//
// Run discovery on match bits m[j] = (d[j] == d[j + S]) (the oracle's runs_next): a maximal stretch of L >= S set bits from position s
// means that the bytes [s, s + L + S) have period S.  The reference's scan starts a run at the first p >= (end of the run before) inside the stretch
// with S bits left, takes whole symbols while they repeat and -- byte-aligned codecs, while a whole symbol would still fit in front of the end of
// the input -- the matching leading bytes of the next one.  A stretch therefore gives at most one run, and its start depends on the run before it
// only where that run's tail (at most S bytes behind its stretch) reaches into this stretch: those few candidates wait for their left neighbour
// in the same DPP passes that settle the emit decisions.
#pragma once

#include "hsrle_encode8p.hip.h"

namespace hsrle {

// the S-byte symbol at LDS byte position `at` of the input image (dwords beyond S bytes zero)
template <int S>
__device__ __forceinline__ uint64_t pp_symbol(const uint8_t *inb, uint32_t at)
{
  const uint32_t *const w = (const uint32_t *)(inb + (at & ~3u));
  const uint32_t sb = at & 3u;
  const uint32_t q0 = w[0], q1 = w[1];
  uint32_t lo = alignbyte(q1, q0, sb), hi = 0u;
  if constexpr (S > 4) { const uint32_t q2 = w[2]; hi = alignbyte(q2, q1, sb); }
  if constexpr (S == 1) lo &= 0xFFu;
  if constexpr (S == 2) lo &= 0xFFFFu;
  if constexpr (S == 3) lo &= 0xFFFFFFu;
  if constexpr (S == 6) hi &= 0xFFFFu;
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// OR the low nb (1 .. 8) bytes of v into the stream image at byte position `at` (LDS atomics on the dwords it touches)
__device__ __forceinline__ void pp_or_bytes(uint8_t *img, uint32_t at, uint64_t v, uint32_t nb)
{
  if (nb < 8u) v &= (1ull << (8u * nb)) - 1ull;
  uint32_t *const wp = (uint32_t *)(img + (at & ~3u));
  const uint32_t sh = 8u * (at & 3u);
  const uint64_t lo = v << sh;
  const uint32_t top = sh ? (uint32_t)(v >> (64u - sh)) : 0u;          // what the shift pushed beyond 64 bits
  atomicOr(wp, (uint32_t)lo);
  if ((uint32_t)(lo >> 32) != 0u) atomicOr(wp + 1, (uint32_t)(lo >> 32));
  if (top != 0u) atomicOr(wp + 2, top);
}

template <int FAM, int S, int AL, int MODE>
__device__ __forceinline__ void ppS_block(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t b, uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets,
                                          uint8_t *__restrict__ payload, const PpScratch &sc, PpShared<MODE != 0, true, FAM == LUT3 || FAM == SHORT3> &sh, const u32x4 (&x)[4], uint32_t rec0)
{
  static_assert(FAM == PLAIN || FAM == PACKED || FAM == LUT3 || FAM == SHORT0 || FAM == SHORT1 || FAM == SHORT3, "plain, Packed, 3 symbol LUT, Short with no / a one- / a three-symbol list");
  // Short with the three-symbol list: a new symbol costs S bytes of penalty, so only from 6 byte symbols on is every run (>= 2 S bytes) stored whatever the list
  // says (2 + S + 2 + 2 <= 2 S; rleX_Xsl_short.h:152-215) -- the narrower ones have runs whose storing depends on the list
  static_assert(FAM != SHORT3 || S >= 6, "rle48 / rle64 only");
  static_assert(S == 1 || S == 2 || S == 3 || S == 4 || S == 6 || S == 8, "symbols of 1 (Short family only), 2, 3, 4, 6 or 8 bytes");
  // (8 bit symbols: rle8_multi_short / rle8_1symlut_short -- maximal runs of equal bytes, the same process_symbol, rleX_Xsl_short.h:380-667; the extreme 8 bit
  //  codecs have their own kernel, hsrle_encode8p.hip.h: their reference bodies have rules of their own)
  static_assert(S != 1 || ((FAM == SHORT0 || FAM == SHORT1) && AL == 0), "8 bit: Short without a list / with a one-symbol list");
  static_assert(FAM != LUT3 || S >= 3, "the LUT codecs of 1 and 2 byte symbols have runs whose storing depends on the list");
  using TR = Traits<FAM, S, AL>;
  constexpr bool SH = FAM == SHORT0 || FAM == SHORT1 || FAM == SHORT3; // Short family: one-byte packed header or the 3-byte form (rleX_Xsl_short.h:152-357)
  constexpr bool SH1 = FAM == SHORT1;                                  // ... with a one-symbol list: the state is (lastRLE, last stored symbol), as for Packed
  constexpr bool SH3 = FAM == SHORT3;                                  // ... with the three-symbol list: every run stored, the index as for LUT3
  constexpr bool PK = FAM == PACKED;
  constexpr bool LUT = FAM == LUT3;
  constexpr bool MTF3 = LUT || SH3;                                    // three-symbol move-to-front list, every run stored
  constexpr uint32_t SU = (uint32_t)S;
  constexpr uint32_t SHORT = TR::SHORT, MEDIUM = TR::MEDIUM, LONG = TR::LONG, MAXR = TR::MAXRANGE;
  constexpr bool R7 = TR::kRange7;
  // bytes of the literal terminator's fixed part (plain / Packed: of either terminator's).  Short: 3 header bytes, u16 0, u32 literals + 2 and, without a list,
  // a zero symbol; the end terminator: 3 header bytes, u16 0, u16 0 and, without a list, one zero byte (rleX_Xsl_short.h:976-1032)
  constexpr uint32_t TERM = SH ? ((SH1 || SH3) ? 9u : 9u + SU) : (LUT ? 8u : (PK ? 5u : SU + 5u) + (R7 ? 4u : 5u));
  constexpr uint32_t TERM_END = SH ? ((SH1 || SH3) ? 7u : 8u) : (LUT ? 6u : TERM);   // LUT: the end terminator is shorter (rleX_Xsl.h:319-338)
  constexpr uint32_t HDR = 8u;                                          // stream header: u32 uncompressed, u32 compressed
  const uint32_t lane = threadIdx.x;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint32_t base = lane * 64u;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const uint32_t *const myRecs = sc.recs + (uint64_t)b * sc.recStride;
  // (the stream's place: asked for at the start, needed at the very end -- not a dependent load in front of the copy-out)
  [[maybe_unused]] uint64_t myOffset = 0;
  if constexpr (MODE == 1) myOffset = offsets[b];
  // (and the block's second 64 records, where its record area holds that many: a block of more than 64 stored runs -- the rule on video-shaped data -- does not wait for them in its second round)
  [[maybe_unused]] uint32_t rec1 = 0;                                    // (the third and fourth 64 too: measured no better, four loads for every block)
  if constexpr (MODE == 1) { if (sc.recStride >= 128u) rec1 = myRecs[64u + threadIdx.x]; }

  uint32_t recN = kPpNoRecords;
  if constexpr (MODE == 1) recN = sc.recCount[b];
  const bool fromRecs = MODE == 1 && recN != kPpNoRecords;

  // the input image (both modes: symbols are compared and written from it), and the zeroed stream image (MODE 1)
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);
  if (lane < 2u) lds_st128(sh.inb + kPpInPad + kPpMaxBlock + 16u * lane, zero4);      // (symbol reads reach up to 11 bytes behind the last position)

  // ---- 1. 64 match bits per lane: m[j] = (d[j] == d[j + S]) ----
  uint32_t R = recN;
  uint64_t candLeft = 0;
  uint32_t candAt = 0;
  if (!fromRecs)
  {
    uint32_t w[18];
#pragma unroll
    for (int j = 0; j < 4; j++) { w[4 * j] = x[j].x; w[4 * j + 1] = x[j].y; w[4 * j + 2] = x[j].z; w[4 * j + 3] = x[j].w; }
    w[16] = wave_shl1(x[0].x, 0u);
    w[17] = wave_shl1(x[0].y, 0u);
    uint64_t m64 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      uint32_t t[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        const int i = 4 * j + k;
        uint32_t s;
        if constexpr (S == 1) s = alignbyte(w[i + 1], w[i], 1);
        else if constexpr (S == 2) s = alignbyte(w[i + 1], w[i], 2);
        else if constexpr (S == 3) s = alignbyte(w[i + 1], w[i], 3);
        else if constexpr (S == 4) s = w[i + 1];
        else if constexpr (S == 6) s = alignbyte(w[i + 2], w[i + 1], 2);
        else s = w[i + 2];
        t[k] = w[i] ^ s;
      }
      m64 |= (uint64_t)zero_mask16(t[0], t[1], t[2], t[3]) << (16 * j);
    }
    // position j matches only if j + S < n (bytes at or beyond n never match)
    const int32_t vb = (int32_t)n - (int32_t)SU - (int32_t)base;
    const uint32_t validBits = vb <= 0 ? 0u : (vb >= 64 ? 64u : (uint32_t)vb);
    m64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(m64 >> 63), 0u);
    const uint64_t prev = (m64 << 1) | carry;
    const uint64_t starts = m64 & ~prev;
    const uint64_t ends = ~m64 & prev;                                     // bit i: a stretch of set bits ends in front of position base + i
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
    // candidates: the stretches of at least S bits (data over a small alphabet is full of shorter ones: d[j] == d[j + S] by accident -- with every stretch
    // a candidate the video-shaped rle64 buffer took 8 rounds per block).  full bit i: the S bits up to and including position i are all set
    // (log-step ANDs of the bits shifted up, the bits that come in from the lane in front taken from its top dword)
    auto shl_in = [&](uint64_t v, uint32_t t) __attribute__((always_inline)) -> uint64_t {
      const uint32_t top = wave_shr1((uint32_t)(v >> 32), 0u);
      return (v << t) | (uint64_t)(top >> (32u - t));
    };
    uint64_t full;
    {
      const uint64_t c2 = m64 & shl_in(m64, 1u);
      if constexpr (S == 1) full = (TR::SMINS >= 3u) ? c2 : m64;            // (a run of two bytes can only be stored where the codec's shortest run is two)
      else if constexpr (S == 2) full = c2;
      else if constexpr (S == 3) full = c2 & shl_in(m64, 2u);
      else
      {
        const uint64_t c4 = c2 & shl_in(c2, 2u);
        if constexpr (S == 4) full = c4;
        else if constexpr (S == 6) full = c4 & shl_in(c2, 4u);
        else full = c4 & shl_in(c4, 4u);
      }
    }
    const uint64_t cands = ends & shl_in(full, 1u);                        // the stretch that ends in front of position i is full at i - 1
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    candLeft = cands;
    candAt = inclCnt - cnt;
  }
  wave_sync();

  [[maybe_unused]] auto put_chunks = [&](uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd) __attribute__((always_inline)) {
    const uint32_t de = ds + len, D0 = ds & ~15u;
    for (uint32_t t = t0; t < tEnd; t += tStep)
    {
      const uint32_t D = D0 + 16u * t;
      const uint32_t wa = kPpInPad + src + D - ds;
      const uint32_t *const wq = (const uint32_t *)(sh.inb + (wa & ~3u));
      const uint32_t q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3], q4 = wq[4], sb = wa & 3u;
      const u32x4 v = u32x4{ alignbyte(q1, q0, sb), alignbyte(q2, q1, sb), alignbyte(q3, q2, sb), alignbyte(q4, q3, sb) };
      const uint32_t lo = D < ds ? ds - D : 0u, hi = de - D < 16u ? de - D : 16u;      // chunk bytes [lo, hi)
      const u32x4 mh = lds_ld128(sh.mlut + (hi << 4)), ml = lds_ld128(sh.mlut + (lo << 4));
      unsigned long long *const ip = (unsigned long long *)(sh.img + D);
      const uint64_t w0 = (uint64_t)(v.x & mh.x & ~ml.x) | ((uint64_t)(v.y & mh.y & ~ml.y) << 32), w1 = (uint64_t)(v.z & mh.z & ~ml.z) | ((uint64_t)(v.w & mh.w & ~ml.w) << 32);
      atomicOr(ip, w0);
      atomicOr(ip + 1, w1);
    }
  };

  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t carL = 0;                                 // lastRLE in front of the round's first candidate
  uint64_t carY = 0;                                 // Packed: the last stored symbol (starts as zeros)
  uint32_t carE = 0;                                 // end of the last run found (stored or not): where the scan for the next one resumes
  // LUT: the move-to-front list [lA, lB, lY] (rleX_Xsl.h:279-287: 0x00, 0x7F, 0xFF in every symbol byte)
  constexpr uint64_t SMASK = (S >= 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);
  [[maybe_unused]] uint64_t lA = 0ull, lB = 0x7F7F7F7F7F7F7F7Full & SMASK, lY = 0xFFFFFFFFFFFFFFFFull & SMASK;
  uint32_t pos = HDR;
  uint32_t K = 0;
  bool ended = false;
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    const bool have = r0 + lane < R;
    const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
    uint32_t p = 0, e = 0, inL = 0, outL = 0;
    uint64_t sym = 0;
    bool same = false;
    int k = 0;
    [[maybe_unused]] uint32_t mtf = 0;                                   // LUT: the symbol's place in the list (3: not there)
    if (fromRecs)
    {
      const uint32_t rec = (r0 == 0u) ? rec0 : ((r0 == 64u && sc.recStride >= 128u) ? rec1 : (have ? myRecs[r0 + lane] : 0u));
      p = rec & 0xFFFu; e = ((rec >> 12) & 0xFFFu) + 1u;
      same = ((rec >> 24) & 1u) != 0u;
      k = have ? 1 + (int)((rec >> 25) & 1u) : 0;
      if constexpr (MTF3) { mtf = (rec >> 24) & 3u; k = have ? 1 : 0; }
      outL = e;
      inL = wave_shr1(outL, carL);
      if constexpr (SH3)
      {
        // (one-byte or 3-byte header: from the gap and the count, as decide() below)
        const uint32_t cnt = e - p, scu_ = AL ? cnt / SU - TR::SMINS / SU + 2u : cnt - TR::SMINS + 2u;
        if (have && !(p - inL <= TR::SMAXPR && scu_ - 2u <= TR::SMAXPC)) k = 2;
      }
      sym = pp_symbol<S>(sh.inb, kPpInPad + (have ? p : 0u));
    }
    else
    {
      while (candLeft != 0ull && candAt < r0 + 64u)
      {
        sh.lst[candAt - r0] = (uint16_t)(base + (uint32_t)__builtin_ctzll(candLeft));
        candAt++;
        candLeft &= candLeft - 1ull;
      }
      wave_sync();
      const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;               // the stretch's match bits end in front of position q
      wave_sync();
      // the stretch's first position: the last start bit below q (q itself is a clear bit)
      const uint32_t qm = have ? q - 1u : 0u, iq = qm >> 6, bit = qm & 63u;
      const uint64_t st = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      const uint32_t s0 = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;

      // -- where does the run start and end?  The scan resumes at the end of the run before: known at once unless that run's tail can reach in here
      const uint32_t qLeft = wave_shr1(have ? q : 0u, 0u);                  // (lane 0: decided by carE below)
      bool geoKnown = !have || (lane != 0u ? s0 >= qLeft + SU : s0 >= carE);
      uint32_t outE = 0;                                                   // end of the last run up to and including this candidate
      bool outEKnown = false;
      auto run_from = [&](uint32_t resume) __attribute__((always_inline)) {
        // (p, e) of this stretch's run when the scan resumes at `resume`; e == 0: no run
        const uint32_t ps = resume > s0 ? resume : s0;
        p = ps; e = 0u;
        if (have && q >= ps + SU)
        {
          const uint32_t Leff = q - ps;
          const uint32_t whole = ((Leff + SU) / SU) * SU;
          const uint32_t eW = ps + whole;
          e = (!AL && eW + SU <= n) ? q + SU : eW;
        }
      };
      if (geoKnown) { run_from(0u); outE = e; outEKnown = have && e != 0u; }   // (a candidate without a run hands its left neighbour's end through: not known yet)
      if (!have) { outEKnown = true; outE = 0u; }
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outEKnown ? 1u : 0u, 1u), le = wave_shr1(outE, carE);
        if (have && !outEKnown && lk != 0u)
        {
          if (!geoKnown) { run_from(le); geoKnown = true; }
          outE = (e != 0u) ? e : le;
          outEKnown = true;
        }
        if (__ballot(!outEKnown) == 0ull) break;
      }
      carE = wave_lane(outE, lastLane);
      const bool isRun = have && e != 0u;
      const uint32_t count = e - p;
      sym = pp_symbol<S>(sh.inb, kPpInPad + (isRun ? p : 0u));

      // -- emit decisions (rleX_extreme_cpu_encode.h:174-311): a chain through (lastRLE, last symbol)
      // (LUT, symbols of 3 bytes and more: EVERY run is stored -- count >= 2 S >= 6 >= 3 + the largest penalty a block can produce, rleX_Xsl.h:116-132 --
      //  so the chain below only hands lastRLE through the candidates without a run; what the list decides is the symbol's index, further down)
      const bool sure = isRun && (MTF3 || count >= (SH ? TR::SMINL : LONG));
      auto decide = [&](uint32_t iL, uint64_t iY, bool &sm) __attribute__((always_inline)) -> int {
        if constexpr (LUT) { sm = false; return 1; }
        if constexpr (SH)
        {
          // process_symbol of the Short family (rleX_Xsl_short.h:152-215): the packet's size over the one-byte form is a penalty on the shortest run stored;
          // 1: one-byte header, 2: the 3-byte form (with 16 bit fields behind it where count or range need them)
          const uint32_t gp = p - iL, range = gp + 2u;
          sm = SH1 && sym == iY;
          const uint32_t scu = AL ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u;
          const bool pack1 = gp <= TR::SMAXPR && scu - 2u <= TR::SMAXPC;
          uint32_t pen = (SH1 && !sm) ? SU : 0u;
          if (!pack1) pen += 2u + (range <= TR::SMAXTR ? 0u : 2u) + (scu <= TR::SMAXTC ? 0u : 2u);      // (nothing in a block needs a 32 bit field)
          if (!SH3 && !(count >= TR::SMINL || count >= TR::SMINS + pen)) return 0;
          return pack1 ? 1 : 2;
        }
        const uint32_t rng = p - iL + 1u;
        sm = PK && sym == iY;
        const bool shortOk = rng <= MAXR && (PK ? (sm || count >= MEDIUM) : count >= SHORT);   // (count >= 2 S >= the Packed SHORT of 3)
        return shortOk ? 1 : (count >= LONG ? 2 : 0);
      };
      uint64_t outY = sym, inY = 0;
      outL = e;
      bool outKnown = sure || !have, inKnown = !have;
      // (a candidate without a run changes nothing: it hands the state through like one whose run is not stored)
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u);
        const uint32_t lr = wave_shr1(outL, carL), lylo = wave_shr1((uint32_t)outY, (uint32_t)carY), lyhi = wave_shr1((uint32_t)(outY >> 32), (uint32_t)(carY >> 32));
        if (!inKnown && lk != 0u) { inKnown = true; inL = lr; inY = (uint64_t)lylo | ((uint64_t)lyhi << 32); }
        if (inKnown && !outKnown)
        {
          bool sm;
          if (!isRun || decide(inL, inY, sm) == 0) { outL = inL; outY = inY; }
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      k = isRun ? decide(inL, inY, same) : 0;
      carY = (uint64_t)wave_lane((uint32_t)outY, lastLane) | ((uint64_t)wave_lane((uint32_t)(outY >> 32), lastLane) << 32);
      if constexpr (MTF3)
      {
        // The list holds the three most recent DISTINCT symbols.  A run whose symbol is the one before it has index 0 and changes nothing; the others
        // ("heads" u_0, u_1, ...: consecutive heads differ) see the list [u_t-1, u_t-2, y_t] with y_t = (u_t-1 == u_t-3) ? y_t-1 : u_t-3 -- "copy from the
        // left unless reset": a prefix maximum over the reset positions, no chain.  The list in front of the round is three heads in front of head 0.
        uint64_t *const symList = sh.mtfScratch, *const headList = symList + 72;
        const uint32_t runIncl = wave_scan_add(isRun ? 1u : 0u), rho = runIncl - (isRun ? 1u : 0u);
        if (lane == 0u) { symList[0] = lA; headList[0] = lY; headList[1] = lB; headList[2] = lA; }
        if (isRun) symList[1u + rho] = sym;
        wave_sync();
        const uint64_t prevSym = symList[isRun ? rho : 0u];
        const bool head = isRun && sym != prevSym;
        const uint32_t headIncl = wave_scan_add(head ? 1u : 0u), t = headIncl - (head ? 1u : 0u);
        if (head) headList[3u + t] = sym;
        wave_sync();
        const uint64_t u1 = headList[2u + (head ? t : 0u)], u2 = headList[1u + (head ? t : 0u)], u3 = headList[head ? t : 0u];
        const int32_t key = (head && u1 != u3) ? (int32_t)t : -1;
        const int32_t tr = wave_scan_max(key);                           // (>= 0 for every head: head 0 resets -- the list's three entries differ)
        const uint64_t y = headList[head ? (uint32_t)tr : 0u];
        mtf = !head ? 0u : (sym == u2 ? 1u : (sym == y ? 2u : 3u));
        // the list behind the round: behind its last head
        const unsigned long long heads = __ballot(head);
        if (heads != 0ull)
        {
          const int hl_ = 63 - __builtin_clzll(heads);
          const uint64_t yNext = (sym == u2) ? y : u2;
          lA = (uint64_t)wave_lane((uint32_t)sym, hl_) | ((uint64_t)wave_lane((uint32_t)(sym >> 32), hl_) << 32);
          lB = (uint64_t)wave_lane((uint32_t)u1, hl_) | ((uint64_t)wave_lane((uint32_t)(u1 >> 32), hl_) << 32);
          lY = (uint64_t)wave_lane((uint32_t)yNext, hl_) | ((uint64_t)wave_lane((uint32_t)(yNext >> 32), hl_) << 32);
        }
        wave_sync();
      }
    }
    const uint32_t count = e - p, gap = p - inL, rng = gap + (LUT ? 2u : 1u);

    // ---- packet header (rleX_extreme_cpu_encode.h:174-311): [count | same] [symbol] [range] (Packed) / [symbol] [count] [range] (plain);
    //      LUT (rleX_Xsl.h:190-250): u16 {index, count, range}, [symbol if new], [u16 count], [u16 range] ----
    const uint32_t cfield = LUT ? (AL ? count / SU - 3u / SU + 2u : count - 1u) : (AL ? count / SU - SHORT / SU + 1u : count - SHORT + 1u);
    const uint32_t cMax = (PK || LUT) ? 127u : 255u;
    const uint32_t cBytes = LUT ? (cfield <= 127u ? 0u : 2u) : (cfield <= cMax ? 1u : 5u);
    const uint32_t sBytes = MTF3 ? (mtf == 3u ? SU : 0u) : (((PK || SH1) && same) ? 0u : SU);
    const uint32_t rBytes = LUT ? (rng <= 127u ? 0u : 2u) : ((k == 1) ? 1u : (R7 ? 4u : 5u));
    // Short: count field value (+ 2), range = gap + 2; the 3-byte form carries 9 bits of count and SRB of range, 16 bit fields follow where that is not enough
    [[maybe_unused]] const uint32_t scu = SH ? (AL ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u) : 0u;
    [[maybe_unused]] const uint32_t srange = gap + 2u;
    const uint32_t hl = !k ? 0u : (SH ? ((k == 1) ? 1u : 3u + (scu > TR::SMAXTC ? 2u : 0u) + (srange > TR::SMAXTR ? 2u : 0u)) + sBytes
                                      : (LUT ? 2u : 0u) + cBytes + sBytes + rBytes);
    const uint32_t myBytes = k ? hl + gap : 0u;
    const uint32_t incl = wave_scan_add(myBytes | (k ? 0x10000u : 0u));
    const uint32_t tot = wave_lane(incl, 63);
    if constexpr (MODE == 0)
    {
      const uint32_t idx = K + (incl >> 16) - 1u;
      if (k && idx < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + idx] = p | ((e - 1u) << 12) | (MTF3 ? mtf << 24 : ((same ? 1u << 24 : 0u) | (k == 2 ? 1u << 25 : 0u)));
    }
    else
    {
      const uint32_t at0 = pos + (incl & 0xFFFFu) - myBytes;
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        uint32_t a = at0;
        if constexpr (LUT)
        {
          const uint32_t c7 = cfield <= 127u ? cfield : 1u, r7 = rng <= 127u ? rng : 1u;   // (1: a 16 bit field follows; nothing in a block needs 32)
          pp_or_bytes(sh.img, a, (uint64_t)((mtf << 14) | (c7 << 7) | r7), 2u); a += 2u;
          if (mtf == 3u) { pp_or_bytes(sh.img, a, sym, SU); a += SU; }
          if (cBytes) { pp_or_bytes(sh.img, a, (uint64_t)cfield, 2u); a += 2u; }
          if (rBytes) pp_or_bytes(sh.img, a, (uint64_t)rng, 2u);
        }
        else if constexpr (SH)
        {
          // [list index | count | range] in one byte, or: [index | all-ones count | 9 bit count | SRB bit range] in three (rleX_Xsl_short.h:216-357)
          const uint32_t mi = (SH3 ? mtf : ((SH1 && !same) ? 1u : 0u)) << (TR::SCB + TR::SRBP);
          if (k == 1) { pp_or_bytes(sh.img, a, (uint64_t)(mi | ((scu - 2u) << TR::SRBP) | gap), 1u); a += 1u; }
          else
          {
            const uint32_t scx = scu <= TR::SMAXTC ? scu : 1u, rx = srange <= TR::SMAXTR ? srange : 1u;      // (1: a 16 bit field follows)
            const uint32_t f = scx << (TR::SRB - 8u);
            const uint32_t b0 = (mi | (TR::SCINV << TR::SRBP) | (f >> 8)) & 0xFFu, b1 = (f | (rx >> 8)) & 0xFFu, b2 = rx & 0xFFu;
            pp_or_bytes(sh.img, a, (uint64_t)(b0 | (b1 << 8) | (b2 << 16)), 3u); a += 3u;
            if (scx != scu) { pp_or_bytes(sh.img, a, (uint64_t)scu, 2u); a += 2u; }
            if (rx != srange) { pp_or_bytes(sh.img, a, (uint64_t)srange, 2u); a += 2u; }
          }
          if (sBytes) pp_or_bytes(sh.img, a, sym, SU);
        }
        else
        {
          const uint64_t cval = cfield <= cMax ? (uint64_t)(cfield | ((PK && same) ? 0x80u : 0u)) : (((uint64_t)cfield << 8) | ((PK && same) ? 0x80u : 0u));
          const uint64_t rval = (k == 1) ? (uint64_t)(R7 ? (rng << 1) & 0xFFu : rng) : (R7 ? (uint64_t)((rng << 1) | 1u) : ((uint64_t)rng << 8));
          if constexpr (PK)
          {
            pp_or_bytes(sh.img, a, cval, cBytes); a += cBytes;
            if (!same) { pp_or_bytes(sh.img, a, sym, SU); a += SU; }
          }
          else
          {
            pp_or_bytes(sh.img, a, sym, SU); a += SU;
            pp_or_bytes(sh.img, a, cval, cBytes); a += cBytes;
          }
          pp_or_bytes(sh.img, a, rval, rBytes);
        }
        ds = at0 + hl;
        if (gap > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)inL | ((uint64_t)ds << 13) | ((uint64_t)gap << 26); }
        else if (gap != 0u) nch = ((ds + gap - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)
      {
        if (t < nch) put_chunks(inL, ds, gap, t, 1u, t + 1u);
        if (t + 1u < nch) put_chunks(inL, ds, gap, t + 1u, 1u, t + 2u);
      }
    }
    carL = wave_lane(outL, lastLane);
    pos += tot & 0xFFFFu;
    K += tot >> 16;
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  // ---- 3. terminator, stream size (rleX_extreme_cpu_encode.h:384-603) ----
  const uint32_t kLit = ended ? 0u : n - carL;
  const uint32_t streamSize = pos + (ended ? TERM_END : TERM) + kLit;
  if constexpr (MODE == 0)
  {
    if (lane == 0u) { sizes[b] = streamSize; sc.recCount[b] = (K <= sc.recStride) ? K : kPpNoRecords; }
    return;
  }
  else
  {
    if (lane < 8u)
    {
      const uint64_t h = (uint64_t)n | ((uint64_t)streamSize << 32);
      sh.img[lane] = (uint8_t)(h >> (8u * lane));
    }
    if (lane == 16u)
    {
      if constexpr (SH)
      {
        // end: [all-ones count] [STB] 01, u16 0, u16 0 (, 00);  literals: [all-ones count] [STB] 00, u16 0, u32 literals + 2 (, zero symbol)
        pp_or_bytes(sh.img, pos, (uint64_t)((TR::SCINV << TR::SRBP) | (TR::STB << 8) | (ended ? 1u << 16 : 0u)), 3u);
        if (!ended) pp_or_bytes(sh.img, pos + 5u, (uint64_t)(kLit + 2u), 4u);
      }
      else if constexpr (LUT)
      {
        // end: u16 (1 << 7) | 1, u16 0, u16 0;  literals: u16 1 << 7, u16 0, u32 literals + 2  (rleX_Xsl.h:319-338)
        sh.img[pos] = ended ? 0x81 : 0x80;
        if (!ended) pp_or_bytes(sh.img, pos + 4u, (uint64_t)(kLit + 2u), 4u);
      }
      else
      {
        // plain: S zero bytes, 00, u32 0 | Packed: 80, u32 0;  then the range field: (u32 value << 1 | 1) where the codec has 7-bit ranges, else 00, u32 value;
        // value = 0 (end) or literals + 1
        const uint32_t a0 = pos + (PK ? 0u : SU);
        if constexpr (PK) sh.img[a0] = 0x80;
        const uint32_t val = ended ? 0u : kLit + 1u;
        if constexpr (R7) pp_or_bytes(sh.img, a0 + 5u, (uint64_t)((val << 1) | 1u), 4u);
        else pp_or_bytes(sh.img, a0 + 5u, (uint64_t)val << 8, 5u);
      }
    }
    wave_sync();
    {
      const uint32_t nj = sh.jobCount;
      for (uint32_t j = 0; j <= nj; j++)
      {
        uint32_t src, ds, len;
        if (j < nj) { const uint64_t jb = sh.jobs[j]; src = (uint32_t)jb & 0x1FFFu; ds = (uint32_t)(jb >> 13) & 0x1FFFu; len = (uint32_t)(jb >> 26); }
        else { src = carL; ds = pos + TERM; len = kLit; }
        if (len != 0u) put_chunks(src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
      }
    }
    wave_sync();
    {
      uint8_t *const dst = payload + myOffset;
      const uint32_t nFull = streamSize >> 4, tail = streamSize & 15u;
      for (uint32_t c = lane; c < nFull; c += 64u)
        st128(dst + 16u * c, lds_ld128(sh.img + 16u * c));
      if (lane < tail) dst[16u * nFull + lane] = sh.img[16u * nFull + lane];
    }
  }
}

template <int FAM, int S, int AL, int MODE>
__global__ __launch_bounds__(64) void k_encodeS_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                   const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpScratch sc)
{
  __shared__ PpShared<MODE != 0, true, FAM == LUT3 || FAM == SHORT3> sh;
  if (MODE != 0 && threadIdx.x < 17u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (c == 16u) ? ~0u : (uint32_t)part, p1 = (c == 16u) ? ~0u : (uint32_t)(part >> 32);
    lds_st128(sh.mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }
  wave_sync();
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
  if (b < nBlocks)
  {
    u32x4 x[4];
    pp_load(in, U, B, b, x);
    uint32_t rec0 = 0;
    if constexpr (MODE == 1) rec0 = sc.recs[(uint64_t)b * sc.recStride + threadIdx.x];
    ppS_block<FAM, S, AL, MODE>(in, U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
  }
}

} // namespace hsrle
