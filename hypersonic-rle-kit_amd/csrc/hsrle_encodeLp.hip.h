// hsrle_encodeLp.hip.h -- POSITION-PARALLEL encoder for the LUT codecs whose packets depend on the move-to-front list beyond a closed form: every 7 symbol
// LUT codec (rle8_7symlut, rle16 / 24 / 32 / 48 / 64 _7symlut_sym / _7symlut_byte: 11 codecs) and the 3 symbol LUT codecs of 1 and 2 byte symbols
// (rle8_3symlut, rle16_3symlut_sym, rle16_3symlut_byte), blocks of at most 4 KiB (round 6).  hsrle_encodeSp.hip.h (one wave per block, records, the payload
// written once) with the list handled in parallel.
//
// Replaces: src/rleX_Xsl.h:93-264 (state, process_symbol: penalty :116-132, move to front :134-188, header :190-250), :269-346 (8 bit wrapper),
//           src/rleX_Xsl_multibyte_encoder.h:18-370 -- and, in this library, the ring / run list encoders + staging slots + k_compact for containers of these codecs.
//
// What depends on the list:
//   * the symbol's INDEX in the packet header.  A run's index is the number of distinct symbols stored since its own symbol was stored last (K: not among the
//     last K).  Per round of 64 candidates: the stored symbols go to an LDS list behind the K symbols of the list in front of the round; a candidate compares
//     its symbol with its K predecessors (distance to the nearest equal one: `dist`), and the distinct symbols in between are the predecessors whose own
//     nearest equal lies outside that window -- two passes of K reads, no chain.  Only where the symbol is not among the K predecessors AND those hold a
//     duplicate (fewer than K distinct) a lane walks further back, keeping the distinct symbols it has seen;
//   * 1 and 2 byte symbols: WHETHER a run is stored (count >= 3 + penalty, + 1 for a symbol that is not in the list, rleX_Xsl.h:116-132).  Only runs of exactly
//     3 + (2 if the range needs a 16 bit field) bytes are affected (8 bit: 3 or 5 bytes; 16 bit: 5 bytes); runs of >= 6 bytes are stored whatever the state.
//     The decisions are the fixed point of: guess the affected candidates' "not in the list" flags -> chain through lastRLE (DPP passes, as for plain) ->
//     indices -> flags again.  A candidate's flag is right once everything in front of it is, so every iteration settles at least one more of them; blocks
//     without such candidates (nearly all of the synthetic data) take one iteration.  Symbols of >= 3 bytes: every run is stored (count >= 2 S >= 6).
#pragma once

#include "hsrle_encode8sp.hip.h"   // pp_put_chunks; hsrle_encodeSp.hip.h: pp_symbol, pp_or_bytes

namespace hsrle {

template <bool EMIT, int K>
struct PpLutShared
{
  uint8_t img[EMIT ? (kPpMaxBlock + 193u + 15u + 16u + 15u) / 16u * 16u : 16u] __attribute__((aligned(16)));
  uint8_t inb[kPpInPad + kPpMaxBlock + 32u] __attribute__((aligned(16)));
  uint8_t mlut[EMIT ? 17u * 16u : 16u] __attribute__((aligned(16)));
  uint64_t starts[64];
  uint64_t symList[K + 64 + 1];                                         // [0, K): the list in front of the round, least recent first; then the round's stored symbols
  uint64_t carList[K];                                                  // the list behind the round (same order)
  uint64_t jobs[EMIT ? kPpJobs : 1u];
  uint16_t lst[64];
  uint16_t carryStart[64];
  uint8_t distList[K + 64 + 8];                                         // per list entry: distance to the nearest equal entry in front of it (K + 1: none within K)
  uint64_t symAll[64];                                                  // the round's symbols by lane (for the list behind the round)
  uint32_t jobCount;
};

template <int FAM, int S, int AL, int MODE>
__device__ __forceinline__ void ppL_block(uint64_t U, uint32_t B, uint32_t b, uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload,
                                          const PpScratch &sc, PpLutShared<MODE != 0, ((FAM == LUT3 || FAM == SHORT3) ? 3 : 7)> &sh, const u32x4 (&x)[4], uint32_t rec0)
{
  static_assert(FAM == LUT3 || FAM == LUT7 || FAM == SHORT3 || FAM == SHORT7, "3 / 7 symbol LUT, or the Short family with a 3 / 7 symbol list");
  static_assert(S == 1 || S == 2 || S == 3 || S == 4 || S == 6 || S == 8, "symbols of 1, 2, 3, 4, 6 or 8 bytes");
  static_assert(S != 1 || AL == 0, "8 bit: byte-aligned by nature");
  // Short family (src/rleX_Xsl_short.h:152-372): the same list, one-byte or three-byte headers, and process_symbol's rule -- count >= S + 11, or count >= 2 + the
  // bytes the packet needs beyond the one-byte form, S of them for a symbol that is not in the list: with symbols of up to 4 bytes WHETHER a run is stored depends on
  // the list for every count below S + 11 (from 6 bytes on every run is stored: 2 S >= 2 + S + 4)
  constexpr bool SH = FAM == SHORT3 || FAM == SHORT7;
  using TR = Traits<FAM, S, AL>;
  constexpr int K = (FAM == LUT3 || FAM == SHORT3) ? 3 : 7;
  constexpr uint32_t KU = (uint32_t)K;
  constexpr uint32_t SU = (uint32_t)S;
  constexpr uint32_t RB = K == 3 ? 7u : 6u, MAXR = (1u << RB) - 1u, MAXC = 127u, MSH = K == 3 ? 14u : 13u;
  constexpr bool NARROW = SH ? S <= 4 : S <= 2;                         // storing a run can depend on the list
  constexpr uint32_t TERM = SH ? 9u : 8u, TERM_END = SH ? 7u : 6u, HDR = 8u;
  // a run of this many bytes is stored whatever the state: LUT 3 + (2 + 1); Short 2 + the largest penalty a short count can meet (S for a new symbol, 2 for the
  // three-byte form, 2 for a 16 bit range; a 16 bit count means a run beyond S + 11 anyway)
  constexpr uint32_t SURE = SH ? (TR::SMINS + SU + 4u < TR::SMINL ? TR::SMINS + SU + 4u : TR::SMINL) : 6u;
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

  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);
  if (lane < 2u) lds_st128(sh.inb + kPpInPad + kPpMaxBlock + 16u * lane, zero4);
  if (lane < KU)
  {
    // the list every stream starts with (rleX_Xsl.h:279-287): 00, 7F, FF, 01, 7E, 80, FE in every symbol byte; carList holds it least recent first
    constexpr uint64_t SMASK = (S >= 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);
    const uint32_t i = KU - 1u - lane;
    const uint64_t v = i == 0u ? 0x00ull : (i == 1u ? 0x7Full : (i == 2u ? 0xFFull : (i == 3u ? 0x01ull : (i == 4u ? 0x7Eull : (i == 5u ? 0x80ull : 0xFEull)))));
    sh.carList[lane] = (v * 0x0101010101010101ull) & SMASK;
  }

  // ---- 1. match bits m[j] = (d[j] == d[j + S]), stretches, candidates (hsrle_encodeSp.hip.h) ----
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
    const int32_t vb = (int32_t)n - (int32_t)SU - (int32_t)base;
    const uint32_t validBits = vb <= 0 ? 0u : (vb >= 64 ? 64u : (uint32_t)vb);
    m64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(m64 >> 63), 0u);
    const uint64_t prev = (m64 << 1) | carry;
    const uint64_t starts = m64 & ~prev;
    const uint64_t ends = ~m64 & prev;
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
    auto shl_in = [&](uint64_t v, uint32_t t) __attribute__((always_inline)) -> uint64_t {
      const uint32_t top = wave_shr1((uint32_t)(v >> 32), 0u);
      return (v << t) | (uint64_t)(top >> (32u - t));
    };
    // candidates: the stretches of at least S bits; 8 bit: of at least 2 (a run of two bytes is never stored: count >= 3 + penalty)
    uint64_t full;
    {
      const uint64_t c2 = m64 & shl_in(m64, 1u);
      if constexpr (S == 1 && SH) full = m64;                            // (Short: a run of two bytes of a listed symbol right behind the run before is stored)
      else if constexpr (S <= 2) full = c2;
      else if constexpr (S == 3) full = c2 & shl_in(m64, 2u);
      else
      {
        const uint64_t c4 = c2 & shl_in(c2, 2u);
        if constexpr (S == 4) full = c4;
        else if constexpr (S == 6) full = c4 & shl_in(c2, 4u);
        else full = c4 & shl_in(c4, 4u);
      }
    }
    const uint64_t cands = ends & shl_in(full, 1u);
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    candLeft = cands;
    candAt = inclCnt - cnt;
  }
  wave_sync();

  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t carL = 0, carE = 0;
  bool exactFirst = false;                           // the list indices of the next round come from the exact mode straight away (hsrle_encodeLp.hip.h: see there)
  uint32_t pos = HDR;
  uint32_t K_ = 0;                                   // stored runs so far
  bool ended = false;
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    const bool have = r0 + lane < R;
    const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
    uint32_t p = 0, e = 0, inL = 0, outL = 0;
    uint64_t sym = 0;
    int k = 0;
    uint32_t mtf = 0;
    if (fromRecs)
    {
      const uint32_t rec = (r0 == 0u) ? rec0 : ((r0 == 64u && sc.recStride >= 128u) ? rec1 : (have ? myRecs[r0 + lane] : 0u));
      p = rec & 0xFFFu; e = ((rec >> 12) & 0xFFFu) + 1u;
      mtf = (rec >> 24) & 7u;
      k = have ? 1 : 0;
      outL = e;
      inL = wave_shr1(outL, carL);
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
      if (lane < KU) { sh.symList[lane] = sh.carList[lane]; sh.distList[lane] = (uint8_t)(KU + 1u); }
      wave_sync();
      const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;
      wave_sync();
      const uint32_t qm = have ? q - 1u : 0u, iq = qm >> 6, bit = qm & 63u;
      const uint64_t st = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      const uint32_t s0 = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;

      // -- where does the run start and end?  (hsrle_encodeSp.hip.h)
      const uint32_t qLeft = wave_shr1(have ? q : 0u, 0u);
      bool geoKnown = !have || (lane != 0u ? s0 >= qLeft + SU : s0 >= carE);
      uint32_t outE = 0;
      bool outEKnown = false;
      auto run_from = [&](uint32_t resume) __attribute__((always_inline)) {
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
      if (geoKnown) { run_from(0u); outE = e; outEKnown = have && e != 0u; }
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

      // EXACT MODE (data over a small alphabet: long stretches of the list's own symbols; entered when the K predecessors do not settle a lane, and for the
      // list behind the round).  One trip per DISTINCT stored symbol of the round (few, where this is needed): the lanes that store it as a ballot -> every
      // lane learns that symbol's last occurrence in front of itself.  F = those last occurrences as a bit set = the list's entries that the round has
      // renewed in front of this lane, most recent = highest; prevO = the last occurrence of the lane's own symbol; lastAll = F behind the round.
      uint64_t F = 0, lastAll = 0;
      int prevO = -1;
      bool manySymbols = true, exactDone = false;
      uint32_t dTrips = 0;                               // distinct stored symbols the last call found
      auto distinct_symbols = [&](uint32_t maxTrips = 64u) __attribute__((always_inline)) -> bool {
        F = 0ull; lastAll = 0ull; prevO = -1; dTrips = 0u;
        const uint64_t belowT = (1ull << lane) - 1ull;
        uint64_t rem = __ballot(k != 0);
        while (rem != 0ull)
        {
          if (dTrips++ == maxTrips) return false;
          const int leader = (int)__builtin_ctzll(rem);
          const uint64_t sg = (uint64_t)wave_lane((uint32_t)sym, leader) | ((uint64_t)wave_lane((uint32_t)(sym >> 32), leader) << 32);
          const uint64_t ms = __ballot(k != 0 && sym == sg);
          rem &= ~ms;
          lastAll |= 1ull << (63u - (uint32_t)__builtin_clzll(ms));
          const uint64_t lb = ms & belowT;
          if (lb != 0ull)
          {
            const uint32_t j = 63u - (uint32_t)__builtin_clzll(lb);
            F |= 1ull << j;
            if (sym == sg) prevO = (int)j;
          }
        }
        return true;
      };
      // my symbol's place in the list in front of the round (K: not there), and the places that lanes in front of me store again (a prefix OR over the lanes)
      auto list_place = [&](uint32_t &place, uint32_t &renewedBefore, uint32_t &renewedAll) __attribute__((always_inline)) {
        place = KU;
#pragma unroll
        for (uint32_t i = 0; i < KU; i++) if (sh.symList[KU - 1u - i] == sym) place = i;
        uint32_t v = (k && place < KU) ? 1u << place : 0u;
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
        renewedBefore = wave_shr1(v, 0u);
        renewedAll = wave_lane(v, 63);
      };

      // -- which runs are stored, and the symbols' list indices: a fixed point (see the header); one iteration where no run is short enough to depend on the list
      const bool sure = isRun && (!NARROW || count >= SURE);
      const bool anyOpen = NARROW && __ballot(isRun && !sure) != 0ull;
      bool notIn = true;                                                   // the guess: the symbol is not in the list ...
      if (anyOpen)
      {
        // ... unless the list in front of the round holds it (data over a small alphabet: nearly always the final answer)
#pragma unroll
        for (uint32_t i = 0; i < KU; i++) if (sh.symList[i] == sym) notIn = false;
      }
      for (uint32_t iter = 0; iter < 70u; iter++)
      {
        auto stored_with = [&](uint32_t iL, bool flagNotIn) __attribute__((always_inline)) -> bool {
          if constexpr (SH)
          {
            const uint32_t gp = p - iL;
            const uint32_t scu = (AL && S != 1) ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u;
            const bool pack1 = gp <= TR::SMAXPR && scu - 2u <= TR::SMAXPC;
            const uint32_t pen = (flagNotIn ? SU : 0u) + (pack1 ? 0u : 2u + (gp + 2u <= TR::SMAXTR ? 0u : 2u) + (scu <= TR::SMAXTC ? 0u : 2u));
            return count >= TR::SMINL || count >= TR::SMINS + pen;
          }
          else
          {
            const uint32_t rng = p - iL + 2u;
            const uint32_t pen = (rng <= MAXR ? 0u : 2u) + (flagNotIn ? 1u : 0u);   // (a count field beyond 127 means a run of >= S + 10 bytes)
            return count >= SU + 10u || count >= 3u + pen;
          }
        };
        auto stored_if = [&](uint32_t iL) __attribute__((always_inline)) -> bool { return stored_with(iL, notIn); };
        outL = e;
        bool outKnown = sure || !have, inKnown = !have;
        for (uint32_t pass = 0; pass < 66u; pass++)
        {
          const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), lr = wave_shr1(outL, carL);
          if (!inKnown && lk != 0u) { inKnown = true; inL = lr; }
          if (inKnown && !outKnown)
          {
            if (!isRun || !stored_if(inL)) outL = inL;
            outKnown = true;
          }
          if (__ballot(!inKnown) == 0ull) break;
        }
        k = (isRun && stored_if(inL)) ? 1 : 0;

        // the stored symbols behind the list in front of the round; every candidate looks at its K predecessors
        const uint32_t stIncl = wave_scan_add(k ? 1u : 0u), rho = stIncl - (k ? 1u : 0u);
        if (k) sh.symList[KU + rho] = sym;
        wave_sync();
        const uint32_t t0 = KU + rho;                                      // my predecessors: entries t0 - 1, t0 - 2, ...
        // the index from the exact mode's sets (distinct_symbols() has run): the distinct symbols stored behind my symbol's last occurrence, or -- not stored in
        // this round yet -- the round's distinct symbols in front of me + the listed symbols in front of mine that have not been renewed
        auto exact_index = [&](uint32_t mIn) __attribute__((always_inline)) -> uint32_t {
          uint32_t place, renewedBefore, renewedAll;
          list_place(place, renewedBefore, renewedAll);
          uint32_t mm = mIn;
          if (isRun)
          {
            if (prevO >= 0) mm = (uint32_t)__builtin_popcountll(F & ~((2ull << prevO) - 1ull));
            else
            {
              const uint32_t dR = (uint32_t)__builtin_popcountll(F);
              mm = (dR >= KU || place >= KU) ? KU : dR + (uint32_t)__builtin_popcount(~renewedBefore & ((1u << place) - 1u));
            }
            if (mm > KU) mm = KU;
          }
          return mm;
        };
        uint32_t m = KU;
        exactDone = false;
        // (a block whose last round ran in the exact mode with few distinct symbols starts there: data over a small alphabet stays that way)
        bool viaPredecessors = true;
        if (exactFirst)
        {
          if (distinct_symbols(24u)) { exactDone = true; manySymbols = false; m = exact_index(KU); viaPredecessors = false; }
          else exactFirst = false;
        }
        if (viaPredecessors)
        {
        uint32_t dist = KU + 1u;
#pragma unroll
        for (int ii = K; ii >= 1; ii--)
          if (sh.symList[t0 - (uint32_t)ii] == sym) dist = (uint32_t)ii;
        if (k) sh.distList[t0] = (uint8_t)dist;
        wave_sync();
        // distinct symbols among the predecessors in front of the match (all K of them where there is none): those whose own nearest equal lies outside
        uint32_t distinct = 0;
#pragma unroll
        for (uint32_t i = 1u; i <= KU; i++)
        {
          const uint32_t dw = (uint32_t)sh.distList[t0 - i];               // predecessor i; the window reaches back to predecessor (dist - 1), or to K
          const uint32_t reach = (dist <= KU ? dist - 1u : KU);
          if (i <= reach && dw > reach - i) distinct++;
        }
        bool slow = false;
        if (isRun)
        {
          if (dist <= KU) m = distinct;                                    // found: the index is the number of distinct symbols in front of it
          else if (distinct < KU) slow = true;                             // not among the K predecessors, and those hold a duplicate: look further back
        }
        manySymbols = (uint32_t)__builtin_popcountll(__ballot(k != 0 && dist > KU)) > 16u;
        if (__ballot(slow) != 0ull && manySymbols)
        {
          // many different symbols in the round (their first occurrences have no equal among their K predecessors): the few unsettled lanes walk further back,
          // keeping the distinct symbols they have seen -- a handful of steps on such data; whoever is not done after 3 K steps takes the exact mode below
          if (slow)
          {
            int32_t t = (int32_t)t0 - 1;
            if constexpr (S == 1)
            {
              // (8 bit symbols: the seen symbols as the bytes of one word, unused places hold my own symbol -- which no visited entry equals)
              uint64_t seenB = (sym & 0xFFull) * 0x0101010101010101ull;
              uint32_t cntSeen = 0;
              for (int steps = 0; t >= 0 && steps < 3 * K; t--, steps++)
              {
                const uint64_t h = sh.symList[t];
                if (h == sym) { m = cntSeen; slow = false; break; }
                const uint64_t xr = seenB ^ ((h & 0xFFull) * 0x0101010101010101ull);
                if ((zero_bytes((uint32_t)xr) | zero_bytes((uint32_t)(xr >> 32))) == 0u)
                {
                  seenB = (seenB << 8) | (h & 0xFFull);
                  cntSeen++;
                  if (cntSeen == KU) { slow = false; break; }
                }
              }
            }
            else
            {
              uint64_t seen[K];
#pragma unroll
              for (int i = 0; i < K; i++) seen[i] = sym;
              uint32_t cntSeen = 0;
              for (int steps = 0; t >= 0 && steps < 3 * K; t--, steps++)
              {
                const uint64_t h = sh.symList[t];
                if (h == sym) { m = cntSeen; slow = false; break; }
                bool isNew = true;
#pragma unroll
                for (int i = 0; i < K; i++) isNew = isNew && seen[i] != h;
                if (isNew)
                {
#pragma unroll
                  for (int i = K - 1; i > 0; i--) seen[i] = seen[i - 1];
                  seen[0] = h;
                  cntSeen++;
                  if (cntSeen == KU) { slow = false; break; }
                }
              }
            }
            if (t < 0) slow = false;                                         // (walked through the list in front of the round: not there)
          }
        }
        if (__ballot(slow) != 0ull)
        {
          distinct_symbols();
          exactDone = true;
          m = exact_index(m);
        }
        }
        mtf = m;
        const bool notInNow = m == KU;
        // (the flag matters only where the two answers differ: LUT: a run of exactly 3 + the range's penalty bytes)
        const bool changed = anyOpen && __ballot(isRun && !sure && notInNow != notIn && stored_with(inL, true) != stored_with(inL, false)) != 0ull;
        notIn = notInNow;
        wave_sync();
        if (!changed) break;
      }

      exactFirst = exactDone && dTrips <= 16u;

      // the list behind the round (only where another round follows): the round's last occurrences from the most recent down, then what is left of the old list
      if (r0 + 64u < R)
      {
        // (cheap first: the last 2 K + 2 entries -- stored symbols, then the list in front of the round -- usually hold K distinct ones.  Lane i takes the entry
        //  i places from the end and learns from the lanes below it whether its symbol has been seen: 2 K + 1 readlane trips for the wave instead of a walk
        //  in which every lane does the same 7-compare steps)
        const uint32_t nSt = wave_lane(wave_scan_add(k ? 1u : 0u), 63);
        constexpr uint32_t WIN = 2u * KU + 2u;
        const int32_t tMine = (int32_t)(KU + nSt) - 1 - (int32_t)lane;
        const bool inWin = manySymbols && lane < WIN && tMine >= 0;
        uint64_t hv = 0ull, newM = 0ull;
        bool dup = false;
        if (manySymbols)                                                     // (wave-uniform; data over a small alphabet goes straight to the exact sets below)
        {
          hv = inWin ? sh.symList[tMine] : 0ull;
#pragma unroll
          for (uint32_t j = 0; j + 1u < WIN; j++)
          {
            const uint64_t hj = (uint64_t)wave_lane((uint32_t)hv, (int)j) | ((uint64_t)wave_lane((uint32_t)(hv >> 32), (int)j) << 32);
            if (lane > j && hj == hv) dup = true;
          }
          newM = __ballot(inWin && !dup);
        }
        const uint32_t cntSeen = (uint32_t)__builtin_popcountll(newM);
        if (cntSeen >= KU)
        {
          const uint32_t rank = (uint32_t)__builtin_popcountll(newM & ((1ull << lane) - 1ull));
          wave_sync();
          if (inWin && !dup && rank < KU) sh.carList[KU - 1u - rank] = hv;   // (rank 0: the most recent)
          wave_sync();
        }
        else
        {
          if (!exactDone) distinct_symbols();
          uint32_t place, renewedBefore, renewedAll;
          list_place(place, renewedBefore, renewedAll);
          sh.symAll[lane] = sym;
          wave_sync();
          if (lane == 0u)
          {
            uint32_t filled = 0;
            uint64_t lm = lastAll;
            while (lm != 0ull && filled < KU)
            {
              const uint32_t j = 63u - (uint32_t)__builtin_clzll(lm);
              lm &= ~(1ull << j);
              sh.carList[KU - 1u - filled] = sh.symAll[j];
              filled++;
            }
            for (uint32_t i = 0; i < KU && filled < KU; i++)
              if (((renewedAll >> i) & 1u) == 0u) { sh.carList[KU - 1u - filled] = sh.symList[KU - 1u - i]; filled++; }
          }
          wave_sync();
        }
      }
    }
    const uint32_t count = e - p, gap = p - inL, rng = gap + 2u;

    // ---- packet header (rleX_Xsl.h:190-250): u16 {index, count, range}, [symbol if new], [u16 count], [u16 range] ----
    const uint32_t cfield = (AL && S != 1) ? count / SU - 3u / SU + 2u : count - 1u;
    const uint32_t cBytes = cfield <= MAXC ? 0u : 2u, sBytes = mtf == KU ? SU : 0u, rBytes = rng <= MAXR ? 0u : 2u;
    // Short: count field value (+ 2); the one-byte form where gap and count fit, else three bytes with 16 bit fields behind them where 9 / SRB bits are not enough
    [[maybe_unused]] const uint32_t scu = (AL && S != 1) ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u;
    [[maybe_unused]] const bool pack1 = gap <= TR::SMAXPR && scu - 2u <= TR::SMAXPC;
    const uint32_t hl = !k ? 0u : (SH ? (pack1 ? 1u : 3u + (scu > TR::SMAXTC ? 2u : 0u) + (rng > TR::SMAXTR ? 2u : 0u)) + sBytes : 2u + cBytes + sBytes + rBytes);
    const uint32_t myBytes = k ? hl + gap : 0u;
    const uint32_t incl = wave_scan_add(myBytes | (k ? 0x10000u : 0u));
    const uint32_t tot = wave_lane(incl, 63);
    if constexpr (MODE == 0)
    {
      const uint32_t idx = K_ + (incl >> 16) - 1u;
      if (k && idx < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + idx] = p | ((e - 1u) << 12) | (mtf << 24);
    }
    else
    {
      const uint32_t at0 = pos + (incl & 0xFFFFu) - myBytes;
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        uint32_t a = at0;
        if constexpr (SH)
        {
          // [list index | count | range] in one byte, or [index | all-ones count | 9 bit count | SRB bit range] in three (rleX_Xsl_short.h:216-357), then the symbol if new
          const uint32_t mi = mtf << (TR::SCB + TR::SRBP);
          if (pack1) { pp_or_bytes(sh.img, a, (uint64_t)(mi | ((scu - 2u) << TR::SRBP) | gap), 1u); a += 1u; }
          else
          {
            const uint32_t scx = scu <= TR::SMAXTC ? scu : 1u, rx = rng <= TR::SMAXTR ? rng : 1u;      // (1: a 16 bit field follows)
            const uint32_t f = scx << (TR::SRB - 8u);
            const uint32_t b0 = (mi | (TR::SCINV << TR::SRBP) | (f >> 8)) & 0xFFu, b1 = (f | (rx >> 8)) & 0xFFu, b2 = rx & 0xFFu;
            pp_or_bytes(sh.img, a, (uint64_t)(b0 | (b1 << 8) | (b2 << 16)), 3u); a += 3u;
            if (scx != scu) { pp_or_bytes(sh.img, a, (uint64_t)scu, 2u); a += 2u; }
            if (rx != rng) { pp_or_bytes(sh.img, a, (uint64_t)rng, 2u); a += 2u; }
          }
          if (mtf == KU) pp_or_bytes(sh.img, a, sym, SU);
        }
        else
        {
        const uint32_t c7 = cfield <= MAXC ? cfield : 1u, r7 = rng <= MAXR ? rng : 1u;   // (1: a 16 bit field follows; nothing in a block needs 32)
        pp_or_bytes(sh.img, a, (uint64_t)((mtf << MSH) | (c7 << RB) | r7), 2u); a += 2u;
        if (mtf == KU) { pp_or_bytes(sh.img, a, sym, SU); a += SU; }
        if (cBytes) { pp_or_bytes(sh.img, a, (uint64_t)cfield, 2u); a += 2u; }
        if (rBytes) pp_or_bytes(sh.img, a, (uint64_t)rng, 2u);
        }
        ds = at0 + hl;
        if (gap > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)inL | ((uint64_t)ds << 13) | ((uint64_t)gap << 26); }
        else if (gap != 0u) nch = ((ds + gap - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)
      {
        if (t < nch) pp_put_chunks(sh, inL, ds, gap, t, 1u, t + 1u);
        if (t + 1u < nch) pp_put_chunks(sh, inL, ds, gap, t + 1u, 1u, t + 2u);
      }
    }
    carL = wave_lane(outL, lastLane);
    pos += tot & 0xFFFFu;
    K_ += tot >> 16;
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  // ---- 3. terminator, stream size (rleX_Xsl.h:319-338) ----
  const uint32_t kLit = ended ? 0u : n - carL;
  const uint32_t streamSize = pos + (ended ? TERM_END : TERM) + kLit;
  if constexpr (MODE == 0)
  {
    if (lane == 0u) { sizes[b] = streamSize; sc.recCount[b] = (K_ <= sc.recStride) ? K_ : kPpNoRecords; }
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
      // end: u16 (1 << RB) | 1, u16 0, u16 0;  literals: u16 1 << RB, u16 0, u32 literals + 2
      if constexpr (SH)
      {
        // end: [all-ones count] [STB] 01, u16 0, u16 0;  literals: [all-ones count] [STB] 00, u16 0, u32 literals + 2  (rleX_Xsl_short.h:976-1032; a list: no symbol behind)
        pp_or_bytes(sh.img, pos, (uint64_t)((TR::SCINV << TR::SRBP) | (TR::STB << 8) | (ended ? 1u << 16 : 0u)), 3u);
        if (!ended) pp_or_bytes(sh.img, pos + 5u, (uint64_t)(kLit + 2u), 4u);
      }
      else
      {
        sh.img[pos] = (uint8_t)((1u << RB) | (ended ? 1u : 0u));
        if (!ended) pp_or_bytes(sh.img, pos + 4u, (uint64_t)(kLit + 2u), 4u);
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
        if (len != 0u) pp_put_chunks(sh, src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
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
__global__ __launch_bounds__(64) void k_encodeL_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                   const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpScratch sc)
{
  __shared__ PpLutShared<MODE != 0, ((FAM == LUT3 || FAM == SHORT3) ? 3 : 7)> sh;
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
    ppL_block<FAM, S, AL, MODE>(U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
  }
}

} // namespace hsrle
