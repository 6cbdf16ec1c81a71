// hsrle_encodeLpw.hip.h -- the position-parallel LUT encoder of hsrle_encodeLp.hip.h (every 7 symbol LUT codec, the 3 symbol LUT codecs of 1 / 2 byte symbols, the Short codecs
// with a 3 / 7 symbol list) for BLOCKS ABOVE 4 KiB: a block walked in windows of 4 KiB, as hsrle_encodeSpw.hip.h does it for the codecs of hsrle_encodeSp.hip.h.  Round 6.
//
// Replaces: src/rleX_Xsl.h:93-346, src/rleX_Xsl_multibyte_encoder.h:18-370, src/rleX_Xsl_short.h:152-372 as hsrle_encodeLp.hip.h does, for blocks above 4 KiB -- so far the
//           lane-per-block ring encoders + staging slots + k_compact.
//
// The list is a set of symbol VALUES that the block kernel already carries from one round of 64 candidates to the next (sh.carList): a window's edge is one more round
// boundary.  Pass 1 (one wave per block, window after window) keeps the list in LDS; the state record of a window holds it for pass 2 (one wave per window), which needs it
// only where a window stored more runs than its records hold and repeats the decisions.  Everything else -- the match bits' history, runs that begin in front of the
// window, the literal bytes in front of it -- is hsrle_encodeSpw.hip.h's.  The LUT / Short header forms choose their field widths by value (16 or 32 bits beyond the packed
// field); the host keeps these codecs to blocks below 1 MiB, where the reference's penalty thresholds (0xFFFFF: rleX_Xsl.h:130, rleX_Xsl_short.h:178) cannot be reached.
#pragma once

#include "hsrle_encodeLp.hip.h"
#include "hsrle_encodeSpw.hip.h"   // hsrle_encode8pw.hip.h: ppw_load, ppw_img16

namespace hsrle {

// (kPpwLStateWords = 32, hsrle_launch.h: posW, lastRLE, openStart, carE, the list x 2 K, stored runs, unit, window, start of the run that began in front of the window)

struct PpLwCarry { uint32_t carL, carE, pos, openStart; bool ended; };

template <int FAM, int S, int AL, int MODE>
__device__ __forceinline__ void ppLw_window(const uint8_t *__restrict__ d, uint32_t n, uint32_t wi, PpLwCarry &cs, uint32_t pCar, uint32_t *__restrict__ st, uint32_t *__restrict__ myRecs,
                                            uint32_t unit, uint32_t unitSize, uint8_t *__restrict__ dst, PpLutShared<MODE != 0, ((FAM == LUT3 || FAM == SHORT3) ? 3 : 7)> &sh,
                                            const u32x4 (&x)[4], uint32_t recN, uint32_t rec0)
{
  static_assert(FAM == LUT3 || FAM == LUT7 || FAM == SHORT3 || FAM == SHORT7, "3 / 7 symbol LUT, or the Short family with a 3 / 7 symbol list");
  static_assert(S == 1 || S == 2 || S == 3 || S == 4 || S == 6 || S == 8, "symbols of 1, 2, 3, 4, 6 or 8 bytes");
  static_assert(S != 1 || AL == 0, "8 bit: byte-aligned by nature");
  constexpr bool SH = FAM == SHORT3 || FAM == SHORT7;
  using TR = Traits<FAM, S, AL>;
  constexpr int K = (FAM == LUT3 || FAM == SHORT3) ? 3 : 7;
  constexpr uint32_t KU = (uint32_t)K;
  constexpr uint32_t SU = (uint32_t)S;
  constexpr uint32_t RB = K == 3 ? 7u : 6u, MAXR = (1u << RB) - 1u, MAXC = 127u, MSH = K == 3 ? 14u : 13u;
  constexpr bool NARROW = SH ? S <= 4 : S <= 2;
  constexpr uint32_t TERM = SH ? 9u : 8u, TERM_END = SH ? 7u : 6u, HDR = 8u;
  constexpr uint32_t SURE = SH ? (TR::SMINS + SU + 4u < TR::SMINL ? TR::SMINS + SU + 4u : TR::SMINL) : 6u;
  const uint32_t lane = threadIdx.x;
  const uint32_t ws = wi * kPpwWindow;
  const uint32_t base = lane * 64u;                    // (window relative)
  const bool lastWindow = ws + kPpwWindow >= n;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const bool fromRecs = MODE == 1 && recN != kPpNoRecords;
  const uint32_t carL0 = cs.carL, pos0 = (wi == 0u) ? 0u : cs.pos, openIn = cs.openStart;

  if constexpr (MODE == 0)
  {
    if (wi == 0u && lane < KU)
    {
      // the list every stream starts with (rleX_Xsl.h:279-287): 00, 7F, FF, 01, 7E, 80, FE in every symbol byte; carList holds it least recent first
      constexpr uint64_t SMASK = (S >= 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);
      const uint32_t i = KU - 1u - lane;
      const uint64_t v = i == 0u ? 0x00ull : (i == 1u ? 0x7Full : (i == 2u ? 0xFFull : (i == 3u ? 0x01ull : (i == 4u ? 0x7Eull : (i == 5u ? 0x80ull : 0xFEull)))));
      sh.carList[lane] = (v * 0x0101010101010101ull) & SMASK;
    }
    wave_sync();
    // the state in front of this window
    if (lane == 0u) { st[0] = pos0; st[1] = cs.carL; st[2] = openIn; st[3] = cs.carE; st[4 + 2u * KU + 1u] = unit; st[4 + 2u * KU + 2u] = wi; }
    if (lane < KU) { const uint64_t v = sh.carList[lane]; st[4u + 2u * lane] = (uint32_t)v; st[5u + 2u * lane] = (uint32_t)(v >> 32); }
  }
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
    if (lane < KU) sh.carList[lane] = (uint64_t)st[4u + 2u * lane] | ((uint64_t)st[5u + 2u * lane] << 32);   // (st: the window's state record)
  }
  // the input image: the window, the 16 bytes in front of it and the 16 + 16 behind it (hsrle_encodeSpw.hip.h)
  u32x4 front = zero4, back0 = zero4;
  if (lane == 0u && ws >= 16u) front = ld128(d + ws - 16u);
  if (lane == 63u)
  {
    const uint32_t bp = ws + kPpwWindow;
    if (bp + 16u <= n) back0 = ld128(d + bp);
    else if (bp < n) back0 = load16_edge(d, (int64_t)bp, (uint64_t)n);
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);
  if (lane == 0u) lds_st128(sh.inb, front);
  if (lane == 63u) { lds_st128(sh.inb + kPpInPad + kPpMaxBlock, back0); lds_st128(sh.inb + kPpInPad + kPpMaxBlock + 16u, zero4); }

  // the S-byte symbol of the run that starts at p (absolute) and reaches into this window
  auto symbol_at = [&](uint32_t p) __attribute__((always_inline)) -> uint64_t {
    uint32_t a = p;
    if (ws >= 16u && p < ws - 16u) a = p + ((ws - 16u - p + SU - 1u) / SU) * SU;      // (a later period of the run)
    return pp_symbol<S>(sh.inb, kPpInPad + a - ws);
  };

  // ---- 1. match bits m[j] = (d[j] == d[j + S]), stretches, candidates (hsrle_encodeSpw.hip.h) ----
  uint32_t R = recN;
  uint64_t candLeft = 0;
  uint32_t candAt = 0;
  if (!fromRecs)
  {
    uint32_t wd[18];
#pragma unroll
    for (int j = 0; j < 4; j++) { wd[4 * j] = x[j].x; wd[4 * j + 1] = x[j].y; wd[4 * j + 2] = x[j].z; wd[4 * j + 3] = x[j].w; }
    wd[16] = wave_shl1(x[0].x, back0.x);
    wd[17] = wave_shl1(x[0].y, back0.y);
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
        if constexpr (S == 1) s = alignbyte(wd[i + 1], wd[i], 1);
        else if constexpr (S == 2) s = alignbyte(wd[i + 1], wd[i], 2);
        else if constexpr (S == 3) s = alignbyte(wd[i + 1], wd[i], 3);
        else if constexpr (S == 4) s = wd[i + 1];
        else if constexpr (S == 6) s = alignbyte(wd[i + 2], wd[i + 1], 2);
        else s = wd[i + 2];
        t[k] = wd[i] ^ s;
      }
      m64 |= (uint64_t)zero_mask16(t[0], t[1], t[2], t[3]) << (16 * j);
    }
    const int64_t vb = (int64_t)n - (int64_t)SU - (int64_t)(ws + base);
    const uint32_t validBits = vb <= 0 ? 0u : (vb >= 64 ? 64u : (uint32_t)vb);
    m64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    // lane 0: the match bits of the 8 positions in front of the window, in the top byte of a dword
    uint32_t histM = 0;
    if (lane == 0u && ws >= 16u)
    {
      const uint64_t f8 = (uint64_t)front.z | ((uint64_t)front.w << 32), a8 = (uint64_t)x[0].x | ((uint64_t)x[0].y << 32), b8 = (uint64_t)x[0].z | ((uint64_t)x[0].w << 32);
#pragma unroll
      for (uint32_t i = 0; i < 8u; i++)
      {
        const uint32_t j = i + SU;
        const uint32_t lhs = (uint32_t)(f8 >> (8u * i)) & 0xFFu;
        const uint32_t rhs = (j < 8u) ? (uint32_t)(f8 >> (8u * j)) & 0xFFu : ((j < 16u) ? (uint32_t)(a8 >> (8u * (j - 8u))) & 0xFFu : (uint32_t)(b8 >> (8u * (j - 16u))) & 0xFFu);
        if (lhs == rhs && ws - 8u + i + SU < n) histM |= 1u << (24u + i);
      }
    }
    const uint32_t histC2 = histM & (histM << 1), histC4 = histC2 & (histC2 << 2);
    const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(m64 >> 63), histM >> 31);
    const uint64_t prev = (m64 << 1) | carry;
    const uint64_t starts = m64 & ~prev;
    const uint64_t ends = ~m64 & prev;
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const int32_t inclStart = wave_scan_max(ownStart);
    const uint32_t carryStart = wave_shr1((uint32_t)inclStart, 0xFFFFFFFFu);
    auto shl_in = [&](uint64_t v, uint32_t t, uint32_t hist) __attribute__((always_inline)) -> uint64_t {
      const uint32_t top = wave_shr1((uint32_t)(v >> 32), hist);
      return (v << t) | (uint64_t)(top >> (32u - t));
    };
    // candidates: the stretches of at least S bits; 8 bit: of at least 2 (Short: of 1 -- a run of two bytes of a listed symbol right behind the run before is stored)
    uint64_t full;
    uint32_t histFull;
    {
      const uint64_t c2 = m64 & shl_in(m64, 1u, histM);
      if constexpr (S == 1 && SH) { full = m64; histFull = histM; }
      else if constexpr (S <= 2) { full = c2; histFull = histC2; }
      else if constexpr (S == 3) { full = c2 & shl_in(m64, 2u, histM); histFull = histC2 & (histM << 2); }
      else
      {
        const uint64_t c4 = c2 & shl_in(c2, 2u, histC2);
        if constexpr (S == 4) { full = c4; histFull = histC4; }
        else if constexpr (S == 6) { full = c4 & shl_in(c2, 4u, histC2); histFull = histC4 & (histC2 << 4); }
        else { full = c4 & shl_in(c4, 4u, histC4); histFull = histC4 & (histC4 << 4); }
      }
    }
    const uint64_t cands = ends & shl_in(full, 1u, histFull);
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    candLeft = cands;
    candAt = inclCnt - cnt;
    if constexpr (MODE == 0)
    {
      const int32_t lastStart = (int32_t)wave_lane((uint32_t)inclStart, 63);
      if (lastStart >= 0) cs.openStart = ws + (uint32_t)lastStart;
    }
  }
  wave_sync();

  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t carL = cs.carL, carE = cs.carE;
  bool exactFirst = false;
  uint32_t imgPos = (wi == 0u) ? HDR : 0u;            // MODE 1: image position of the round's first packet
  uint32_t pos = cs.pos;                                // MODE 0: stream position
  uint32_t K_ = 0;                                      // stored runs of this window
  bool ended = cs.ended;
  uint32_t pCarOut = 0xFFFFFFFFu;
  [[maybe_unused]] bool firstSeen = false;
  [[maybe_unused]] uint32_t hlFirst = 0, extLen = 0, imgShift = 0;    // imgShift: the image starts this many bytes in, so that what follows the first header is 16-byte aligned in LDS
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
      const uint32_t rec = (r0 == 0u) ? rec0 : (have ? myRecs[r0 + lane] : 0u);
      e = ws + ((rec >> 12) & 0x1FFFu) + 1u;
      p = ((rec >> 28) & 1u) ? pCar : ws + (rec & 0xFFFu);
      mtf = (rec >> 25) & 7u;
      k = have ? 1 : 0;
      outL = e;
      inL = wave_shr1(outL, carL);
      sym = symbol_at(have ? p : ws);
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
      const uint32_t qr = have ? (uint32_t)sh.lst[lane] : 0u;              // (window relative)
      wave_sync();
      const uint32_t q = ws + qr;
      uint32_t s0 = openIn;
      if (qr != 0u)
      {
        const uint32_t qm = qr - 1u, iq = qm >> 6, bit = qm & 63u;
        const uint64_t stq = sh.starts[iq];
        const uint32_t cst = (uint32_t)sh.carryStart[iq];
        const uint64_t sBelow = stq & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
        s0 = (sBelow != 0ull) ? ws + (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : (cst != 0xFFFFu ? ws + cst : openIn);
      }

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
      sym = symbol_at(isRun ? p : ws);

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

      // the list behind the round (only where another round -- or, pass 1, another window -- follows): the round's last occurrences from the most recent down, then what is left of the old list
      if (r0 + 64u < R || (MODE == 0 && !lastWindow))
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
    const uint32_t inLw = inL > ws ? inL : ws, pw = p > ws ? p : ws;     // the part of the literal stretch that lies in this window
    const uint32_t gapImg = pw - inLw;

    // ---- packet header (rleX_Xsl.h:190-250): u16 {index, count, range}, [symbol if new], [u16 / u32 count], [u16 / u32 range] ----
    const uint32_t cfield = (AL && S != 1) ? count / SU - 3u / SU + 2u : count - 1u;
    const uint32_t cBytes = cfield <= MAXC ? 0u : (cfield <= 0xFFFFu ? 2u : 4u), sBytes = mtf == KU ? SU : 0u, rBytes = rng <= MAXR ? 0u : (rng <= 0xFFFFu ? 2u : 4u);
    [[maybe_unused]] const uint32_t scu = (AL && S != 1) ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u;
    [[maybe_unused]] const bool pack1 = gap <= TR::SMAXPR && scu - 2u <= TR::SMAXPC;
    const uint32_t hl = !k ? 0u : (SH ? (pack1 ? 1u : 3u + (scu > TR::SMAXTC ? (scu <= 0xFFFFu ? 2u : 4u) : 0u) + (rng > TR::SMAXTR ? (rng <= 0xFFFFu ? 2u : 4u) : 0u)) + sBytes
                                      : 2u + cBytes + sBytes + rBytes);
    if constexpr (MODE == 0)
    {
      const uint32_t inclB = wave_scan_add(k ? hl + gap : 0u), inclK = wave_scan_add(k ? 1u : 0u);
      const uint32_t idx = K_ + inclK - 1u;
      if (k && idx < kPpwStride) myRecs[idx] = (p >= ws ? p - ws : 0u) | ((e - 1u - ws) << 12) | (mtf << 25) | (p < ws ? 1u << 28 : 0u);
      const uint64_t carried = __ballot(k != 0 && p < ws);
      if (carried != 0ull) pCarOut = wave_lane(p, (int)__builtin_ctzll(carried));
      pos += wave_lane(inclB, 63);
      K_ += wave_lane(inclK, 63);
    }
    else
    {
      const uint32_t myBytes = k ? hl + gapImg : 0u;
      const uint32_t incl = wave_scan_add(myBytes);
      if (!firstSeen)
      {
        const uint64_t stored = __ballot(k != 0);
        if (stored != 0ull)
        {
          const int fl = (int)__builtin_ctzll(stored);
          firstSeen = true;
          hlFirst = wave_lane(hl, fl);
          extLen = wave_lane(gap - gapImg, fl);
          // (literals in front of the window: the image is written in two pieces around them -- the second one starts on a 16-byte boundary of the LDS image)
          if (extLen != 0u) { imgShift = (16u - hlFirst) & 15u; imgPos += imgShift; }
        }
      }
      const uint32_t at0 = imgPos + incl - myBytes;
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        uint32_t a = at0;
        if constexpr (SH)
        {
          const uint32_t mi = mtf << (TR::SCB + TR::SRBP);
          if (pack1) { pp_or_bytes(sh.img, a, (uint64_t)(mi | ((scu - 2u) << TR::SRBP) | gap), 1u); a += 1u; }
          else
          {
            const uint32_t scx = scu <= TR::SMAXTC ? scu : (scu <= 0xFFFFu ? 1u : 0u), rx = rng <= TR::SMAXTR ? rng : (rng <= 0xFFFFu ? 1u : 0u);      // (1: a 16 bit field follows, 0: a 32 bit one)
            const uint32_t f = scx << (TR::SRB - 8u);
            const uint32_t b0 = (mi | (TR::SCINV << TR::SRBP) | (f >> 8)) & 0xFFu, b1 = (f | (rx >> 8)) & 0xFFu, b2 = rx & 0xFFu;
            pp_or_bytes(sh.img, a, (uint64_t)(b0 | (b1 << 8) | (b2 << 16)), 3u); a += 3u;
            if (scx != scu) { const uint32_t nb = scu <= 0xFFFFu ? 2u : 4u; pp_or_bytes(sh.img, a, (uint64_t)scu, nb); a += nb; }
            if (rx != rng) { const uint32_t nb = rng <= 0xFFFFu ? 2u : 4u; pp_or_bytes(sh.img, a, (uint64_t)rng, nb); a += nb; }
          }
          if (mtf == KU) pp_or_bytes(sh.img, a, sym, SU);
        }
        else
        {
          const uint32_t c7 = cfield <= MAXC ? cfield : (cfield <= 0xFFFFu ? 1u : 0u), r7 = rng <= MAXR ? rng : (rng <= 0xFFFFu ? 1u : 0u);
          pp_or_bytes(sh.img, a, (uint64_t)((mtf << MSH) | (c7 << RB) | r7), 2u); a += 2u;
          if (mtf == KU) { pp_or_bytes(sh.img, a, sym, SU); a += SU; }
          if (cBytes) { pp_or_bytes(sh.img, a, (uint64_t)cfield, cBytes); a += cBytes; }
          if (rBytes) pp_or_bytes(sh.img, a, (uint64_t)rng, rBytes);
        }
        ds = at0 + hl;
        if (gapImg > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)(inLw - ws) | ((uint64_t)ds << 13) | ((uint64_t)gapImg << 26); }
        else if (gapImg != 0u) nch = ((ds + gapImg - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)
      {
        if (t < nch) pp_put_chunks(sh, inLw - ws, ds, gapImg, t, 1u, t + 1u);
        if (t + 1u < nch) pp_put_chunks(sh, inLw - ws, ds, gapImg, t + 1u, 1u, t + 2u);
      }
      imgPos += wave_lane(incl, 63);
    }
    carL = wave_lane(outL, lastLane);
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  if constexpr (MODE == 0)
  {
    if (lane == 0u) { st[4u + 2u * KU] = (K_ <= kPpwStride) ? K_ : kPpNoRecords; st[4u + 2u * KU + 3u] = pCarOut; }
    cs.carL = carL; cs.carE = carE; cs.pos = pos; cs.ended = ended;
    return;
  }
  else
  {
    // ---- 3. header, terminator + trailing literals (the last window's packet; rleX_Xsl.h:319-338, rleX_Xsl_short.h:976-1032) ----
    uint32_t imgSize = imgPos;
    if (wi == 0u && lane < 8u)
    {
      const uint64_t h = (uint64_t)n | ((uint64_t)unitSize << 32);
      sh.img[lane] = (uint8_t)(h >> (8u * lane));
    }
    uint32_t tailSrc = 0, tailLen = 0;
    if (lastWindow)
    {
      const uint32_t kLit = ended ? 0u : n - carL;
      if (!ended)
      {
        const uint32_t from = carL > ws ? carL : ws;
        tailSrc = from - ws; tailLen = n - from;
        if (!firstSeen)
        {
          // (no stored run in this window: the terminator is its first "header", the trailing literals may begin in front of the window)
          firstSeen = true; hlFirst = TERM; extLen = from - carL;
          if (extLen != 0u) { imgShift = (16u - TERM) & 15u; imgPos += imgShift; }
        }
      }
      if (lane == 16u)
      {
        if constexpr (SH)
        {
          pp_or_bytes(sh.img, imgPos, (uint64_t)((TR::SCINV << TR::SRBP) | (TR::STB << 8) | (ended ? 1u << 16 : 0u)), 3u);
          if (!ended) pp_or_bytes(sh.img, imgPos + 5u, (uint64_t)(kLit + 2u), 4u);
        }
        else
        {
          sh.img[imgPos] = (uint8_t)((1u << RB) | (ended ? 1u : 0u));
          if (!ended) pp_or_bytes(sh.img, imgPos + 4u, (uint64_t)(kLit + 2u), 4u);
        }
      }
      imgSize = imgPos + (ended ? TERM_END : TERM) + tailLen;
    }
    wave_sync();
    {
      const uint32_t nj = sh.jobCount;
      for (uint32_t j = 0; j <= nj; j++)
      {
        uint32_t src, ds, len;
        if (j < nj) { const uint64_t jb = sh.jobs[j]; src = (uint32_t)jb & 0x1FFFu; ds = (uint32_t)(jb >> 13) & 0x1FFFu; len = (uint32_t)(jb >> 26); }
        else { src = tailSrc; ds = imgPos + TERM; len = tailLen; }
        if (len != 0u) pp_put_chunks(sh, src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
      }
    }
    wave_sync();

    // ---- 4. the image leaves LDS once; the literals in front of the window go from the input straight to their place behind the first header ----
    uint8_t *const out = dst + pos0;
    if (extLen == 0u)
    {
      const uint32_t nFull = imgSize >> 4, tail = imgSize & 15u;
      for (uint32_t c = lane; c < nFull; c += 64u)
        st128(out + 16u * c, lds_ld128(sh.img + 16u * c));
      if (lane < tail) out[16u * nFull + lane] = sh.img[16u * nFull + lane];
    }
    else
    {
      if (lane < hlFirst) out[lane] = sh.img[imgShift + lane];
      {
        const uint8_t *const src = d + carL0;
        uint8_t *const to = out + hlFirst;
        const uint32_t nFull = extLen >> 4, tail = extLen & 15u;
        for (uint32_t c = lane; c < nFull; c += 64u)
          st128(to + 16u * c, ld128(src + 16u * c));
        if (lane < tail) to[16u * nFull + lane] = src[16u * nFull + lane];
      }
      {
        uint8_t *const to = out + hlFirst + extLen;
        const uint32_t from = imgShift + hlFirst;                              // (a multiple of 16)
        const uint32_t rest = imgSize - from;
        const uint32_t nFull = rest >> 4, tail = rest & 15u;
        for (uint32_t c = lane; c < nFull; c += 64u)
          st128(to + 16u * c, lds_ld128(sh.img + from + 16u * c));
        if (lane < tail) to[16u * nFull + lane] = sh.img[from + 16u * nFull + lane];
      }
    }
  }
}

// Pass 1: one wave per block
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encodeL_ppw_scan(PpwArgs a)
{
  constexpr int K = (FAM == LUT3 || FAM == SHORT3) ? 3 : 7;
  constexpr bool SH = FAM == SHORT3 || FAM == SHORT7;
  constexpr uint32_t TERM = SH ? 9u : 8u, TERM_END = SH ? 7u : 6u;
  __shared__ PpLutShared<false, K> sh;
  const uint32_t u = xcd_tile(blockIdx.x, gridDim.x);
  if (u >= a.nUnits) return;
  const uint64_t at = (uint64_t)u * a.B;
  const uint8_t *const d = a.in + at;
  const uint32_t n = (uint32_t)((a.U - at) < (uint64_t)a.B ? (a.U - at) : (uint64_t)a.B);
  const uint32_t slots = (a.B + kPpwWindow - 1u) / kPpwWindow, windows = (n + kPpwWindow - 1u) / kPpwWindow;
  const uint64_t gw0 = (uint64_t)u * slots;
  PpLwCarry cs;
  cs.carL = 0u; cs.carE = 0u; cs.pos = 8u; cs.openStart = 0u; cs.ended = false;
  u32x4 x[4], xn[4];
  ppw_load(d, n, 0u, x);
  for (uint32_t w = 0; w < windows; w++)
  {
    if (w + 1u < windows) ppw_load(d, n, (w + 1u) * kPpwWindow, xn);
    ppLw_window<FAM, S, AL, 0>(d, n, w, cs, 0u, a.states + (gw0 + w) * kPpwLStateWords, a.recs + (gw0 + w) * kPpwStride, u, 0u, nullptr, sh, x, kPpNoRecords, 0u);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = xn[j];
  }
  if (threadIdx.x == 0u) a.sizes[u] = cs.pos + (cs.ended ? TERM_END : TERM + (n - cs.carL));
}

// Pass 2: one wave per window
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encodeL_ppw_emit(PpwArgs a)
{
  constexpr int K = (FAM == LUT3 || FAM == SHORT3) ? 3 : 7;
  __shared__ PpLutShared<true, K> sh;
  if (threadIdx.x < 17u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (c == 16u) ? ~0u : (uint32_t)part, p1 = (c == 16u) ? ~0u : (uint32_t)(part >> 32);
    lds_st128(sh.mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }
  const uint32_t gw = xcd_tile(blockIdx.x, gridDim.x);
  if (gw >= a.nWindows) return;
  // (which block and window this is follows from the window's number: the state, the first records and the input are asked for together)
  uint32_t *const st = a.states + (uint64_t)gw * kPpwLStateWords;
  const uint32_t *const myRecs = a.recs + (uint64_t)gw * kPpwStride;
  const uint32_t slots = (a.B + kPpwWindow - 1u) / kPpwWindow;
  const uint32_t u = gw / slots, w = gw - u * slots;
  const uint64_t at = (uint64_t)u * a.B;
  const uint8_t *const d = a.in + at;
  const uint32_t n = (uint32_t)((a.U - at) < (uint64_t)a.B ? (a.U - at) : (uint64_t)a.B);
  if (w * kPpwWindow >= n) return;
  const uint32_t sv = (threadIdx.x < kPpwLStateWords) ? st[threadIdx.x] : 0u;
  const uint32_t rec0 = myRecs[threadIdx.x];
  const uint32_t unitSize = a.sizes[u];
  const uint64_t unitAt = a.offsets[u];
  u32x4 x[4];
  ppw_load(d, n, w * kPpwWindow, x);
  const uint32_t recN = wave_lane(sv, 4 + 2 * K);
  PpLwCarry cs;
  cs.pos = wave_lane(sv, 0); cs.carL = wave_lane(sv, 1); cs.openStart = wave_lane(sv, 2); cs.carE = wave_lane(sv, 3);
  cs.ended = false;
  wave_sync();
  ppLw_window<FAM, S, AL, 1>(d, n, w, cs, wave_lane(sv, 4 + 2 * K + 3), st, const_cast<uint32_t *>(myRecs), u, unitSize, a.payload + unitAt, sh, x, recN, rec0);
}

} // namespace hsrle
