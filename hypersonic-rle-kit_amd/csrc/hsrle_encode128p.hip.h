// hsrle_encode128p.hip.h -- POSITION-PARALLEL encoder for the four 128 bit codecs (rle128_sym, rle128_sym_packed, rle128_byte, rle128_byte_packed), blocks
// of at most 4 KiB: one wave per block, the payload written once (round 6).  hsrle_encodeSp.hip.h with the 128 bit scanner's own rules.
//
// Replaces: src/rle128_extreme_cpu.h:32-497 (extension :79-228, emit rule, pair search :233-268, byte steps :270-300, terminators) -- and, in this library,
//           the lane-per-block k_encode128_blocks + staging slots + k_compact for containers of these codecs (hsrle_encode128.hip.h).
//
// On bits E[j] = (d[j] == d[j + 16]) the reference's walk has a closed form (checked against the oracle on the CPU: tools/rle128_pp_model.py):
//   * the block starts INSIDE a run of its first 16 bytes (count 0, no pair needed): a run at p = 0 whatever E says;
//   * the pair search hops behind the highest clear bit of its 16-bit window, so it stops at the first p >= (end of the run before) with E[p .. p + 16) all set
//     -- as long as p < n - 32.  A maximal stretch of >= 16 set bits gives at most one run; where it starts depends on the run before only if that run's
//     last symbol reaches into the stretch (the same DPP passes as hsrle_encodeSp.hip.h);
//   * the run takes whole symbols while they repeat and the position stays in front of n - 16, and -- byte-aligned codecs, position still in front of n - 16 --
//     the matching leading bytes of the next one;
//   * every run the pair search finds has >= 32 bytes >= LONG: it is stored whatever the state, only its header form (range, same symbol) depends on the run
//     before.  The leading run (16 .. 31 bytes) is the one candidate that may stay literals;
//   * behind the last run the hops are replayed (wave-uniform, on the bits in LDS) to the first position >= n - 32.  If that is n - 32 exactly and a pair
//     starts there, it is a last run of 32 bytes.  Else the reference steps byte by byte with a symbol it re-reads at every step, and "extends" it by the
//     bytes that equal their left neighbour: the Packed codecs store such a `run` of 3 .. 16 bytes when the 16 bytes at the step are the last stored
//     symbol again (:270-300 with the emit rule's same-symbol clause) -- rare, serial, at most four of them.
// MODE 0 leaves sizes[b] and one record per stored run, MODE 1 writes the stream from the records (a block never has more runs than records: >= 32 bytes per
// run of the body, at most six others).
#pragma once

#include "hsrle_encode8sp.hip.h"   // pp_put_chunks; hsrle_encodeSp.hip.h: pp_symbol, pp_or_bytes

namespace hsrle {

template <bool EMIT>
struct Pp128Shared
{
  uint8_t img[EMIT ? (kPpMaxBlock + 193u + 15u + 16u + 15u) / 16u * 16u : 16u] __attribute__((aligned(16)));
  uint8_t inb[kPpInPad + kPpMaxBlock + 32u] __attribute__((aligned(16)));
  uint8_t mlut[EMIT ? 17u * 16u : 16u] __attribute__((aligned(16)));
  uint64_t starts[64];
  uint64_t mbits[66];                                                   // E bits, two zero words behind them
  uint64_t jobs[EMIT ? kPpJobs : 1u];
  uint16_t lst[64];
  uint16_t carryStart[64];
  uint32_t jobCount;
};

// the 16 bytes at LDS byte position `at` of the input image
__device__ __forceinline__ u32x4 pp_sym16(const uint8_t *inb, uint32_t at)
{
  const uint32_t *const w = (const uint32_t *)(inb + (at & ~3u));
  const uint32_t sb = at & 3u;
  const uint32_t q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3], q4 = w[4];
  return u32x4{ alignbyte(q1, q0, sb), alignbyte(q2, q1, sb), alignbyte(q3, q2, sb), alignbyte(q4, q3, sb) };
}

template <bool PK, int AL, int MODE>
__device__ __forceinline__ void pp128_block(uint64_t U, uint32_t B, uint32_t b, uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload,
                                            const PpScratch &sc, Pp128Shared<MODE != 0> &sh, const u32x4 (&x)[4], uint32_t rec0)
{
  using TR = Traits<PK ? PACKED : PLAIN, 16, AL>;
  constexpr uint32_t SU = 16u;
  constexpr uint32_t SHORT = TR::SHORT, MEDIUM = TR::MEDIUM, LONG = TR::LONG, MAXR = TR::MAXRANGE;
  constexpr bool R7 = TR::kRange7;
  constexpr uint32_t TERM = (PK ? 5u : SU + 5u) + (R7 ? 4u : 5u);      // literal terminator's fixed part
  constexpr uint32_t TERM_END = (PK ? 5u : SU + 5u) + 5u;              // the end terminator always has the plain range field
  constexpr uint32_t HDR = 8u;
  constexpr uint32_t cMax = PK ? 127u : 255u;
  constexpr uint32_t NOSYM = 0x2000u;                                   // "no run stored yet": the last symbol is sixteen zero bytes
  const uint32_t lane = threadIdx.x;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint32_t base = lane * 64u;
  const int32_t T = (int32_t)n - 32;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const uint8_t *const bytes = sh.inb + kPpInPad;

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

  auto cfield_of = [&](uint32_t count) __attribute__((always_inline)) -> uint32_t { return AL ? count / SU - SHORT / SU + 1u : count - SHORT + 1u; };

  if constexpr (MODE == 0)
  {
    // ---- 1. E bits, stretches, candidates ----
    uint32_t w[20];
#pragma unroll
    for (int j = 0; j < 4; j++) { w[4 * j] = x[j].x; w[4 * j + 1] = x[j].y; w[4 * j + 2] = x[j].z; w[4 * j + 3] = x[j].w; }
    w[16] = wave_shl1(x[0].x, 0u); w[17] = wave_shl1(x[0].y, 0u); w[18] = wave_shl1(x[0].z, 0u); w[19] = wave_shl1(x[0].w, 0u);
    uint64_t m64 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      m64 |= (uint64_t)zero_mask16(w[4 * j] ^ w[4 * j + 4], w[4 * j + 1] ^ w[4 * j + 5], w[4 * j + 2] ^ w[4 * j + 6], w[4 * j + 3] ^ w[4 * j + 7]) << (16 * j);
    const int32_t vb = (int32_t)n - 16 - (int32_t)base;                    // position j matches only if j + 16 < n
    const uint32_t validBits = vb <= 0 ? 0u : (vb >= 64 ? 64u : (uint32_t)vb);
    m64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(m64 >> 63), 0u);
    const uint64_t prev = (m64 << 1) | carry;
    const uint64_t starts = m64 & ~prev;
    const uint64_t ends = ~m64 & prev;                                     // bit i: a stretch of set bits ends in front of position base + i
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
    auto shl_in = [&](uint64_t v, uint32_t t) __attribute__((always_inline)) -> uint64_t {
      const uint32_t top = wave_shr1((uint32_t)(v >> 32), 0u);
      return (v << t) | (uint64_t)(top >> (32u - t));
    };
    const uint64_t c2 = m64 & shl_in(m64, 1u);
    const uint64_t c4 = c2 & shl_in(c2, 2u);
    const uint64_t c8 = c4 & shl_in(c4, 4u);
    const uint64_t full = c8 & shl_in(c8, 8u);                              // bit i: the 16 bits up to and including position i are all set
    const uint64_t cands = ends & shl_in(full, 1u);
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    const uint32_t R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    sh.mbits[lane] = m64;
    if (lane < 2u) sh.mbits[64u + lane] = 0ull;
    uint64_t candLeft = cands;
    uint32_t candAt = inclCnt - cnt;
    // the leading run where it is not an ordinary candidate (fewer than 16 set bits at position 0, or no room for a pair search at all)
    const uint32_t L0 = wave_lane((m64 == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~m64), 0);
    const bool lead = n > 16u && (L0 < 16u || n <= 32u);
    const uint32_t leadE = (n > 32u && !AL) ? 16u + L0 : 16u;
    wave_sync();

    // 16 E bits from position w on (wave-uniform)
    auto ebits16 = [&](uint32_t wp) __attribute__((always_inline)) -> uint32_t {
      const uint32_t wd = wp >> 6, ws = wp & 63u;
      const uint64_t b0 = sh.mbits[wd], b1 = sh.mbits[wd + 1u];
      return (uint32_t)(ws ? (b0 >> ws) | (b1 << (64u - ws)) : b0) & 0xFFFFu;
    };
    // is the symbol at position a the last stored one?  st: lastRLE | position of the last stored run << 13 | NOSYM << 13
    auto same_as = [&](const u32x4 &sy, uint32_t st) __attribute__((always_inline)) -> bool {
      const bool none = (st & (NOSYM << 13)) != 0u;
      const u32x4 o = pp_sym16(sh.inb, kPpInPad + (none ? 0u : (st >> 13) & 0x1FFFu));
      const u32x4 c = none ? zero4 : o;
      return ((sy.x ^ c.x) | (sy.y ^ c.y) | (sy.z ^ c.z) | (sy.w ^ c.w)) == 0u;
    };

    // ---- 2. one candidate per lane, 64 per round ----
    uint32_t carS = NOSYM << 13;                       // lastRLE = 0, no symbol stored
    uint32_t carE = 0;                                 // end of the last run found: where the search resumes
    uint32_t pos = HDR;
    uint32_t K = 0;
    uint32_t nextCand = 0, keep = lead ? 1u : 0u;      // (the leading run takes lane 0 of the first round)
    bool leadRound = lead;
    for (;;)
    {
      const uint32_t fresh = (R - nextCand) < (64u - keep) ? (R - nextCand) : (64u - keep);
      const uint32_t inRound = keep + fresh;
      if (inRound == 0u) break;
      const bool have = lane < inRound;
      const int lastLane = (int)inRound - 1;
      while (candLeft != 0ull && candAt < nextCand + fresh)
      {
        sh.lst[keep + candAt - nextCand] = (uint16_t)(base + (uint32_t)__builtin_ctzll(candLeft));
        candAt++;
        candLeft &= candLeft - 1ull;
      }
      wave_sync();
      const bool isLead = leadRound && lane == 0u;
      const uint32_t q = isLead ? leadE - SU : (have ? (uint32_t)sh.lst[lane] : 0u);   // the stretch's bits end in front of position q
      wave_sync();
      const uint32_t qm = (have && !isLead) ? q - 1u : 0u, iq = qm >> 6, bit = qm & 63u;
      const uint64_t st = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      const uint32_t s0 = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;

      // -- where does the run start and end?
      uint32_t p = 0, e = 0;
      auto run_from = [&](uint32_t resume) __attribute__((always_inline)) {
        const uint32_t ps = resume > s0 ? resume : s0;
        p = ps; e = 0u;
        if (have && q >= ps + SU && (int32_t)ps < T)
        {
          const uint32_t L = q - ps;
          const uint32_t iL = ps + SU * (L / SU + 1u);                      // first position of the walk that the stretch does not cover
          const int32_t room = (int32_t)n - 16 - (int32_t)(ps + 32u);
          const uint32_t iN = ps + 32u + (room > 0 ? (((uint32_t)room + 15u) & ~15u) : 0u);   // first position of the walk at or beyond n - 16
          const uint32_t iS = iL < iN ? iL : iN;
          e = (!AL && (int32_t)iS < (int32_t)n - 16) ? q + SU : iS;
        }
      };
      const uint32_t qLeft = wave_shr1(have ? q : 0u, 0u);
      bool geoKnown = !have || isLead || (lane != 0u ? s0 >= qLeft + SU : s0 >= carE);
      uint32_t outE = 0;
      bool outEKnown = false;
      if (isLead) { p = 0u; e = leadE; outE = e; outEKnown = true; }
      else if (geoKnown) { run_from(0u); outE = e; outEKnown = have && e != 0u; }
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
      const u32x4 sym = pp_sym16(sh.inb, kPpInPad + (isRun ? p : 0u));

      // -- emit decisions (every run of the body has count >= 32 >= LONG: stored; the leading run may stay literals)
      auto decide = [&](uint32_t iS, bool &sm) __attribute__((always_inline)) -> int {
        const uint32_t rng = p - (iS & 0x1FFFu) + 1u;
        sm = PK && same_as(sym, iS);
        const bool shortOk = rng <= MAXR && (PK ? (count >= SHORT && (sm || count >= MEDIUM)) : count >= SHORT);
        return shortOk ? 1 : (count >= LONG ? 2 : 0);
      };
      const bool sure = isRun && count >= LONG;
      uint32_t outS = e | (p << 13), inS = 0;
      bool outKnown = sure || !have, inKnown = !have;
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), ls = wave_shr1(outS, carS);
        if (!inKnown && lk != 0u) { inKnown = true; inS = ls; }
        if (inKnown && !outKnown)
        {
          bool sm;
          if (!isRun || decide(inS, sm) == 0) outS = inS;
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      bool same = false;
      const int k = isRun ? decide(inS, same) : 0;
      const uint32_t inL = inS & 0x1FFFu;
      const uint32_t gap = p - inL;
      const uint32_t cfield = cfield_of(count);
      const uint32_t hl = (cfield <= cMax ? 1u : 5u) + ((PK && same) ? 0u : SU) + ((k == 1) ? 1u : (R7 ? 4u : 5u));
      const uint32_t myBytes = k ? hl + gap : 0u;
      const uint32_t incl = wave_scan_add(myBytes | (k ? 0x10000u : 0u));
      const uint32_t tot = wave_lane(incl, 63);
      const uint32_t idx = K + (incl >> 16) - 1u;
      if (k && idx < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + idx] = p | ((e - 1u) << 12) | (same ? 1u << 24 : 0u) | (k == 2 ? 1u << 25 : 0u);
      carS = wave_lane(outS, lastLane);
      pos += tot & 0xFFFFu;
      K += tot >> 16;
      nextCand += fresh;
      keep = 0u;
      leadRound = false;
    }

    // ---- 3. behind the last run: the pair search's hops to the first position >= n - 32 (:233-268), then the byte steps (:270-300) ----
    uint32_t lastRLE = carS & 0x1FFFu;
    bool ended = lastRLE >= n && K != 0u;
    {
      uint32_t i = carE;
      while ((int32_t)i < T)
      {
        const uint32_t win = ebits16(i);
        if (win == 0xFFFFu) break;                                          // (cannot happen: every such window in front of n - 32 became a run)
        i += 32u - (uint32_t)__builtin_clz((~win) & 0xFFFFu);               // behind the highest clear bit
      }
      auto put_tail_run = [&](uint32_t p, uint32_t e, bool same, int k) __attribute__((always_inline)) {
        const uint32_t count = e - p, cf = cfield_of(count);
        pos += (cf <= cMax ? 1u : 5u) + ((PK && same) ? 0u : SU) + ((k == 1) ? 1u : (R7 ? 4u : 5u)) + (p - lastRLE);
        if (lane == 0u && K < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + K] = p | ((e - 1u) << 12) | (same ? 1u << 24 : 0u) | (k == 2 ? 1u << 25 : 0u);
        K++;
        lastRLE = e;
      };
      if (T >= 0 && (int32_t)i == T && ebits16((uint32_t)T) == 0xFFFFu)
      {
        // a pair at n - 32 exactly: a last run of 32 bytes (>= LONG: stored)
        const u32x4 sy = pp_sym16(sh.inb, kPpInPad + (uint32_t)T);
        const bool sm = PK && same_as(sy, carS);
        const uint32_t rng = (uint32_t)T - lastRLE + 1u;
        const bool shortOk = rng <= MAXR;                                   // (count = 32 >= SHORT, MEDIUM)
        put_tail_run((uint32_t)T, n, sm, shortOk ? 1 : 2);
        ended = true;
      }
      else if constexpr (PK)
      {
        // byte steps: `runs` of bytes that equal their left neighbour, stored only as the last stored symbol again
        const uint32_t tb = T > 0 ? (uint32_t)T : 0u;
        const uint32_t tp = tb + lane;
        const uint64_t e1 = __ballot(lane < 32u && tp + 1u < n && bytes[tp] == bytes[tp + 1u]);   // bit k: d[tb + k] == d[tb + k + 1]
        const uint64_t three = e1 & (e1 >> 1) & (e1 >> 2);
        if (i >= tb && three != 0ull)
        {
          uint32_t j = i;
          while ((int32_t)j + 1 < (int32_t)n - 16)
          {
            const uint32_t c1 = (uint32_t)__builtin_ctzll(~(e1 >> (j - tb)));
            const uint32_t count = c1 >= 16u ? 16u : (AL ? 0u : c1);
            const uint32_t i2 = j + 1u + count;
            if (count >= SHORT)
            {
              const uint32_t rng = j + 2u - lastRLE;
              const bool eq = __ballot(lane < 16u && bytes[j + lane] != (((carS & (NOSYM << 13)) != 0u) ? 0u : (uint32_t)bytes[((carS >> 13) & 0x1FFFu) + lane])) == 0ull;
              if (rng <= MAXR && eq) put_tail_run(i2 - count, i2, true, 1);
            }
            j = i2;
          }
        }
      }
    }

    // ---- 4. stream size ----
    const uint32_t kLit = ended ? 0u : n - lastRLE;
    const uint32_t streamSize = pos + (ended ? TERM_END : TERM) + kLit;
    if (lane == 0u) { sizes[b] = streamSize; sc.recCount[b] = K; }
  }
  else
  {
    // ---- MODE 1: the stream from the block's records ----
    const uint32_t *const myRecs = sc.recs + (uint64_t)b * sc.recStride;
    const uint32_t R = sc.recCount[b];
    const uint64_t myOffset = offsets[b];                                 // (asked for at the start, needed at the very end)
    wave_sync();
    uint32_t carL = 0, pos = HDR;
    bool ended = false;
    for (uint32_t r0 = 0; r0 < R; r0 += 64u)
    {
      const bool have = r0 + lane < R;
      const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
      const uint32_t rec = (r0 == 0u) ? rec0 : (have ? myRecs[r0 + lane] : 0u);
      const uint32_t p = rec & 0xFFFu, e = ((rec >> 12) & 0xFFFu) + 1u;
      const bool same = ((rec >> 24) & 1u) != 0u;
      const int k = have ? 1 + (int)((rec >> 25) & 1u) : 0;
      const uint32_t inL = wave_shr1(e, carL);
      const uint32_t count = e - p, gap = p - inL, rng = gap + 1u;
      const uint32_t cfield = cfield_of(count);
      const uint32_t cBytes = cfield <= cMax ? 1u : 5u, sBytes = (PK && same) ? 0u : SU, rBytes = (k == 1) ? 1u : (R7 ? 4u : 5u);
      const uint32_t hl = cBytes + sBytes + rBytes;
      const uint32_t myBytes = k ? hl + gap : 0u;
      const uint32_t incl = wave_scan_add(myBytes);
      const uint32_t tot = wave_lane(incl, 63);
      const uint32_t at0 = pos + incl - myBytes;
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        // [count | same] [symbol] [range] (Packed) / [symbol] [count] [range] (plain)
        uint32_t a = at0;
        const uint64_t cval = cfield <= cMax ? (uint64_t)(cfield | ((PK && same) ? 0x80u : 0u)) : (((uint64_t)cfield << 8) | ((PK && same) ? 0x80u : 0u));
        const uint64_t rval = (k == 1) ? (uint64_t)(R7 ? (rng << 1) & 0xFFu : rng) : (R7 ? (uint64_t)((rng << 1) | 1u) : ((uint64_t)rng << 8));
        if constexpr (PK) { pp_or_bytes(sh.img, a, cval, cBytes); a += cBytes; }
        if (sBytes)
        {
          pp_or_bytes(sh.img, a, pp_symbol<8>(sh.inb, kPpInPad + p), 8u);
          pp_or_bytes(sh.img, a + 8u, pp_symbol<8>(sh.inb, kPpInPad + p + 8u), 8u);
          a += SU;
        }
        if constexpr (!PK) { pp_or_bytes(sh.img, a, cval, cBytes); a += cBytes; }
        pp_or_bytes(sh.img, a, rval, rBytes);
        ds = at0 + hl;
        if (gap > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)inL | ((uint64_t)ds << 13) | ((uint64_t)gap << 26); }
        else if (gap != 0u) nch = ((ds + gap - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)
      {
        if (t < nch) pp_put_chunks(sh, inL, ds, gap, t, 1u, t + 1u);
        if (t + 1u < nch) pp_put_chunks(sh, inL, ds, gap, t + 1u, 1u, t + 2u);
      }
      carL = wave_lane(e, lastLane);
      pos += tot;
      if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
    }
    const uint32_t kLit = ended ? 0u : n - carL;
    const uint32_t streamSize = pos + (ended ? TERM_END : TERM) + kLit;
    if (lane < 8u)
    {
      const uint64_t h = (uint64_t)n | ((uint64_t)streamSize << 32);
      sh.img[lane] = (uint8_t)(h >> (8u * lane));
    }
    if (lane == 16u)
    {
      // plain: 16 zero bytes, 00, u32 0 | Packed: 80, u32 0;  then: end -- 00, u32 0 (always the plain form);  literals -- (u32 (literals + 1) << 1 | 1) where the
      // codec has 7-bit ranges, else 00, u32 literals + 1
      const uint32_t a0 = pos + (PK ? 0u : SU);
      if constexpr (PK) sh.img[a0] = 0x80;
      if (!ended)
      {
        if constexpr (R7) pp_or_bytes(sh.img, a0 + 5u, (uint64_t)(((kLit + 1u) << 1) | 1u), 4u);
        else pp_or_bytes(sh.img, a0 + 5u, (uint64_t)(kLit + 1u) << 8, 5u);
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

template <bool PK, int AL, int MODE>
__global__ __launch_bounds__(64) void k_encode128_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                     const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpScratch sc)
{
  __shared__ Pp128Shared<MODE != 0> sh;
  if constexpr (MODE != 0)
  {
    if (threadIdx.x < 17u)
    {
      const uint32_t c = threadIdx.x;
      const uint64_t part = ~(~0ull << (8u * (c & 7u)));
      const bool hiHalf = c >= 8u;
      const uint32_t p0 = (c == 16u) ? ~0u : (uint32_t)part, p1 = (c == 16u) ? ~0u : (uint32_t)(part >> 32);
      lds_st128(sh.mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
    }
    wave_sync();
  }
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
  if (b < nBlocks)
  {
    u32x4 x[4];
    pp_load(in, U, B, b, x);
    uint32_t rec0 = 0;
    if constexpr (MODE == 1) rec0 = sc.recs[(uint64_t)b * sc.recStride + threadIdx.x];
    pp128_block<PK, AL, MODE>(U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
  }
}

} // namespace hsrle
