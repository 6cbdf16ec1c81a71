// hsrle_encode8sp.hip.h -- POSITION-PARALLEL encoder for the two 8 bit Single codecs (rle8_single, rle8_packed_single), blocks of at most 4 KiB:
// one wave per block, symbol pick + run discovery + emit decisions in ONE pass over the block, the payload written once (round 6).
//
// Replaces: src/rle8_extreme_cpu.c:53-153 (symbol pick), src/rle8_extreme_cpu.h:346-700 (wrapper, scalar tail, final block), :1103-1321 (the SSE2 body
//           with its `wastedChances` back-tracking) -- and, in this library, k_single_pick + the lane-per-block k_encode8_single_blocks + staging slots +
//           k_compact for containers of these codecs (hsrle_encode8s.hip.h: 2.4 + 7.3 + 3.0 ms per 8 GiB, the input read twice, the payload written twice).
//
// The reference's scanner walks 16-byte windows at a data dependent phase.  What it decides does not depend on the phase, except in one place
// (closed forms checked against the oracle on the CPU: tools/single_pp_model.py):
//   * every maximal run of the symbol of >= SHORT bytes is found with its true start (a window that holds its first byte is never skipped: either the
//     whole run lies inside -- SHORT matches -- or it reaches the window's last byte, :1295);
//   * a run (p, L) is judged by the BODY iff the trip that ends it starts in front of n - 16: p == 0 ? 16 (L / 16) : p + 1 + 16 ((L - 1) / 16) -- else by
//     the scalar tail (:383-510) or the final block (:512-694), which have no wasted-chances logic and no MEDIUM rule;
//   * body: range <= 255 -> short form; L >= LONG, Packed: L >= MEDIUM -> long form whatever the range ("sure": the state behind it is known at once);
//     else a wasted chance -- and the third one within 255 bytes of the first goes back and stores the first in the long form, which makes the other two
//     short-form runs (:1244-1285).  As a chain: the state is (lastRLE, wasted, firstWasted), a lane's candidate maps it forward, the lane that sees the
//     third chance tells its two left neighbours;
//   * where the body's search runs off n - 16 (:1291-1312 leaves i at the first window start >= n - 16, the for loop adds one) one byte is never looked
//     at: a run that starts exactly there loses its first byte.  THAT depends on the phase, so -- only when a run of the symbol starts in the last 32
//     bytes -- the windows are replayed (wave-uniform, on the match bits in LDS) from the end of the last run whose trip is known to lie in the body.
// The pick is k_single_pick's closed form (hsrle_encode8s.hip.h: a maximal run of L >= 2 equal bytes that ends in front of n - 16 adds L - (L - 1) / 16
// to prob[s] and 1 to pcount[s]; tools/pick_model.py) on the same registers: equality bits of neighbouring bytes as in hsrle_encode8p.hip.h, every lane
// the runs that END in its 64 positions.
// Two launches around the size scan, as hsrle_encode8p.hip.h: MODE 0 leaves sizes[b], the symbol and one record per stored run; MODE 1 places headers and
// literals from the records into an LDS image of the stream and writes it once.
#pragma once

#include "hsrle_encodeSp.hip.h"   // pp_or_bytes; hsrle_encode8p.hip.h: the DPP primitives, PpScratch, pp_load

namespace hsrle {

constexpr uint32_t kPpsNoRecords = 0xFFFFu;                     // low half of recCount[b] (the symbol sits in bits 16 .. 23)

template <bool EMIT>
struct PpSingleShared
{
  uint8_t img[EMIT ? (kPpMaxBlock + 193u + 15u + 16u + 15u) / 16u * 16u : 16u] __attribute__((aligned(16)));   // the stream under construction
  uint8_t inb[kPpInPad + kPpMaxBlock + 32u] __attribute__((aligned(16)));                                       // the block's input
  uint8_t mlut[EMIT ? 17u * 16u : 16u] __attribute__((aligned(16)));   // entry c: the low c bytes, c = 0 .. 16
  uint64_t starts[64];                                                  // run-start bits (runs of the symbol) of every lane's 64 positions
  uint64_t mbits[66];                                                   // match bits d[j] == symbol, two zero words behind them
  uint32_t table[EMIT ? 1u : 256u];                                     // the pick: prob | pcount << 16 per symbol
  uint64_t jobs[EMIT ? kPpJobs : 1u];                                   // literal stretches for the whole wave
  uint16_t lst[64];                                                     // the round's candidates (last byte positions)
  uint16_t carryStart[64];                                              // start of the run that is open where a lane's positions begin
  uint32_t jobCount;
};

// literal bytes [src, src + len) of the block -> image bytes [ds, ds + len): the destination chunks t0, t0 + tStep, ... below tEnd (hsrle_encode8p.hip.h: put_chunks)
template <class SH>
__device__ __forceinline__ void pp_put_chunks(SH &sh, uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd)
{
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
}

// the block's symbol (rle8_extreme_cpu.c:53-153 as k_single_pick's closed form): x = this lane's 64 bytes, the input image is in sh.inb
template <bool EMIT>
__device__ __forceinline__ uint32_t pps_pick(PpSingleShared<EMIT> &sh, const u32x4 (&x)[4], uint32_t n)
{
  const uint32_t lane = threadIdx.x, base = lane * 64u;
  const uint8_t *const bytes = sh.inb + kPpInPad;
#pragma unroll
  for (int k = 0; k < 4; k++) sh.table[lane * 4u + k] = 0u;
  // equality bits of neighbouring bytes: bit i = (d[base + i] == d[base + i + 1]) and base + i + 1 < n
  const uint32_t nextFirst = wave_shl1(x[0].x, 0u);
  uint64_t e64 = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++)
  {
    const u32x4 a = x[j];
    const uint32_t s = (j < 3u) ? x[j < 3u ? j + 1u : 3u].x : nextFirst;
    e64 |= (uint64_t)zero_mask16(a.x ^ alignbyte(a.y, a.x, 1), a.y ^ alignbyte(a.z, a.y, 1), a.z ^ alignbyte(a.w, a.z, 1), a.w ^ alignbyte(s, a.w, 1)) << (16u * j);
  }
  const uint32_t validBits = (n > base + 1u) ? ((n - 1u - base) < 64u ? (n - 1u - base) : 64u) : 0u;
  e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
  const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(e64 >> 63), 0u);
  const uint64_t prev = (e64 << 1) | carry;
  const uint64_t starts = e64 & ~prev;
  uint64_t ends = ~e64 & prev;                                             // bit i: a run's last byte is base + i
  const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
  const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
  wave_sync();                                                             // (the zeroed table, the input image)

  // the runs that end in my 64 positions.  Safe runs (they end in front of n - 16) go to the table; of the others the first one is kept for lane 0
  const int32_t end = (int32_t)n - 16;
  int32_t lastSafeEnd = 0;                                                 // max over the safe runs of j + L
  int32_t lateKey = -1;                                                    // max over the late runs of 0x7FFFFFFF - ((j << 16) | (L - 1))
  while (ends != 0ull)
  {
    const uint32_t i = (uint32_t)__builtin_ctzll(ends);
    ends &= ends - 1ull;
    const uint32_t q = base + i;
    const uint64_t sBelow = starts & ((1ull << i) - 1ull);
    const uint32_t p = (sBelow != 0ull) ? base + 63u - (uint32_t)__builtin_clzll(sBelow) : carryStart;
    const uint32_t L = q - p + 1u;
    if ((int32_t)(q + 1u) < end)
    {
      atomicAdd(&sh.table[bytes[q]], (1u << 16) | (L - (L - 1u) / 16u));
      lastSafeEnd = imax(lastSafeEnd, (int32_t)(q + 1u));
    }
    else
      lateKey = imax(lateKey, 0x7FFFFFFF - (int32_t)((p << 16) | (L - 1u)));
  }
  lastSafeEnd = (int32_t)wave_lane((uint32_t)wave_scan_max(lastSafeEnd), 63);
  lateKey = (int32_t)wave_lane((uint32_t)wave_scan_max(lateKey), 63);
  wave_sync();

  // the scanner's first window, the run that reaches n - 16 and the final registration (rle8_extreme_cpu.c:66-139)
  if (lane == 0u)
  {
    const uint32_t d0 = bytes[0];
    const uint32_t inv = (~d0) & 0xFFu;
    uint32_t finSym, finCount;
    if (end <= 0) { finSym = inv; finCount = 0u; }
    else
    {
      bool any = false;
      for (uint32_t k = 0; k < 16u; k++) any = any || bytes[k] == inv;
      if (any) sh.table[inv] += 1u << 16;                                  // registered with count 0
      const uint32_t i0 = (uint32_t)lastSafeEnd;                           // the search behind the last safe run starts here (< end)
      bool have = false;
      finSym = 0; finCount = 1u;
      if (lateKey >= 0)
      {
        const uint32_t firstLate = (uint32_t)(0x7FFFFFFF - lateKey);
        const uint32_t j = firstLate >> 16, L = (firstLate & 0xFFFFu) + 1u;
        const uint32_t q = i0 + 15u * ((j - i0) / 15u);                    // the search trip that would find it
        if ((int32_t)q < end)
        {
          have = true;
          uint32_t i = j + 1u, count = 1u;
          bool registered = false;
          while ((int32_t)i < end)
          {
            const uint32_t rem = L - (i - j);
            if (rem >= 16u) { count += 15u; i += 16u; }
            else
            {
              count += rem; i += rem;
              sh.table[bytes[j]] += (1u << 16) | count;
              registered = true;
              break;
            }
          }
          if (registered) { finSym = bytes[i]; finCount = 1u; }
          else { finSym = bytes[j]; finCount = count; }
        }
      }
      if (!have)
      {
        const uint32_t i = ((int32_t)i0 < end) ? i0 + 15u * (((uint32_t)end - i0 + 14u) / 15u) : i0;   // the search runs off the end
        finSym = bytes[i]; finCount = 1u;
      }
    }
    sh.table[finSym] += (1u << 16) | finCount;
  }
  wave_sync();

  // argmax of prob - 2 pcount over the symbols with pcount > 0 and prob / pcount > 2; the first maximum wins
  const bool zeroStartsFull = bytes[0] != 0u;                              // pcount[0] starts as 0xFFFFFFFF unless d[0] == 0 (:61-62)
  int32_t bestKey = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t s = lane * 4u + (uint32_t)k;
    const uint32_t v = sh.table[s];
    const uint32_t prob = v & 0xFFFFu;
    uint32_t pc = v >> 16;
    if (s == 0u && zeroStartsFull) pc -= 1u;                               // modulo 2^32, as the reference's counter
    if (pc > 0u && prob / pc > 2u)
    {
      const uint32_t saved = prob - pc * 2u;
      bestKey = imax(bestKey, (int32_t)((saved << 8) | (255u - s)));
    }
  }
  bestKey = (int32_t)wave_lane((uint32_t)wave_scan_max(bestKey), 63);
  return ((uint32_t)bestKey >> 8) != 0u ? (255u - ((uint32_t)bestKey & 0xFFu)) : 0u;
}

// one block by one wave.  MODE 0: sizes[b], the symbol and the block's records; MODE 1: the stream, written to payload + offsets[b]
// CODEC 0: rle8_single, 1: rle8_packed_single, 2: rle8_single_short (the Short family's Single codec, src/rleX_Xsl_short.h with SINGLE: wrapper :380-523, body
// :1058-1120 -- the same estimator and the same window scanner; every run end goes through process_symbol, :152-372: stored iff count >= 11 or count >= 2 + the
// bytes its packet needs beyond the one-byte form; one rule for body and tail, no wasted chances, no symbol in the packets; tools/single_pp_model.py: model_short)
template <int CODEC, int MODE>
__device__ __forceinline__ void pps_block(uint64_t U, uint32_t B, uint32_t b, uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload,
                                          const PpScratch &sc, PpSingleShared<MODE != 0> &sh, const u32x4 (&x)[4], uint32_t rec0)
{
  constexpr bool PK = CODEC == 1, SS = CODEC == 2;
  constexpr uint32_t SHORT = (PK || SS) ? 2u : 4u, LONG = SS ? 11u : (PK ? 10u : 8u);
  constexpr uint32_t SURE = SS ? 11u : (PK ? 6u : 8u);   // body: stored whatever the range (Packed: the MEDIUM rule, :1195; Short: SMINL)
  constexpr uint32_t TERM = SS ? 9u : 10u;           // 00 | u32 0 | 00 | u32 (0 or literals + 1);  Short: F0 08 00 | u16 0 | u32 literals + 2
  constexpr uint32_t TERM_END = SS ? 7u : 10u;       // Short: F0 08 01 | u16 0 | u16 0
  constexpr uint32_t HDR = SS ? 9u : 10u;            // u32 uncompressed, u32 compressed, mode = 1, the symbol;  Short: no mode byte
  // Short header parameters without a list (hsrle_common.hip.h: Traits<SHORT_SINGLE, 1, 0>): one byte [count - 2 : 4 | gap : 4], or three: [F | 9 bit count | 11 bit gap + 2]
  // with 16 bit fields behind them where count > 511 or gap + 2 > 2047
  constexpr uint32_t S_MAXPR = 15u, S_MAXPC = 14u, S_MAXTR = 2047u, S_MAXTC = 511u;
  const uint32_t lane = threadIdx.x;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint32_t base = lane * 64u;
  const int32_t end = (int32_t)n - 16;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const uint32_t *const myRecs = sc.recs + (uint64_t)b * sc.recStride;
  // (the stream's place: asked for at the start, needed at the very end -- not a dependent load in front of the copy-out)
  [[maybe_unused]] uint64_t myOffset = 0;
  if constexpr (MODE == 1) myOffset = offsets[b];
  // (and the block's second 64 records, where its record area holds that many: a block of more than 64 stored runs -- the rule on video-shaped data -- does not wait for them in its second round)
  [[maybe_unused]] uint32_t rec1 = 0;                                    // (the third and fourth 64 too: measured no better, four loads for every block)
  if constexpr (MODE == 1) { if (sc.recStride >= 128u) rec1 = myRecs[64u + threadIdx.x]; }

  uint32_t recWord = kPpsNoRecords;
  if constexpr (MODE == 1) recWord = sc.recCount[b];
  const uint32_t recN = recWord & 0xFFFFu;
  const bool fromRecs = MODE == 1 && recN != kPpsNoRecords;

  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);

  uint32_t sym;
  if constexpr (MODE == 0) sym = pps_pick(sh, x, n);
  else sym = (recWord >> 16) & 0xFFu;

  // ---- 1. match bits against the symbol, runs, candidates, the skipped byte ----
  uint32_t R = recN;
  uint64_t candLeft = 0;
  uint32_t candAt = 0;
  uint32_t quirk = 0xFFFFFFFFu;
  if (!fromRecs)
  {
    const uint32_t pv = sym * 0x01010101u;
    uint64_t m64 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
      m64 |= (uint64_t)zero_mask16(x[j].x ^ pv, x[j].y ^ pv, x[j].z ^ pv, x[j].w ^ pv) << (16u * j);
    const uint32_t vb = n > base ? (n - base < 64u ? n - base : 64u) : 0u;
    m64 &= (vb >= 64u) ? ~0ull : ((1ull << vb) - 1ull);
    const uint32_t prevTop = wave_shr1((uint32_t)(m64 >> 32), 0u);          // the top dword of the lane in front
    const uint64_t nextBit0 = (uint64_t)wave_shl1((uint32_t)m64 & 1u, 0u);
    const uint64_t starts = m64 & ~((m64 << 1) | (uint64_t)(prevTop >> 31));
    const uint64_t lasts = m64 & ~((m64 >> 1) | (nextBit0 << 63));          // bit i: a run's last byte is base + i
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
    // candidates: runs of at least SHORT bytes (no start at the last byte or in the SHORT - 2 positions in front of it)
    uint64_t near = starts;
    if constexpr (SHORT == 4u)
    {
      const uint32_t sTop = wave_shr1((uint32_t)(starts >> 32), 0u);
      near |= ((starts << 1) | (uint64_t)(sTop >> 31)) | ((starts << 2) | (uint64_t)(sTop >> 30));
    }
    const uint64_t cands = lasts & ~near;
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    sh.mbits[lane] = m64;
    if (lane < 2u) sh.mbits[64u + lane] = 0ull;
    candLeft = cands;
    candAt = inclCnt - cnt;
    wave_sync();

    // the byte the body's search never looks at (see the header): only where a run of the symbol starts in [n - 16, n)
    if (end > 0)
    {
      const int32_t lo = (end > 1 ? end : 1) - (int32_t)base, hi = end + 15 - (int32_t)base;      // my positions [lo, hi]
      uint64_t rm = 0ull;
      if (hi >= 0 && lo <= 63)
      {
        const uint32_t l0 = lo > 0 ? (uint32_t)lo : 0u, h0 = hi < 63 ? (uint32_t)hi : 63u;
        rm = ((h0 >= 63u) ? ~0ull : ((2ull << h0) - 1ull)) & ~((1ull << l0) - 1ull);
      }
      if (__ballot((starts & rm) != 0ull) != 0ull)
      {
        // where the phase is known: behind the last run of >= SHORT bytes that ends in front of n - 16 (its trip lies in the body), or behind the block's leading run
        const int32_t lim = end - 1 - (int32_t)base;                        // candidates at my positions < lim
        const uint64_t cm = lim <= 0 ? 0ull : (lim >= 64 ? cands : cands & ((1ull << lim) - 1ull));
        int32_t a0 = (cm != 0ull) ? (int32_t)(base + 64u - (uint32_t)__builtin_clzll(cm)) : 0;
        if (lane == 0u && (m64 & 1ull) != 0ull) a0 = imax(a0, (m64 == ~0ull) ? 64 : (int32_t)__builtin_ctzll(~m64));
        uint32_t w = wave_lane((uint32_t)wave_scan_max(a0), 63);
        bool hit = false;
        while ((int32_t)w < end)
        {
          const uint32_t wd = w >> 6, ws = w & 63u;
          const uint64_t b0 = sh.mbits[wd], b1 = sh.mbits[wd + 1u];
          const uint64_t bits = ws ? (b0 >> ws) | (b1 << (64u - ws)) : b0;   // match bits from position w on
          const uint32_t win = (uint32_t)bits & 0xFFFFu;
          if (win == 0u || ((win & 0x8000u) == 0u && (uint32_t)__builtin_popcount(win) < SHORT)) { w += 16u; continue; }   // (:1295)
          const uint32_t f = (uint32_t)__builtin_ctz(win);
          const uint32_t Lf = (uint32_t)__builtin_ctzll(~(bits >> f) | (1ull << 40));
          if (Lf >= SHORT) { hit = true; break; }                            // a run of >= SHORT bytes: its trip decides, nothing is skipped
          w += f + Lf;
        }
        if (!hit && ((sh.mbits[w >> 6] >> (w & 63u)) & 1ull) != 0ull) quirk = w;
      }
    }
  }
  else
    wave_sync();

  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t carL = 0;                                 // lastRLE in front of the round's first candidate
  uint32_t pos = HDR;
  uint32_t K = 0;
  bool ended = false;
  uint32_t nextCand = 0, keep = 0;                   // candidates from nextCand on are not listed yet; `keep` entries at the head of the list stay for this round
  for (;;)
  {
    const uint32_t fresh = (R - nextCand) < (64u - keep) ? (R - nextCand) : (64u - keep);
    const uint32_t inRound = keep + fresh;
    if (inRound == 0u) break;
    const bool have = lane < inRound;
    const int lastLane = (int)inRound - 1;
    uint32_t p = 0, e = 0, inL = 0, outL = 0;
    int k = 0;
    uint32_t keepNext = 0;
    if (fromRecs)
    {
      const uint32_t rec = (nextCand == 0u) ? rec0 : ((nextCand == 64u && sc.recStride >= 128u) ? rec1 : (have ? myRecs[nextCand + lane] : 0u));
      p = rec & 0xFFFu; e = ((rec >> 12) & 0xFFFu) + 1u;
      k = have ? 1 + (int)((rec >> 25) & 1u) : 0;
      outL = e;
      inL = wave_shr1(outL, carL);
    }
    else
    {
      while (candLeft != 0ull && candAt < nextCand + fresh)
      {
        sh.lst[keep + candAt - nextCand] = (uint16_t)(base + (uint32_t)__builtin_ctzll(candLeft));
        candAt++;
        candLeft &= candLeft - 1ull;
      }
      wave_sync();
      const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;
      wave_sync();
      const uint32_t iq = q >> 6, bit = q & 63u;
      const uint64_t st = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      const uint32_t pT = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;
      e = q + 1u;
      const uint32_t LT = e - pT;
      const uint32_t ie = (pT == 0u) ? 16u * (LT / 16u) : pT + 1u + 16u * ((LT - 1u) / 16u);      // where the trip that ends the run starts
      const bool vec = have && end > 0 && (int32_t)ie < end;
      p = (pT == quirk) ? pT + 1u : pT;
      const uint32_t L = e - p;
      const bool cand = have && L >= SHORT;                                // (only a run that lost its first byte can fall below)

      // state: lastRLE | firstWasted << 13 | wasted << 26.  Returns the state behind the candidate; kk: 0 not stored, 1 short form, 2 long form, 3 short form
      // as the third wasted chance (the two candidates in front are stored after all)
      auto step = [&](uint32_t in, int &kk) __attribute__((always_inline)) -> uint32_t {
        kk = 0;
        if (!cand) return in;
        const uint32_t iL = in & 0x1FFFu;
        if constexpr (SS)
        {
          const uint32_t gp = p - iL;
          const bool one = gp <= S_MAXPR && L - 2u <= S_MAXPC;
          const uint32_t pen = one ? 0u : 2u + (gp + 2u <= S_MAXTR ? 0u : 2u) + (L <= S_MAXTC ? 0u : 2u);
          if (L >= 11u || L >= 2u + pen) { kk = one ? 1 : 2; return e; }
          return in;
        }
        const uint32_t rng = p - iL + 1u;
        if (rng <= 255u) { kk = 1; return e; }
        if (L >= (vec ? SURE : LONG)) { kk = 2; return e; }
        if (!vec) return in;
        const uint32_t fw = (in >> 13) & 0x1FFFu, w = (in >> 26) + 1u;
        if (w == 1u || e - fw > 255u) return iL | (p << 13) | (1u << 26);
        if (w == 2u) return iL | (fw << 13) | (2u << 26);
        kk = 3;
        return e;
      };
      const bool sure = cand && L >= (vec ? SURE : LONG);
      uint32_t outS = e, inS = 0;
      bool outKnown = sure || !have, inKnown = !have;
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), ls = wave_shr1(outS, carL);
        if (!inKnown && lk != 0u) { inKnown = true; inS = ls; }
        if (inKnown && !outKnown)
        {
          int kk;
          outS = step(inS, kk);
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      int kk;
      outS = step(inS, kk);
      const uint32_t rb0 = (kk == 3) ? 1u : 0u, rb1 = wave_shl1(rb0, 0u), rb2 = wave_shl1(rb1, 0u);
      const uint32_t ePrev = wave_shr1(e, 0u);
      inL = (rb0 | rb1) ? ePrev : (inS & 0x1FFFu);
      k = !have ? 0 : ((rb0 | rb1) ? 1 : (rb2 ? 2 : (kk == 3 ? 1 : kk)));
      outL = outS & 0x1FFFu;
      // wasted chances that are still open behind the round's last candidate: those candidates are judged again at the head of the next round, with what
      // follows them (they are the round's last one or two candidates, the state in front of them holds no wasted chance)
      const uint32_t stLast = wave_lane(outS, lastLane);
      const bool lastVec = ((__ballot(vec) >> lastLane) & 1ull) != 0ull;
      if (nextCand + fresh < R && lastVec) keepNext = stLast >> 26;
      if (keepNext != 0u)
      {
        const uint32_t t = (lane < keepNext) ? (uint32_t)sh.lst[64u - keepNext + lane] : 0u;
        wave_sync();
        if (lane < keepNext) sh.lst[lane] = (uint16_t)t;
      }
    }
    const uint32_t L = e - p, gap = p - inL, rng = gap + 1u;
    const uint32_t cfield = L - SHORT + 1u;
    const uint32_t cb = cfield <= 255u ? 1u : 5u, rbytes = (k == 1) ? 1u : 5u;
    const uint32_t hl = SS ? ((k == 1) ? 1u : 3u + (L > S_MAXTC ? 2u : 0u) + (rng + 1u > S_MAXTR ? 2u : 0u)) : cb + rbytes;
    const uint32_t myBytes = k ? hl + gap : 0u;
    const uint32_t incl = wave_scan_add(myBytes | (k ? 0x10000u : 0u));    // bytes below bit 16, stored runs above
    const uint32_t tot = wave_lane(incl, 63);
    if constexpr (MODE == 0)
    {
      const uint32_t idx = K + (incl >> 16) - 1u;
      if (k && idx < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + idx] = p | ((e - 1u) << 12) | (k == 2 ? 1u << 25 : 0u);
    }
    else
    {
      const uint32_t at0 = pos + (incl & 0xFFFFu) - myBytes;               // the packet's place in the stream
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        // header (rle8_extreme_cpu.h:1148-1182): [count - SHORT + 1, or 00 and a u32] [range, or 00 and a u32]
        if constexpr (SS)
        {
          // (rleX_Xsl_short.h:216-357 without a list index)
          const uint32_t srange = gap + 2u;
          if (k == 1) pp_or_bytes(sh.img, at0, (uint64_t)(((L - 2u) << 4) | gap), 1u);
          else
          {
            const uint32_t scx = L <= S_MAXTC ? L : 1u, rx = srange <= S_MAXTR ? srange : 1u;       // (1: a 16 bit field follows)
            const uint32_t f = scx << 3;
            pp_or_bytes(sh.img, at0, (uint64_t)(((0xF0u | (f >> 8)) & 0xFFu) | (((f | (rx >> 8)) & 0xFFu) << 8) | ((rx & 0xFFu) << 16)), 3u);
            uint32_t a = at0 + 3u;
            if (scx != L) { pp_or_bytes(sh.img, a, (uint64_t)L, 2u); a += 2u; }
            if (rx != srange) pp_or_bytes(sh.img, a, (uint64_t)srange, 2u);
          }
        }
        else if (cb == 1u && k == 1) pp_or_bytes(sh.img, at0, (uint64_t)(cfield | (rng << 8)), 2u);
        else
        {
          pp_or_bytes(sh.img, at0, cb == 1u ? (uint64_t)cfield : (uint64_t)cfield << 8, cb);
          pp_or_bytes(sh.img, at0 + cb, (k == 1) ? (uint64_t)rng : (uint64_t)rng << 8, rbytes);
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
    K += tot >> 16;
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
    nextCand += fresh;
    keep = keepNext;
  }

  // ---- 3. terminator, stream size (:512-694) ----
  const uint32_t kLit = ended ? 0u : n - carL;
  const uint32_t streamSize = pos + (ended ? TERM_END : TERM) + kLit;
  if constexpr (MODE == 0)
  {
    if (lane == 0u) { sizes[b] = streamSize; sc.recCount[b] = ((K <= sc.recStride) ? K : kPpsNoRecords) | (sym << 16); }
    return;
  }
  else
  {
    if (lane < HDR)
    {
      const uint64_t h = (uint64_t)n | ((uint64_t)streamSize << 32);
      sh.img[lane] = lane < 8u ? (uint8_t)(h >> (8u * lane)) : ((lane == 8u && !SS) ? (uint8_t)1 : (uint8_t)sym);
    }
    if constexpr (SS)
    {
      // end: F0 08 01, u16 0, u16 0;  literals: F0 08 00, u16 0, u32 literals + 2  (rleX_Xsl_short.h:976-1032 with SINGLE)
      if (lane == 16u)
      {
        pp_or_bytes(sh.img, pos, (uint64_t)(0xF0u | (8u << 8) | (ended ? 1u << 16 : 0u)), 3u);
        if (!ended) pp_or_bytes(sh.img, pos + 5u, (uint64_t)(kLit + 2u), 4u);
      }
    }
    else if (lane == 16u && !ended) pp_or_bytes(sh.img, pos + 6u, (uint64_t)(kLit + 1u), 4u);
    wave_sync();
    // the long stretches and the literals behind the last stored run: every lane a chunk
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
    // the image leaves LDS once
    {
      uint8_t *const dst = payload + myOffset;
      const uint32_t nFull = streamSize >> 4, tail = streamSize & 15u;
      for (uint32_t c = lane; c < nFull; c += 64u)
        st128(dst + 16u * c, lds_ld128(sh.img + 16u * c));
      if (lane < tail) dst[16u * nFull + lane] = sh.img[16u * nFull + lane];
    }
  }
}

template <int CODEC, int MODE>
__global__ __launch_bounds__(64) void k_encode8s_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                    const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpScratch sc)
{
  __shared__ PpSingleShared<MODE != 0> sh;
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
    pps_block<CODEC, MODE>(U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
  }
}

} // namespace hsrle
