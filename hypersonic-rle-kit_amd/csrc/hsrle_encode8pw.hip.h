// hsrle_encode8pw.hip.h -- the position-parallel 8 bit encoder (hsrle_encode8p.hip.h: rle8_multi, rle8_packed_multi) for UNITS OF ANY LENGTH: blocks above
// 4 KiB and the chunks of one monolithic stream (hsrle_mono_encode.hip.h).  Round 6.
//
// Replaces: src/rle8_extreme_cpu.h:936-1099 (AVX2 body), :111-199 (scalar tail), :203-338 (terminators) as hsrle_encode8p.hip.h does, for the cases that
//           kernel cannot hold in one wave's 64 x 64 bytes -- so far the lane-per-block ring encoders + staging slots + k_compact / k_compact_var.
//
// A unit is walked in WINDOWS of 4 KiB.  What crosses a window's left edge is little, and wave-uniform:
//   * lastRLE and lastSymbol (the decision chain's state -- it already is a pair of scalars carried from round to round in hsrle_encode8p.hip.h);
//   * the start of the run that is open at the edge (a run belongs to the window it ENDS in; its first bytes may lie any number of windows back);
//   * two bits: does the window's first byte continue a run, and did that run start on the byte in front of the edge (a run of exactly two).
// Pass 1 (k_encode8_ppw_scan, one wave per unit, window after window): decisions, the unit's stream size, and per window a state record (stream
// position of its first packet, the three carried values) + one 32 bit record per stored run that ends in it.  Pass 2 (k_encode8_ppw_emit, one wave per
// WINDOW -- the windows of a unit are independent now): the packets of the runs that end in the window, built in LDS as hsrle_encode8p.hip.h builds a
// block's stream, written once.  A packet's literal bytes in front of the window's edge (only the window's first packet can have them; any length) do
// not go through LDS: the wave copies them from the input to their place behind the packet's header.  The trailing literals + terminator are the last
// window's "packet".
#pragma once

#include "hsrle_encode8p.hip.h"

namespace hsrle {

// (PpwArgs, kPpwWindow / kPpwStateWords / kPpwStride / kPpwEmpty, ppw_scratch_bytes: hsrle_launch.h -- the host side needs them too)

// lane l's 64 bytes of the window at `ws` of a unit of n bytes (zeros behind its end)
__device__ __forceinline__ void ppw_load(const uint8_t *__restrict__ d, uint32_t n, uint32_t ws, u32x4 (&x)[4])
{
  const uint32_t base = ws + threadIdx.x * 64u;
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++)
  {
    const uint32_t pos = base + 16u * j;
    u32x4 v = u32x4{ 0, 0, 0, 0 };
    if (pos + 16u <= n) v = ld128(d + pos);
    else if (pos < n) v = load16_edge(d, (int64_t)pos, (uint64_t)n);
    x[j] = v;
  }
}

// 16 image bytes from any offset (dword reads + a byte funnel: LDS dword reads need no 16-byte alignment)
__device__ __forceinline__ u32x4 ppw_img16(const uint8_t *img, uint32_t off)
{
  const uint32_t *const q = (const uint32_t *)(img + (off & ~3u));
  const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], sb = off & 3u;
  return u32x4{ alignbyte(q1, q0, sb), alignbyte(q2, q1, sb), alignbyte(q3, q2, sb), alignbyte(q4, q3, sb) };
}

// One window by one wave.  d / n / nT: the unit's bytes, its length, the bytes from its start to the end of the stream's input (the Packed body / tail rule
// looks at the input's end).  carL / carY / pos / openStart / ended: the state in front of the window, and behind it on return (MODE 0).
// MODE 0: the window's state + records; MODE 1: its packets -> dst (= the unit's stream) + pos.
template <int FAM, int MODE>
__device__ __forceinline__ void ppw_window(const uint8_t *__restrict__ d, uint32_t n, uint32_t nT, uint32_t w, bool hasHeader, bool hasTerm, uint32_t &carL, uint32_t &carY, uint32_t &pos,
                                           uint32_t &openStart, bool &ended, uint32_t *__restrict__ st, uint32_t *__restrict__ myRecs, uint32_t unit, uint32_t unitSize,
                                           uint8_t *__restrict__ dst, PpShared<MODE != 0, true> &sh, const u32x4 (&x)[4], uint32_t recN, uint32_t rec0)
{
  static_assert(FAM == PLAIN || FAM == PACKED, "the two list-free 8 bit multi-symbol codecs");
  constexpr bool PK = FAM == PACKED;
  constexpr uint32_t LONGC = PK ? 11u : 6u;
  constexpr uint32_t TERM = PK ? 9u : 11u;
  const uint32_t lane = threadIdx.x;
  const uint32_t ws = w * kPpwWindow;
  const uint32_t base = lane * 64u;                    // (window relative)
  const bool lastWindow = ws + kPpwWindow >= n;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const bool fromRecs = MODE == 1 && recN != kPpNoRecords;
  const uint32_t carL0 = carL, pos0 = (w == 0u) ? 0u : pos, openIn = openStart;   // (openIn: start of the run that is open at the window's left edge)

  if constexpr (MODE == 0)
  {
    // the state in front of this window (the stored-run count follows at the end)
    if (lane == 0u) { st[0] = pos0; st[1] = carL; st[2] = openIn; st[3] = carY; st[5] = unit; st[6] = w; }
  }
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);

  // ---- 1. equality bits, run starts / ends, candidates: the runs that END in this window ----
  uint32_t R = recN;
  uint64_t candLeft = 0;
  uint32_t candAt = 0;
  if (!fromRecs)
  {
    // the bytes around the window's edges: lane 63 the one behind it, lane 0 the two in front
    uint32_t edge = 0;
    if (lane == 63u && ws + kPpwWindow < n) edge = d[ws + kPpwWindow];
    if (lane == 0u && ws >= 2u) edge = (uint32_t)d[ws - 1u] | ((uint32_t)d[ws - 2u] << 8) | 0x10000u;
    const bool cIn = lane == 0u && (edge & 0x10000u) != 0u && (edge & 0xFFu) == (x[0].x & 0xFFu);               // the window's first byte continues a run
    const bool sIn = cIn && ((edge >> 8) & 0xFFu) != (edge & 0xFFu);                                             // ... that began on the byte in front of it
    const uint32_t nextFirst = wave_shl1(x[0].x, edge);
    uint64_t e64 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const u32x4 a = x[j];
      const uint32_t s = (j < 3u) ? x[j < 3u ? j + 1u : 3u].x : nextFirst;
      e64 |= (uint64_t)zero_mask16(a.x ^ alignbyte(a.y, a.x, 1), a.y ^ alignbyte(a.z, a.y, 1), a.z ^ alignbyte(a.w, a.z, 1), a.w ^ alignbyte(s, a.w, 1)) << (16u * j);
    }
    const uint32_t ab = ws + base;                                          // (position i matches only if its successor exists)
    const uint32_t validBits = (n > ab + 1u) ? ((n - 1u - ab) < 64u ? (n - 1u - ab) : 64u) : 0u;
    e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(e64 >> 63), cIn ? 1u : 0u);
    const uint64_t prev = (e64 << 1) | carry;
    const uint64_t starts = e64 & ~prev;
    const uint64_t ends = ~e64 & prev;
    const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    const int32_t inclStart = wave_scan_max(ownStart);
    const uint32_t carryStart = wave_shr1((uint32_t)inclStart, 0xFFFFFFFFu);
    const uint64_t startPrev63 = (uint64_t)wave_shr1((uint32_t)(starts >> 63), sIn ? 1u : 0u);
    const uint64_t cands = ends & ~((starts << 1) | startPrev63);
    const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
    const uint32_t inclCnt = wave_scan_add(cnt);
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;                            // (0xFFFF: no start in this window in front of the lane)
    candLeft = cands;
    candAt = inclCnt - cnt;
    if constexpr (MODE == 0)
    {
      // the run that is open behind this window began at its last start (or is the one that was open in front of it)
      const int32_t lastStart = (int32_t)wave_lane((uint32_t)inclStart, 63);
      if (lastStart >= 0) openStart = ws + (uint32_t)lastStart;
    }
  }
  wave_sync();

  [[maybe_unused]] auto put_chunks = [&](uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd) __attribute__((always_inline)) {
    // literal bytes [src, src + len) of the WINDOW -> image bytes [ds, ds + len) (hsrle_encode8p.hip.h)
    const uint32_t de = ds + len, D0 = ds & ~15u;
    for (uint32_t t = t0; t < tEnd; t += tStep)
    {
      const uint32_t D = D0 + 16u * t;
      const uint32_t wa = kPpInPad + src + D - ds;
      const uint32_t *const wq = (const uint32_t *)(sh.inb + (wa & ~3u));
      const uint32_t q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3], q4 = wq[4], sb = wa & 3u;
      const u32x4 v = u32x4{ alignbyte(q1, q0, sb), alignbyte(q2, q1, sb), alignbyte(q3, q2, sb), alignbyte(q4, q3, sb) };
      const uint32_t lo = D < ds ? ds - D : 0u, hi = de - D < 16u ? de - D : 16u;
      const u32x4 mh = lds_ld128(sh.mlut + (hi << 4)), ml = lds_ld128(sh.mlut + (lo << 4));
      unsigned long long *const ip = (unsigned long long *)(sh.img + D);
      const uint64_t w0 = (uint64_t)(v.x & mh.x & ~ml.x) | ((uint64_t)(v.y & mh.y & ~ml.y) << 32), w1 = (uint64_t)(v.z & mh.z & ~ml.z) | ((uint64_t)(v.w & mh.w & ~ml.w) << 32);
      atomicOr(ip, w0);
      atomicOr(ip + 1, w1);
    }
  };

  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t imgPos = (hasHeader && w == 0u) ? 9u : 0u;  // MODE 1: image position of the round's first packet
  uint32_t K = 0;                                      // stored runs of this window
  // MODE 1: the window's first packet may reach back in front of the window: its header's length, the literal bytes in front of the edge
  [[maybe_unused]] bool firstSeen = false;
  [[maybe_unused]] uint32_t hlFirst = 0, extLen = 0, imgShift = 0;    // imgShift: the image starts this many bytes in, so that what follows the first header is 16-byte aligned in LDS
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    const bool have = r0 + lane < R;
    const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
    uint32_t p = 0, e = 0, sym = 0, inL = 0, outL = 0;
    bool same = false;
    int k = 0;
    if (fromRecs)
    {
      const uint32_t rec = (r0 == 0u) ? rec0 : (have ? myRecs[r0 + lane] : 0u);
      e = ws + ((rec >> 12) & 0xFFFu) + 1u;
      p = ((rec >> 26) & 1u) ? openIn : ws + (rec & 0xFFFu);
      same = ((rec >> 24) & 1u) != 0u;
      k = have ? 1 + (int)((rec >> 25) & 1u) : 0;
      outL = e;
      inL = wave_shr1(outL, carL);
      if constexpr (MODE != 0) sym = (uint32_t)sh.inb[kPpInPad + (have ? e - 1u - ws : 0u)];
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
      const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;
      wave_sync();
      const uint32_t iq = q >> 6, bit = q & 63u;
      const uint64_t stq = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = stq & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      p = (sBelow != 0ull) ? ws + (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : (cs != 0xFFFFu ? ws + cs : openIn);
      e = ws + q + 1u;
      const uint32_t count = e - p;
      sym = (uint32_t)sh.inb[kPpInPad + q];                                // (the run's last byte: its first may lie in front of the window)
      bool body = true;
      if constexpr (PK)
      {
        // body / tail split of the canonical AVX2 encoder; a run of 2^26 bytes or more is stored under either rule
        const int64_t kk = (int64_t)(count - 1u) / 32;
        body = (e < nT) && ((int64_t)p + 1 + 32 * kk < (int64_t)nT - 32);
      }
      const bool sure = have && count >= LONGC;
      auto decide = [&](uint32_t iL, uint32_t iY, bool &sm) __attribute__((always_inline)) -> int {
        const uint32_t rng = p - iL + 1u;
        sm = false;
        if constexpr (PK)
        {
          if (!body) return (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
          sm = sym == iY;
          const bool emit = count >= 11u || (rng <= 127u && ((sm && count >= 3u) || count >= 4u));     // rle8_extreme_cpu.h:978
          return emit ? (rng <= 127u ? 1 : 2) : 0;
        }
        else
          return (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;                                               // :974
      };
      uint32_t outY = sym, inY = 0;
      outL = e;
      bool outKnown = sure || !have, inKnown = !have;
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), lr = wave_shr1(outL, carL), ls = wave_shr1(outY, carY);
        if (!inKnown && lk != 0u) { inKnown = true; inL = lr; inY = ls; }
        if (inKnown && !outKnown)
        {
          bool sm;
          if (decide(inL, inY, sm) == 0) { outL = inL; outY = inY; }
          else if (PK && !body) outY = inY;
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      k = have ? decide(inL, inY, same) : 0;
      carY = wave_lane(outY, lastLane);
    }
    const uint32_t count = e - p, gap = p - inL, rng = gap + 1u;
    // the part of the literal stretch that lies in this window
    const uint32_t inLw = inL > ws ? inL : ws, pw = p > ws ? p : ws;
    const uint32_t gapImg = pw - inLw;

    // the packet header (rle8_extreme_cpu.h:1000-1058; hsrle_encode8p.hip.h)
    uint64_t hlo = 0; uint32_t hhi = 0, hl = 0;
    const uint32_t cfield = count - (PK ? 2u : 5u);
    const bool small = k == 1 && cfield <= (PK ? 127u : 255u);
    if (k)
    {
      if (small)
      {
        if constexpr (PK) { hlo = same ? (cfield | 0x80u | (rng << 9)) : (cfield | (sym << 8) | (rng << 17)); hl = same ? 2u : 3u; }
        else { hlo = sym | (cfield << 8) | (rng << 16); hl = 3u; }
      }
      else
      {
        uint32_t rv, rn;
        if constexpr (PK)
        {
          const uint32_t sm = same ? 0x80u : 0u;
          const bool wide = cfield > 127u;
          hlo = wide ? ((uint64_t)cfield << 8) | sm : (uint64_t)(cfield | sm);
          hl = wide ? 5u : 1u;
          if (!same) { hlo |= (uint64_t)sym << (8u * hl); hl++; }
          rv = (k == 1) ? (rng << 1) & 0xFFu : ((rng << 1) | 1u);
          rn = (k == 1) ? 1u : 4u;
        }
        else
        {
          const bool wide = cfield > 255u;
          hlo = wide ? (uint64_t)sym | ((uint64_t)cfield << 16) : (uint64_t)(sym | (cfield << 8));
          hl = wide ? 6u : 2u;
          if (k != 1) hl++;
          rv = rng;
          rn = (k == 1) ? 1u : 4u;
        }
        hlo |= (uint64_t)rv << (8u * hl);
        hhi = (hl > 4u) ? (uint32_t)(((uint64_t)rv << 32) >> (96u - 8u * hl)) : 0u;
        hl += rn;
      }
    }
    if constexpr (MODE == 0)
    {
      const uint32_t inclB = wave_scan_add(k ? hl + gap : 0u), inclK = wave_scan_add(k ? 1u : 0u);
      const uint32_t idx = K + inclK - 1u;
      if (k && idx < kPpwStride)
        myRecs[idx] = (p >= ws ? p - ws : 0u) | ((e - 1u - ws) << 12) | (same ? 1u << 24 : 0u) | (k == 2 ? 1u << 25 : 0u) | (p < ws ? 1u << 26 : 0u);
      pos += wave_lane(inclB, 63);
      K += wave_lane(inclK, 63);
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
        uint32_t *const wp = (uint32_t *)(sh.img + (at0 & ~3u));
        if (small)
        {
          const uint64_t hv = (uint64_t)(uint32_t)hlo << (8u * (at0 & 3u));
          atomicOr(wp, (uint32_t)hv);
          if ((uint32_t)(hv >> 32) != 0u) atomicOr(wp + 1, (uint32_t)(hv >> 32));
        }
        else
        {
          const uint32_t sft = 32u - 8u * (at0 & 3u);
          const uint32_t d0 = (uint32_t)hlo, d1 = (uint32_t)(hlo >> 32);
          atomicOr(wp, (uint32_t)(((uint64_t)d0 << 32) >> sft));
          atomicOr(wp + 1, (uint32_t)((((uint64_t)d1 << 32) | d0) >> sft));
          atomicOr(wp + 2, (uint32_t)((((uint64_t)hhi << 32) | d1) >> sft));
          if (hl > 7u) atomicOr(wp + 3, (uint32_t)((uint64_t)hhi >> sft));
        }
        ds = at0 + hl;
        if (gapImg > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)(inLw - ws) | ((uint64_t)ds << 13) | ((uint64_t)gapImg << 26); }
        else if (gapImg != 0u) nch = ((ds + gapImg - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)
      {
        if (t < nch) put_chunks(inLw - ws, ds, gapImg, t, 1u, t + 1u);
        if (t + 1u < nch) put_chunks(inLw - ws, ds, gapImg, t + 1u, 1u, t + 2u);
      }
      imgPos += wave_lane(incl, 63);
    }
    carL = wave_lane(outL, lastLane);
    if (__ballot(k != 0 && e >= nT) != 0ull) ended = true;
  }

  if constexpr (MODE == 0)
  {
    if (lane == 0u) st[4] = (K <= kPpwStride) ? K : kPpNoRecords;
    return;
  }
  else
  {
    // ---- 3. header, terminator + trailing literals (the last window's packet) ----
    const bool term = hasTerm && lastWindow;
    uint32_t imgSize = imgPos;
    if (hasHeader && w == 0u && lane < 8u)
    {
      const uint64_t h = (uint64_t)n | ((uint64_t)unitSize << 32);        // stream header (rle8_extreme_cpu.c:5-15): u32 uncompressed, u32 compressed, u8 mode = 0
      sh.img[lane] = (uint8_t)(h >> (8u * lane));
    }
    uint32_t tailSrc = 0, tailLen = 0;
    if (term)
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
      if (lane >= 16u && lane < 16u + TERM)
      {
        const uint32_t t = lane - 16u;
        uint32_t v = 0;
        if constexpr (PK)
        {
          const uint32_t wv = ended ? 1u : (((kLit + 1u) << 1) | 1u);
          v = (t == 0u) ? 0x80u : (t >= 5u ? (wv >> (8u * (t - 5u))) & 0xFFu : 0u);
        }
        else
        {
          const uint32_t wv = ended ? 0u : kLit + 1u;
          v = (t >= 7u) ? (wv >> (8u * (t - 7u))) & 0xFFu : 0u;
        }
        if (v != 0u) sh.img[imgPos + t] = (uint8_t)v;
      }
      imgSize = imgPos + TERM + tailLen;
    }
    wave_sync();
    {
      const uint32_t nj = sh.jobCount;
      for (uint32_t j = 0; j <= nj; j++)
      {
        uint32_t src, ds, len;
        if (j < nj) { const uint64_t jb = sh.jobs[j]; src = (uint32_t)jb & 0x1FFFu; ds = (uint32_t)(jb >> 13) & 0x1FFFu; len = (uint32_t)(jb >> 26); }
        else { src = tailSrc; ds = imgPos + TERM; len = tailLen; }
        if (len != 0u) put_chunks(src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
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

// what a unit is: its bytes, the stream it belongs to
struct PpwUnit { const uint8_t *d; uint32_t n, nT, y0; bool hasHeader, hasTerm; uint32_t gw0, slots; };
__device__ __forceinline__ bool ppw_unit(const PpwArgs &a, uint32_t u, PpwUnit &q)
{
  if (a.B != 0u)
  {
    const uint64_t at = (uint64_t)u * a.B;
    q.d = a.in + at;
    q.n = (uint32_t)((a.U - at) < (uint64_t)a.B ? (a.U - at) : (uint64_t)a.B);
    q.nT = q.n; q.y0 = 0u; q.hasHeader = true; q.hasTerm = true;
    q.slots = (a.B + kPpwWindow - 1u) / kPpwWindow;
    q.gw0 = u * q.slots;
    return true;
  }
  const uint32_t chunks = a.count[0];
  if (u >= chunks) return false;
  const uint64_t at = a.starts[u], to = a.starts[u + 1u];
  q.d = a.in + at;
  q.n = (uint32_t)(to - at);
  q.nT = (uint32_t)((a.U - at) < 0xFFFFFFFFull ? (a.U - at) : 0xFFFFFFFFull);
  q.y0 = (uint32_t)(a.syms[u] & 0xFFull);
  q.hasHeader = false; q.hasTerm = to >= a.U;
  q.gw0 = (uint32_t)(at >> 12) + u;
  q.slots = (uint32_t)(to >> 12) + u + 1u - q.gw0;
  return true;
}

// Pass 1: one wave per unit
template <int FAM>
__global__ __launch_bounds__(64) void k_encode8_ppw_scan(PpwArgs a)
{
  __shared__ PpShared<false, true> sh;
  constexpr uint32_t TERM = (FAM == PACKED) ? 9u : 11u;
  const uint32_t u = xcd_tile(blockIdx.x, gridDim.x);
  if (u >= a.nUnits) return;
  PpwUnit q;
  if (!ppw_unit(a, u, q))
  {
    // (chunk table: a unit behind the last chunk -- its size is zero, and it marks the window slot nobody else owns)
    if (threadIdx.x == 0u) { a.sizes[u] = 0u; a.states[(uint64_t)((uint32_t)(a.U >> 12) + u) * kPpwStateWords + 4u] = kPpwEmpty; }
    return;
  }
  const uint32_t windows = (q.n + kPpwWindow - 1u) / kPpwWindow;
  uint32_t carL = 0, carY = q.y0, pos = q.hasHeader ? 9u : 0u, openStart = 0;
  bool ended = false;
  u32x4 x[4], xn[4];
  ppw_load(q.d, q.n, 0u, x);
  for (uint32_t w = 0; w < windows; w++)
  {
    if (w + 1u < windows) ppw_load(q.d, q.n, (w + 1u) * kPpwWindow, xn);   // (the next window's bytes are on their way while this one is worked on)
    const uint64_t gw = (uint64_t)q.gw0 + w;
    ppw_window<FAM, 0>(q.d, q.n, q.nT, w, q.hasHeader, q.hasTerm, carL, carY, pos, openStart, ended, a.states + gw * kPpwStateWords, a.recs + gw * kPpwStride, u, 0u, nullptr, sh, x,
                       kPpNoRecords, 0u);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = xn[j];
  }
  if (threadIdx.x == 0u) a.sizes[u] = pos + (q.hasTerm ? TERM + (ended ? 0u : q.n - carL) : 0u);
  for (uint32_t s = windows + threadIdx.x; s < q.slots; s += 64u) a.states[((uint64_t)q.gw0 + s) * kPpwStateWords + 4u] = kPpwEmpty;
}

// Pass 2: one wave per window
template <int FAM>
__global__ __launch_bounds__(64) void k_encode8_ppw_emit(PpwArgs a)
{
  __shared__ PpShared<true, true> sh;
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
  const uint32_t *const st = a.states + (uint64_t)gw * kPpwStateWords;
  const uint32_t *const myRecs = a.recs + (uint64_t)gw * kPpwStride;
  uint32_t sv, rec0, recN, u, w, unitSize = 0;
  uint64_t unitAt = 0;
  PpwUnit q;
  u32x4 x[4];
  if (a.B != 0u)
  {
    // blocks: which block and window this is follows from the window's number -- the state, the first records and the input are asked for together
    // (one memory round trip less in a wave that lives for a handful of them)
    const uint32_t slots = (a.B + kPpwWindow - 1u) / kPpwWindow;
    u = gw / slots; w = gw - u * slots;
    ppw_unit(a, u, q);
    if (w * kPpwWindow >= q.n) return;                                   // (the last, short block: no such window)
    sv = (threadIdx.x < kPpwStateWords) ? st[threadIdx.x] : 0u;
    rec0 = myRecs[threadIdx.x];
    unitSize = a.sizes[u]; unitAt = a.offsets[u];
    ppw_load(q.d, q.n, w * kPpwWindow, x);
    recN = wave_lane(sv, 4);
  }
  else
  {
    sv = (threadIdx.x < kPpwStateWords) ? st[threadIdx.x] : 0u;
    recN = wave_lane(sv, 4);
    if (recN == kPpwEmpty) return;
    rec0 = myRecs[threadIdx.x];
    u = wave_lane(sv, 5); w = wave_lane(sv, 6);
    if (!ppw_unit(a, u, q)) return;
    unitSize = a.sizes[u]; unitAt = a.offsets[u];
    ppw_load(q.d, q.n, w * kPpwWindow, x);
  }
  uint32_t pos = wave_lane(sv, 0), carL = wave_lane(sv, 1), openStart = wave_lane(sv, 2), carY = wave_lane(sv, 3);
  bool ended = false;
  wave_sync();
  ppw_window<FAM, 1>(q.d, q.n, q.nT, w, q.hasHeader, q.hasTerm, carL, carY, pos, openStart, ended, nullptr, const_cast<uint32_t *>(myRecs), u, unitSize, a.payload + unitAt, sh, x, recN, rec0);
}

} // namespace hsrle
