// hsrle_encodeSpw.hip.h -- the position-parallel encoder of hsrle_encodeSp.hip.h (plain / Packed / 3 symbol LUT / Short codecs of 1 .. 8 byte symbols) for BLOCKS OF
// ANY SIZE: a block walked in windows of 4 KiB, as hsrle_encode8pw.hip.h does it for the two 8 bit codecs.  Round 6.
//
// Replaces: src/rleX_extreme_cpu_encode.h:14-609, src/rleX_Xsl.h:114-346, src/rleX_Xsl_short.h:152-1032 as hsrle_encodeSp.hip.h does, for blocks above 4 KiB -- so far the
//           lane-per-block ring encoders + staging slots + k_compact.
//
// What crosses a window's left edge (all wave-uniform): lastRLE, the last stored symbol, the end of the last run found (where the scan resumes), the move-to-front
// list (LUT / Short with three symbols), the start of the match stretch that is open at the edge, and 8 match bits of history (a stretch counts from S set bits on).
// A run belongs to the window in which its match stretch ENDS; its first bytes may lie any number of windows back, its end up to S - 1 bytes behind the window.
// Pass 1 (k_encodeS_ppw_scan): one wave per block, window after window -> sizes[], a state record per window, a 32 bit record per stored run.
// Pass 2 (k_encodeS_ppw_emit): one wave per window -> the packets of the runs that belong to it; the literal bytes of the window's first packet that lie in front of
// the window go from the input straight to their place (hsrle_encode8pw.hip.h).
#pragma once

#include "hsrle_encodeSp.hip.h"
#include "hsrle_encode8pw.hip.h"   // ppw_load, ppw_img16

namespace hsrle {

// (kPpwSStateWords = 16, hsrle_launch.h: posW, lastRLE, openStart, carE, lastSymbol x 2, list x 6, stored runs, unit, window, start of the run that began in front of the window)

struct PpSwCarry
{
  uint32_t carL, carE, pos, openStart;
  uint64_t carY, lA, lB, lY;
  bool ended;
};

template <int FAM, int S, int AL, int MODE>
__device__ __forceinline__ void ppSw_window(const uint8_t *__restrict__ d, uint32_t n, uint32_t w, PpSwCarry &cs, uint32_t pCar, uint32_t *__restrict__ st, uint32_t *__restrict__ myRecs,
                                            uint32_t unit, uint32_t unitSize, uint8_t *__restrict__ dst, PpShared<MODE != 0, true, FAM == LUT3 || FAM == SHORT3> &sh, const u32x4 (&x)[4],
                                            uint32_t recN, uint32_t rec0)
{
  static_assert(FAM == PLAIN || FAM == PACKED || FAM == LUT3 || FAM == SHORT0 || FAM == SHORT1 || FAM == SHORT3, "plain, Packed, 3 symbol LUT, Short with no / a one- / a three-symbol list");
  using TR = Traits<FAM, S, AL>;
  constexpr bool SH = FAM == SHORT0 || FAM == SHORT1 || FAM == SHORT3;
  constexpr bool SH1 = FAM == SHORT1;
  constexpr bool SH3 = FAM == SHORT3;
  constexpr bool PK = FAM == PACKED;
  constexpr bool LUT = FAM == LUT3;
  constexpr bool MTF3 = LUT || SH3;
  constexpr uint32_t SU = (uint32_t)S;
  constexpr uint32_t SHORT = TR::SHORT, MEDIUM = TR::MEDIUM, LONG = TR::LONG, MAXR = TR::MAXRANGE;
  constexpr bool R7 = TR::kRange7;
  constexpr uint32_t TERM = SH ? ((SH1 || SH3) ? 9u : 9u + SU) : (LUT ? 8u : (PK ? 5u : SU + 5u) + (R7 ? 4u : 5u));
  constexpr uint32_t TERM_END = SH ? ((SH1 || SH3) ? 7u : 8u) : (LUT ? 6u : TERM);
  constexpr uint32_t HDR = 8u;
  const uint32_t lane = threadIdx.x;
  const uint32_t ws = w * kPpwWindow;
  const uint32_t base = lane * 64u;                    // (window relative)
  const bool lastWindow = ws + kPpwWindow >= n;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const bool fromRecs = MODE == 1 && recN != kPpNoRecords;
  const uint32_t carL0 = cs.carL, pos0 = (w == 0u) ? 0u : cs.pos, openIn = cs.openStart;

  if constexpr (MODE == 0)
  {
    if (lane == 0u)
    {
      st[0] = pos0; st[1] = cs.carL; st[2] = openIn; st[3] = cs.carE; st[4] = (uint32_t)cs.carY; st[5] = (uint32_t)(cs.carY >> 32);
      st[6] = (uint32_t)cs.lA; st[7] = (uint32_t)(cs.lA >> 32); st[8] = (uint32_t)cs.lB; st[9] = (uint32_t)(cs.lB >> 32); st[10] = (uint32_t)cs.lY; st[11] = (uint32_t)(cs.lY >> 32);
      st[13] = unit; st[14] = w;
    }
  }
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
  // the input image: the window, the 16 bytes in front of it (the symbol of a run that began further back is read from its last period in front of the window)
  // and the 16 + 16 bytes behind it (a run may end, and its symbol be read, up to S - 1 bytes behind the window)
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
    if (ws >= 16u && p < ws - 16u) a = p + ((ws - 16u - p + SU - 1u) / SU) * SU;      // (a later period of the run: its pattern repeats every S bytes)
    return pp_symbol<S>(sh.inb, kPpInPad + a - ws);
  };

  // ---- 1. 64 match bits per lane: m[j] = (d[j] == d[j + S]) ----
  uint32_t R = recN;
  uint64_t candLeft = 0;
  uint32_t candAt = 0;
  if (!fromRecs)
  {
    uint32_t wd[18];
#pragma unroll
    for (int j = 0; j < 4; j++) { wd[4 * j] = x[j].x; wd[4 * j + 1] = x[j].y; wd[4 * j + 2] = x[j].z; wd[4 * j + 3] = x[j].w; }
    wd[16] = wave_shl1(x[0].x, back0.x);                                   // (lane 63: the bytes behind the window)
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
    // position j matches only if j + S < n
    const int64_t vb = (int64_t)n - (int64_t)SU - (int64_t)(ws + base);
    const uint32_t validBits = vb <= 0 ? 0u : (vb >= 64 ? 64u : (uint32_t)vb);
    m64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    // lane 0: the match bits of the 8 positions in front of the window, in the top byte of a dword (what the lane in front would hand over)
    uint32_t histM = 0;
    if (lane == 0u && ws >= 16u)
    {
      // bytes ws - 8 .. ws + 7 and S more: front.z, front.w, x[0]
      const uint64_t f8 = (uint64_t)front.z | ((uint64_t)front.w << 32), a8 = (uint64_t)x[0].x | ((uint64_t)x[0].y << 32), b8 = (uint64_t)x[0].z | ((uint64_t)x[0].w << 32);
#pragma unroll
      for (uint32_t i = 0; i < 8u; i++)
      {
        const uint32_t j = i + SU;                                         // byte i of f8 against byte i + S of (f8, a8, b8)
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
    uint64_t full;
    {
      const uint64_t c2 = m64 & shl_in(m64, 1u, histM);
      if constexpr (S == 1) full = (TR::SMINS >= 3u) ? c2 : m64;
      else if constexpr (S == 2) full = c2;
      else if constexpr (S == 3) full = c2 & shl_in(m64, 2u, histM);
      else
      {
        const uint64_t c4 = c2 & shl_in(c2, 2u, histC2);
        if constexpr (S == 4) full = c4;
        else if constexpr (S == 6) full = c4 & shl_in(c2, 4u, histC2);
        else full = c4 & shl_in(c4, 4u, histC4);
      }
    }
    // (full of the position in front of the window, for the candidate at the window's first position)
    uint32_t histFull;
    if constexpr (S == 1) histFull = (TR::SMINS >= 3u) ? histC2 : histM;
    else if constexpr (S == 2) histFull = histC2;
    else if constexpr (S == 3) histFull = histC2 & (histM << 2);
    else if constexpr (S == 4) histFull = histC4;
    else if constexpr (S == 6) histFull = histC4 & (histC2 << 4);
    else histFull = histC4 & (histC4 << 4);
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

  [[maybe_unused]] auto put_chunks = [&](uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd) __attribute__((always_inline)) {
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
  uint32_t carL = cs.carL, carE = cs.carE;
  uint64_t carY = cs.carY;
  [[maybe_unused]] uint64_t lA = cs.lA, lB = cs.lB, lY = cs.lY;
  uint32_t imgPos = (w == 0u) ? HDR : 0u;             // MODE 1: image position of the round's first packet
  uint32_t pos = cs.pos;                                // MODE 0: stream position
  uint32_t K = 0;
  bool ended = cs.ended;
  uint32_t pCarOut = 0xFFFFFFFFu;                       // MODE 0: the start of the (one) stored run of this window that began in front of it
  [[maybe_unused]] bool firstSeen = false;
  [[maybe_unused]] uint32_t hlFirst = 0, extLen = 0, imgShift = 0;    // imgShift: the image starts this many bytes in, so that what follows the first header is 16-byte aligned in LDS
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    const bool have = r0 + lane < R;
    const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
    uint32_t p = 0, e = 0, inL = 0, outL = 0;
    uint64_t sym = 0;
    bool same = false;
    int k = 0;
    [[maybe_unused]] uint32_t mtf = 0;
    if (fromRecs)
    {
      const uint32_t rec = (r0 == 0u) ? rec0 : (have ? myRecs[r0 + lane] : 0u);
      e = ws + ((rec >> 12) & 0x1FFFu) + 1u;
      p = ((rec >> 28) & 1u) ? pCar : ws + (rec & 0xFFFu);
      same = ((rec >> 25) & 1u) != 0u;
      k = have ? 1 + (int)((rec >> 27) & 1u) : 0;
      if constexpr (MTF3) { mtf = (rec >> 25) & 3u; k = have ? 1 : 0; }
      outL = e;
      inL = wave_shr1(outL, carL);
      if constexpr (SH3)
      {
        const uint32_t cnt = e - p, scu_ = AL ? cnt / SU - TR::SMINS / SU + 2u : cnt - TR::SMINS + 2u;
        if (have && !(p - inL <= TR::SMAXPR && scu_ - 2u <= TR::SMAXPC)) k = 2;
      }
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
      wave_sync();
      const uint32_t qr = have ? (uint32_t)sh.lst[lane] : 0u;              // (window relative) the stretch's match bits end in front of this position
      wave_sync();
      const uint32_t q = ws + qr;
      // the stretch's first position: the last start bit below q -- or, with none in this window, the start of the stretch that is open at the window's edge
      uint32_t s0 = openIn;
      if (qr != 0u)
      {
        const uint32_t qm = qr - 1u, iq = qm >> 6, bit = qm & 63u;
        const uint64_t stq = sh.starts[iq];
        const uint32_t cst = (uint32_t)sh.carryStart[iq];
        const uint64_t sBelow = stq & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
        s0 = (sBelow != 0ull) ? ws + (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : (cst != 0xFFFFu ? ws + cst : openIn);
      }

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

      const bool sure = isRun && (MTF3 || count >= (SH ? TR::SMINL : LONG));
      auto decide = [&](uint32_t iL, uint64_t iY, bool &sm) __attribute__((always_inline)) -> int {
        if constexpr (LUT) { sm = false; return 1; }
        if constexpr (SH)
        {
          const uint32_t gp = p - iL, range = gp + 2u;
          sm = SH1 && sym == iY;
          const uint32_t scu = AL ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u;
          const bool pack1 = gp <= TR::SMAXPR && scu - 2u <= TR::SMAXPC;
          uint32_t pen = (SH1 && !sm) ? SU : 0u;
          if (!pack1) pen += 2u + (range <= TR::SMAXTR ? 0u : 2u) + (scu <= TR::SMAXTC ? 0u : 2u);
          if (!SH3 && !(count >= TR::SMINL || count >= TR::SMINS + pen)) return 0;
          return pack1 ? 1 : 2;
        }
        const uint32_t rng = p - iL + 1u;
        sm = PK && sym == iY;
        const bool shortOk = rng <= MAXR && (PK ? (sm || count >= MEDIUM) : count >= SHORT);
        return shortOk ? 1 : (count >= LONG ? 2 : 0);
      };
      uint64_t outY = sym, inY = 0;
      outL = e;
      bool outKnown = sure || !have, inKnown = !have;
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
        const int32_t tr = wave_scan_max(key);
        const uint64_t y = headList[head ? (uint32_t)tr : 0u];
        mtf = !head ? 0u : (sym == u2 ? 1u : (sym == y ? 2u : 3u));
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
    // the part of the literal stretch that lies in this window
    const uint32_t inLw = inL > ws ? inL : ws, pw = p > ws ? p : ws;
    const uint32_t gapImg = pw - inLw;

    const uint32_t cfield = LUT ? (AL ? count / SU - 3u / SU + 2u : count - 1u) : (AL ? count / SU - SHORT / SU + 1u : count - SHORT + 1u);
    const uint32_t cMax = (PK || LUT) ? 127u : 255u;
    // (blocks above 64 KiB: counts and ranges beyond 16 bits exist -- the LUT / Short forms then say 0 in the packed field and a 32 bit field follows: rleX_Xsl.h:190-250,
    //  rleX_Xsl_short.h:216-357; the host keeps these codecs to blocks below 1 MiB, where the reference's penalty thresholds (0xFFFFF) cannot be reached)
    const uint32_t cBytes = LUT ? (cfield <= 127u ? 0u : (cfield <= 0xFFFFu ? 2u : 4u)) : (cfield <= cMax ? 1u : 5u);
    const uint32_t sBytes = MTF3 ? (mtf == 3u ? SU : 0u) : (((PK || SH1) && same) ? 0u : SU);
    const uint32_t rBytes = LUT ? (rng <= 127u ? 0u : (rng <= 0xFFFFu ? 2u : 4u)) : ((k == 1) ? 1u : (R7 ? 4u : 5u));
    [[maybe_unused]] const uint32_t scu = SH ? (AL ? count / SU - TR::SMINS / SU + 2u : count - TR::SMINS + 2u) : 0u;
    [[maybe_unused]] const uint32_t srange = gap + 2u;
    const uint32_t hl = !k ? 0u : (SH ? ((k == 1) ? 1u : 3u + (scu > TR::SMAXTC ? (scu <= 0xFFFFu ? 2u : 4u) : 0u) + (srange > TR::SMAXTR ? (srange <= 0xFFFFu ? 2u : 4u) : 0u)) + sBytes
                                      : (LUT ? 2u : 0u) + cBytes + sBytes + rBytes);
    if constexpr (MODE == 0)
    {
      const uint32_t inclB = wave_scan_add(k ? hl + gap : 0u), inclK = wave_scan_add(k ? 1u : 0u);
      const uint32_t idx = K + inclK - 1u;
      if (k && idx < kPpwStride)
        myRecs[idx] = (p >= ws ? p - ws : 0u) | ((e - 1u - ws) << 12) | (MTF3 ? mtf << 25 : ((same ? 1u << 25 : 0u) | (k == 2 ? 1u << 27 : 0u))) | (p < ws ? 1u << 28 : 0u);
      const uint64_t carried = __ballot(k != 0 && p < ws);
      if (carried != 0ull) pCarOut = wave_lane(p, (int)__builtin_ctzll(carried));
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
        uint32_t a = at0;
        if constexpr (LUT)
        {
          const uint32_t c7 = cfield <= 127u ? cfield : (cfield <= 0xFFFFu ? 1u : 0u), r7 = rng <= 127u ? rng : (rng <= 0xFFFFu ? 1u : 0u);   // (1: a 16 bit field follows; 0: a 32 bit one)
          pp_or_bytes(sh.img, a, (uint64_t)((mtf << 14) | (c7 << 7) | r7), 2u); a += 2u;
          if (mtf == 3u) { pp_or_bytes(sh.img, a, sym, SU); a += SU; }
          if (cBytes) { pp_or_bytes(sh.img, a, (uint64_t)cfield, cBytes); a += cBytes; }
          if (rBytes) pp_or_bytes(sh.img, a, (uint64_t)rng, rBytes);
        }
        else if constexpr (SH)
        {
          const uint32_t mi = (SH3 ? mtf : ((SH1 && !same) ? 1u : 0u)) << (TR::SCB + TR::SRBP);
          if (k == 1) { pp_or_bytes(sh.img, a, (uint64_t)(mi | ((scu - 2u) << TR::SRBP) | gap), 1u); a += 1u; }
          else
          {
            const uint32_t scx = scu <= TR::SMAXTC ? scu : (scu <= 0xFFFFu ? 1u : 0u), rx = srange <= TR::SMAXTR ? srange : (srange <= 0xFFFFu ? 1u : 0u);
            const uint32_t f = scx << (TR::SRB - 8u);
            const uint32_t b0 = (mi | (TR::SCINV << TR::SRBP) | (f >> 8)) & 0xFFu, b1 = (f | (rx >> 8)) & 0xFFu, b2 = rx & 0xFFu;
            pp_or_bytes(sh.img, a, (uint64_t)(b0 | (b1 << 8) | (b2 << 16)), 3u); a += 3u;
            if (scx != scu) { const uint32_t nb = scu <= 0xFFFFu ? 2u : 4u; pp_or_bytes(sh.img, a, (uint64_t)scu, nb); a += nb; }
            if (rx != srange) { const uint32_t nb = srange <= 0xFFFFu ? 2u : 4u; pp_or_bytes(sh.img, a, (uint64_t)srange, nb); a += nb; }
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
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  if constexpr (MODE == 0)
  {
    if (lane == 0u) { st[12] = (K <= kPpwStride) ? K : kPpNoRecords; st[15] = pCarOut; }
    cs.carL = carL; cs.carE = carE; cs.carY = carY; cs.pos = pos; cs.ended = ended;
    if constexpr (MTF3) { cs.lA = lA; cs.lB = lB; cs.lY = lY; }
    return;
  }
  else
  {
    // ---- 3. header, terminator + trailing literals (the last window's packet) ----
    uint32_t imgSize = imgPos;
    if (w == 0u && lane < 8u)
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
        const uint32_t at = imgPos;
        if constexpr (SH)
        {
          pp_or_bytes(sh.img, at, (uint64_t)((TR::SCINV << TR::SRBP) | (TR::STB << 8) | (ended ? 1u << 16 : 0u)), 3u);
          if (!ended) pp_or_bytes(sh.img, at + 5u, (uint64_t)(kLit + 2u), 4u);
        }
        else if constexpr (LUT)
        {
          sh.img[at] = ended ? 0x81 : 0x80;
          if (!ended) pp_or_bytes(sh.img, at + 4u, (uint64_t)(kLit + 2u), 4u);
        }
        else
        {
          const uint32_t a0 = at + (PK ? 0u : SU);
          if constexpr (PK) sh.img[a0] = 0x80;
          const uint32_t val = ended ? 0u : kLit + 1u;
          if constexpr (R7) pp_or_bytes(sh.img, a0 + 5u, (uint64_t)((val << 1) | 1u), 4u);
          else pp_or_bytes(sh.img, a0 + 5u, (uint64_t)val << 8, 5u);
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

// Pass 1: one wave per block
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encodeS_ppw_scan(PpwArgs a)
{
  __shared__ PpShared<false, true, FAM == LUT3 || FAM == SHORT3> sh;
  using TR = Traits<FAM, S, AL>;
  constexpr bool SH = FAM == SHORT0 || FAM == SHORT1 || FAM == SHORT3;
  constexpr bool SH1 = FAM == SHORT1, SH3 = FAM == SHORT3, PK = FAM == PACKED, LUT = FAM == LUT3;
  constexpr uint32_t SU = (uint32_t)S;
  constexpr bool R7 = TR::kRange7;
  constexpr uint32_t TERM = SH ? ((SH1 || SH3) ? 9u : 9u + SU) : (LUT ? 8u : (PK ? 5u : SU + 5u) + (R7 ? 4u : 5u));
  constexpr uint32_t TERM_END = SH ? ((SH1 || SH3) ? 7u : 8u) : (LUT ? 6u : TERM);
  constexpr uint64_t SMASK = (S >= 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);
  const uint32_t u = xcd_tile(blockIdx.x, gridDim.x);
  if (u >= a.nUnits) return;
  const uint64_t at = (uint64_t)u * a.B;
  const uint8_t *const d = a.in + at;
  const uint32_t n = (uint32_t)((a.U - at) < (uint64_t)a.B ? (a.U - at) : (uint64_t)a.B);
  const uint32_t slots = (a.B + kPpwWindow - 1u) / kPpwWindow, windows = (n + kPpwWindow - 1u) / kPpwWindow;
  const uint64_t gw0 = (uint64_t)u * slots;
  PpSwCarry cs;
  cs.carL = 0u; cs.carE = 0u; cs.pos = 8u; cs.openStart = 0u; cs.carY = 0ull; cs.ended = false;
  cs.lA = 0ull; cs.lB = 0x7F7F7F7F7F7F7F7Full & SMASK; cs.lY = 0xFFFFFFFFFFFFFFFFull & SMASK;
  u32x4 x[4], xn[4];
  ppw_load(d, n, 0u, x);
  for (uint32_t w = 0; w < windows; w++)
  {
    if (w + 1u < windows) ppw_load(d, n, (w + 1u) * kPpwWindow, xn);
    ppSw_window<FAM, S, AL, 0>(d, n, w, cs, 0u, a.states + (gw0 + w) * kPpwSStateWords, a.recs + (gw0 + w) * kPpwStride, u, 0u, nullptr, sh, x, kPpNoRecords, 0u);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = xn[j];
  }
  if (threadIdx.x == 0u) a.sizes[u] = cs.pos + (cs.ended ? TERM_END : TERM + (n - cs.carL));
}

// Pass 2: one wave per window
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encodeS_ppw_emit(PpwArgs a)
{
  __shared__ PpShared<true, true, FAM == LUT3 || FAM == SHORT3> sh;
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
  const uint32_t *const st = a.states + (uint64_t)gw * kPpwSStateWords;
  const uint32_t *const myRecs = a.recs + (uint64_t)gw * kPpwStride;
  const uint32_t slots = (a.B + kPpwWindow - 1u) / kPpwWindow;
  const uint32_t u = gw / slots, w = gw - u * slots;
  const uint64_t at = (uint64_t)u * a.B;
  const uint8_t *const d = a.in + at;
  const uint32_t n = (uint32_t)((a.U - at) < (uint64_t)a.B ? (a.U - at) : (uint64_t)a.B);
  if (w * kPpwWindow >= n) return;                                       // (the last, short block: no such window)
  const uint32_t sv = (threadIdx.x < kPpwSStateWords) ? st[threadIdx.x] : 0u;
  const uint32_t rec0 = myRecs[threadIdx.x];
  const uint32_t unitSize = a.sizes[u];
  const uint64_t unitAt = a.offsets[u];
  u32x4 x[4];
  ppw_load(d, n, w * kPpwWindow, x);
  const uint32_t recN = wave_lane(sv, 12);
  PpSwCarry cs;
  cs.pos = wave_lane(sv, 0); cs.carL = wave_lane(sv, 1); cs.openStart = wave_lane(sv, 2); cs.carE = wave_lane(sv, 3);
  cs.carY = (uint64_t)wave_lane(sv, 4) | ((uint64_t)wave_lane(sv, 5) << 32);
  cs.lA = (uint64_t)wave_lane(sv, 6) | ((uint64_t)wave_lane(sv, 7) << 32);
  cs.lB = (uint64_t)wave_lane(sv, 8) | ((uint64_t)wave_lane(sv, 9) << 32);
  cs.lY = (uint64_t)wave_lane(sv, 10) | ((uint64_t)wave_lane(sv, 11) << 32);
  cs.ended = false;
  wave_sync();
  ppSw_window<FAM, S, AL, 1>(d, n, w, cs, wave_lane(sv, 15), nullptr, const_cast<uint32_t *>(myRecs), u, unitSize, a.payload + unitAt, sh, x, recN, rec0);
}

} // namespace hsrle
