// hsrle_encode128.hip.h -- the 128 bit encoders (rle128_{sym,byte}[_packed]) on the ring data path.
//
// Replaces: src/rle128_extreme_cpu.h:32-497 (restated in oracle/hsrle_oracle.c and, one lane with global loads, in
//           hsrle_encode.hip.h: encode_block_128 -- the first-generation kernel, still used for the one-block drop-in path).
//
// One lane = one block, the ring encoders' top-up and output accumulator (hsrle_encode8.hip.h), literal stretches that have left the ring
// noted and copied by the wave (hsrle_encode8s.hip.h).  Run detection on bits: E[j] = (d[j] == d[j + 16]), sixteen bits per landed pair
// of neighbouring chunks (the chunks are aligned: one SWAR compare), kept in two registers per lane for the last 128 positions.
//   * a run starts at the first p >= the search position with E[p .. p + 16) all set (the reference's pair search, :233-268, hops there:
//     behind the highest differing byte of its window each time; the hops are replayed on the bits -- where they stand at n - 32
//     matters); with L = the set bits from p on, it ends at p + 16 + L (byte-aligned: the matching leading
//     bytes of the next symbol count, :214-228) or at p + 16 + 16 (L / 16) (sym-aligned);
//   * the block starts INSIDE a run of its first symbol (count 0, symbol not inverted: q4), i.e. a run at p = 0 that needs no pair.
// This holds as long as the end of the block is far: the extension wants i < n - 16, the pair search i < n - 32, and behind that the
// reference walks byte by byte with a symbol that is re-read at every step (:270-300).  So the bit scanner works up to n - 48 and hands
// its state (in a run / searching, position, symbol, count) to a literal restatement of the reference's loop for the last bytes.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low_m, wave_sync

namespace hsrle {

template <bool PK, int AL>
__global__ __launch_bounds__(64) void k_encode128_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *__restrict__ slots, uint32_t slotStride,
                                                         uint32_t *__restrict__ sizes)
{
  // (every lambda is always_inline: one that stays a function keeps all it captures by reference in scratch memory)
  using TR = Traits<PK ? PACKED : PLAIN, 16, AL>;
  constexpr int Q = 64;
  constexpr int H = 256;
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;

  __shared__ __attribute__((aligned(16))) uint8_t hist[64 * H];
  __shared__ __attribute__((aligned(16))) uint32_t rinfo[64];
  __shared__ __attribute__((aligned(16))) uint8_t accScratch[64 * 16];
  __shared__ __attribute__((aligned(16))) uint8_t mlut[16 * 16];
  if (threadIdx.x < 16u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (uint32_t)part, p1 = (uint32_t)(part >> 32);
    lds_st128(mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }
  wave_sync();
#define HS_MERGE128(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = xcd_tile(blockIdx.x, gridDim.x) * 64u;
  const uint32_t b = wgFirst + lane;
  const bool active = b < nBlocks;
  auto hsw_of = [](uint32_t r) __attribute__((always_inline)) -> uint32_t { return (r & 7u) << 4; };
  const uint32_t hbase = (lane * (uint32_t)H) ^ hsw_of(lane);

  uint32_t n = 0;
  const uint64_t blockAt = (uint64_t)b * B;
  if (active) n = (uint32_t)((U - blockAt) < (uint64_t)B ? (U - blockAt) : (uint64_t)B);
  uint8_t *const slot = slots + (uint64_t)b * slotStride;

  uint32_t avail = 0;
  bool finished = !active;

  // ---- output: 16-byte accumulator + stream position; noted literal stretches (see hsrle_encode8.hip.h) ----
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  u32x4 oacc = zero4;
  uint32_t opos = 0;
  uint32_t pendSrc = 0, pendDst = 0, pendBytes = 0, pend2Src = 0, pend2Dst = 0, pend2Bytes = 0;
  auto store_bytes = [&](uint8_t *p, u32x4 w, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
    const uint64_t w0 = (uint64_t)w.x | ((uint64_t)w.y << 32), w1 = (uint64_t)w.z | ((uint64_t)w.w << 32);
    for (uint32_t k = lo; k < hi; k++)
      p[k] = (uint8_t)((k < 8u ? w0 >> (8u * k) : w1 >> (8u * (k - 8u))) & 0xFFull);
  };
  auto append = [&](u32x4 hv, uint32_t nb) __attribute__((always_inline)) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);
    const u32x4 w = HS_MERGE128(oacc, lowp, c);
    if (c + nb >= 16u)
    {
      st128(slot + (opos & ~15u), w);
      oacc = (c == 0u) ? zero4 : funnel16(hv, zero4, 16u - c);
    }
    else
      oacc = w;
    opos += nb;
  };
  uint64_t hlo = 0;
  uint32_t hhi = 0, hn = 0;
  auto hpush = [&](uint32_t v, uint32_t k) __attribute__((always_inline)) {
    const uint32_t sh = hn * 8u;
    if (hn < 8u)
    {
      hlo |= (uint64_t)v << sh;
      if (hn + k > 8u) hhi |= v >> (64u - sh);
    }
    else
      hhi |= v << (sh - 64u);
    hn += k;
  };
  auto hb = [&](uint32_t v) __attribute__((always_inline)) { hpush(v & 0xFFu, 1u); };
  auto h32 = [&](uint32_t v) __attribute__((always_inline)) { hpush(v, 4u); };
  auto hflush = [&]() __attribute__((always_inline)) {
    if (hn != 0u) append(u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), hhi, 0u }, hn);
    hlo = 0; hhi = 0; hn = 0;
  };
  auto ring_win = [&](uint32_t p) __attribute__((always_inline)) -> u32x4 {
    const uint32_t a0 = p & ~15u;
    return funnel16(lds_ld128(hist + (hbase ^ (a0 & HM))), lds_ld128(hist + (hbase ^ ((a0 + 16u) & HM))), p & 15u);
  };
  auto in_ring = [&](uint32_t p) __attribute__((always_inline)) -> bool { return p + (uint32_t)H >= avail + 16u; };
  // 16 input bytes at block position p (bytes at or beyond n read as the ring / memory has them: callers mask)
  auto bytes16 = [&](uint32_t p) __attribute__((always_inline)) -> u32x4 {
    if (__builtin_expect(in_ring(p), 1)) return ring_win(p);
    lds_st128(accScratch + lane * 16u, global_window16(in, blockAt, U, p));
    return lds_ld128(accScratch + lane * 16u);
  };
  auto emit_literals = [&](uint32_t from, uint32_t len) __attribute__((always_inline)) {
    if (len == 0u) return;
    const uint32_t c = opos & 15u, total = c + len;
    const uint32_t srcp = from - c;
    uint8_t *const dst = slot + (opos & ~15u);
    if (__builtin_expect(in_ring(from), 1))
    {
      u32x4 w = HS_MERGE128(oacc, ring_win(srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        st128(dst + k, w);
        k += 16u;
        if (k < total) w = ring_win(srcp + k);
      }
      oacc = w;
    }
    else if (len >= kNotedLiteralMin && (pendBytes == 0u || pend2Bytes == 0u))
    {
      if (c != 0u) store_bytes(dst, oacc, 0u, c);
      const uint32_t tail = (opos + len) & 15u, noted = len - tail;
      if (pendBytes == 0u) { pendSrc = from; pendDst = opos; pendBytes = noted; }
      else { pend2Src = from; pend2Dst = opos; pend2Bytes = noted; }
      lds_st128(accScratch + lane * 16u, tail != 0u ? global_window16(in, blockAt, U, from + noted) : zero4);
      oacc = lds_ld128(accScratch + lane * 16u);
    }
    else
    {
      u32x4 w = HS_MERGE128(oacc, global_window16(in, blockAt, U, srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        st128(dst + k, w);
        k += 16u;
        if (k < total) w = global_window16(in, blockAt, U, srcp + k);
      }
      lds_st128(accScratch + lane * 16u, w);
      oacc = lds_ld128(accScratch + lane * 16u);
    }
    opos += len;
  };
  auto coop_flush = [&](bool all) __attribute__((always_inline)) {
    if (__builtin_expect(__ballot(pendBytes != 0u && (all || pend2Bytes != 0u)) == 0ull, 1)) return;
#pragma unroll 1
    for (int slotNo = 0; slotNo < 2; slotNo++)
    {
      const uint32_t ps = slotNo ? pend2Src : pendSrc, pd = slotNo ? pend2Dst : pendDst, pb = slotNo ? pend2Bytes : pendBytes;
      uint64_t todo = __ballot(pb != 0u);
      while (todo != 0ull)
      {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        uint32_t l1 = l0;
        if (todo != 0ull) { l1 = (uint32_t)__builtin_ctzll(todo); todo &= todo - 1ull; }
        const uint8_t *const sp0 = in + (uint64_t)(wgFirst + l0) * B + (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l0);
        const uint8_t *const sp1 = in + (uint64_t)(wgFirst + l1) * B + (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l1);
        uint8_t *const dp0 = slots + (uint64_t)(wgFirst + l0) * slotStride + (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l0);
        uint8_t *const dp1 = slots + (uint64_t)(wgFirst + l1) * slotStride + (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l1);
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l0);
        const uint32_t n1 = (l1 != l0) ? (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l1) : 0u;
        const uint32_t f0 = n0 & ~15u, f1 = n1 & ~15u;
        const uint32_t nmax = f0 > f1 ? f0 : f1;
        for (uint32_t k = lane * 16u; k < nmax; k += 2048u)
        {
          u32x4 a0, a1, b0, b1;
          const bool ha0 = k < f0, ha1 = k + 1024u < f0, hb0 = k < f1, hb1 = k + 1024u < f1;
          if (ha0) a0 = ld128(sp0 + k);
          if (ha1) a1 = ld128(sp0 + k + 1024u);
          if (hb0) b0 = ld128(sp1 + k);
          if (hb1) b1 = ld128(sp1 + k + 1024u);
          if (ha0) st128(dp0 + k, a0);
          if (ha1) st128(dp0 + k + 1024u, a1);
          if (hb0) st128(dp1 + k, b0);
          if (hb1) st128(dp1 + k + 1024u, b1);
        }
        const uint32_t r = lane & 15u;                                    // the last n & 15 bytes: lanes 0..15 / 16..31
        if (lane < 16u) { if (r < (n0 & 15u)) dp0[f0 + r] = sp0[f0 + r]; }
        else if (lane < 32u) { if (r < (n1 & 15u)) dp1[f1 + r] = sp1[f1 + r]; }
      }
    }
    pendBytes = 0u; pend2Bytes = 0u;
  };

  // ---- stream header ----
  if (active) { h32(n); h32(0); hflush(); }

  // ---- input top-up (as k_encode8_blocks) ----
  auto publish = [&](uint32_t v) __attribute__((always_inline)) { rinfo[(lane % (uint32_t)RPL) * (uint32_t)LPR + lane / (uint32_t)RPL] = v; };
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0, landedAt = 0, landedChunks = 0;

  // ---- scanner state ----
  const int32_t bulkEnd = (int32_t)n - 48;   // the bit scanner's territory: run starts and search positions below this
  int32_t i = 0;                             // searching: the search position; in a run: set bits are known on [runP, i)
  int32_t runP = 0;                          // start of the run the scanner is in
  bool inRun = true;                         // the block starts inside a run of its first symbol
  bool bulkDone = !(0 < bulkEnd);            // hand-over to the tail has happened (or there is no bulk)
  int32_t tailMode = 0;                      // hand-over state: 0 = in the extension loop, 1 = in the pair search
  int32_t count = 0, lastRLE = 0;
  u32x4 symbol = zero4, last = zero4;
  bool haveSymbol = false;                   // the first symbol is read once its chunk has landed
  uint64_t M0 = 0, M1 = 0;                   // E bits of [mEnd - 128, mEnd)
  uint32_t mEnd = 0;

  auto issue = [&]() __attribute__((always_inline)) {
    const uint32_t left = (n > avail) ? (n - avail + 15u) >> 4 : 0u;
    wantReq = umin((uint32_t)LPR, left);
    publish(wantReq != 0u ? (avail | wantReq) : 0u);
    wave_sync();
    uint32_t ri[LPR];
#pragma unroll
    for (int q = 0; q < LPR; q++) ri[q] = rinfo[(lane / LPR) * LPR + q];
    wave_sync();
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = lane % LPR;
      const uint32_t nreq = ri[q] & 15u, e = ri[q] & ~15u;
      const bool valid = c < nreq;
      const uint64_t g = (uint64_t)(wgFirst + r) * B + e + c * 16u;
      u32x4 v = u32x4{ 0, 0, 0, 0 };
      if (valid)
      {
        if (g + 16u <= U)
          v = ld128(in + g);
        else
          v = load16_edge(in, (int64_t)g, U);
      }
      pf[q] = v;
      pfAt[q] = (r * (uint32_t)H) ^ hsw_of(r) ^ ((e + c * 16u) & HM);
    }
  };
  auto land = [&]() __attribute__((always_inline)) {
    // the ring keeps what the scanner may still read: from 32 bytes in front of its position (run start symbol: read when the run is found)
    const uint32_t pos = (uint32_t)i;
    const uint32_t keep = ((pos & ~15u) >= 48u) ? (pos & ~15u) - 48u : 0u;
    const uint32_t held = avail - umin(keep, avail);
    const uint32_t fit = (held >= (uint32_t)H) ? 0u : ((uint32_t)H - held) >> 4;
    const uint32_t take = umin(wantReq, fit);
    publish(take);
    wave_sync();
    uint32_t ri[LPR];
#pragma unroll
    for (int q = 0; q < LPR; q++) ri[q] = rinfo[(lane / LPR) * LPR + q];
    wave_sync();
#pragma unroll
    for (int q = 0; q < LPR; q++)
      if (lane % LPR < ri[q])
        lds_st128(hist + pfAt[q], pf[q]);
    landedAt = avail; landedChunks = take;
    avail = umin(avail + (take << 4), n);
  };
  // E bits of the chunks in front of the newly landed ones: chunk c gets its bits when chunk c + 1 is there
  auto extend_masks = [&]() __attribute__((always_inline)) {
    uint64_t fresh = 0;
    uint32_t got = 0;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)LPR; j++)
    {
      const uint32_t at = landedAt + 16u * j;                             // the chunk that landed; bits are for the chunk at - 16
      if (j < landedChunks && at >= 16u)
      {
        const u32x4 x = lds_ld128(hist + (hbase ^ ((at - 16u) & HM))), y = lds_ld128(hist + (hbase ^ (at & HM)));
        const uint32_t zm16 = zero_mask16(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
        fresh |= (uint64_t)zm16 << (16u * got);
        got++;
      }
    }
    const uint32_t sh = 16u * got;
    if (sh == 64u) { M0 = M1; M1 = fresh; }
    else if (sh != 0u)
    {
      M0 = (M0 >> sh) | (M1 << (64u - sh));
      M1 = (M1 >> sh) | (fresh << (64u - sh));
    }
    mEnd += sh;
  };
  // 64 E bits from pos on; bits at or beyond mEnd read 0.  pos >= mEnd - 128 (the scanner never falls back).
  auto bits64 = [&](int32_t pos) __attribute__((always_inline)) -> uint64_t {
    const uint32_t u = (uint32_t)(pos - ((int32_t)mEnd - 128));
    return (u < 64u) ? ((M0 >> u) | ((u != 0u) ? (M1 << (64u - u)) : 0ull)) : ((u < 128u) ? (M1 >> (u - 64u)) : 0ull);
  };

  // ---- packets (rle128_extreme_cpu.h:40-170; thresholds: SURVEY.md A.2) ----
  auto sym_eq4 = [](u32x4 a, u32x4 c) __attribute__((always_inline)) -> bool { return ((a.x ^ c.x) | (a.y ^ c.y) | (a.z ^ c.z) | (a.w ^ c.w)) == 0u; };
  // the run [at - cnt, at) of `symbol`: decide, and if it is stored write the packet.  Returns whether it was stored.
  auto judge = [&](int32_t at, int32_t cnt) __attribute__((always_inline)) -> bool {
    const uint32_t range = (uint32_t)(at - lastRLE - cnt + 1);
    const bool same = PK ? sym_eq4(symbol, last) : false;
    bool shortOk;
    if constexpr (!PK) shortOk = range <= TR::MAXRANGE && (uint32_t)cnt >= TR::SHORT;
    else shortOk = range <= TR::MAXRANGE && (((uint32_t)cnt >= TR::SHORT && same) || (uint32_t)cnt >= TR::MEDIUM);
    const int k = shortOk ? 1 : ((uint32_t)cnt >= TR::LONG ? 2 : 0);
    if (k == 0) return false;
    const uint32_t c = TR::kAligned ? (uint32_t)cnt / 16u - TR::SHORT / 16u + 1u : (uint32_t)cnt - TR::SHORT + 1u;
    if constexpr (!PK)
    {
      append(symbol, 16u);
      if (c <= 255u) hb(c); else { hb(0); h32(c); }
    }
    else
    {
      const uint32_t sm = same ? 0x80u : 0u;
      if (c <= 127u) hb(c | sm); else { hb(sm); h32(c); }
      if (!same) { hflush(); append(symbol, 16u); }
      last = symbol;
    }
    if constexpr (TR::kRange7)
    {
      if (k == 1) hb(range << 1); else h32((range << 1) | 1u);
    }
    else
    {
      if (k == 1) hb(range); else { hb(0); h32(range); }
    }
    hflush();
    emit_literals((uint32_t)lastRLE, (uint32_t)(at - cnt - lastRLE));
    return true;
  };
  auto term_head = [&]() __attribute__((always_inline)) {
    if constexpr (!PK) { append(zero4, 16u); hb(0); h32(0); }
    else { hb(0x80); h32(0); }
  };

  issue();
  land();
  wave_sync();
  extend_masks();

  uint32_t stepsLeft = 2u * (B / (uint32_t)Q) + 64u;
  uint32_t tripsLeft = 4u * B + 1024u;

  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    issue();

    // ---- the bit scanner: one trip = at most one run end (the only place that writes a packet) ----
    for (;;)
    {
      // bits are known below mEnd; a search window needs 16 of them, the extension at least one
      const bool can = !bulkDone && tripsLeft != 0u && (inRun ? (avail >= 16u && ((uint32_t)i < mEnd || !(i < bulkEnd))) : ((uint32_t)i + 16u <= mEnd));
      if (__ballot(can) == 0ull) break;
      if (can)
      {
        tripsLeft--;
        if (!haveSymbol) { symbol = ring_win(0u); haveSymbol = true; }
        if (!inRun)
        {
          // The pair search (:233-268) as it hops: at i the window E[i .. i + 16) is all set (a run starts), or i moves behind the highest
          // byte that differs.  The hops end at the first all-set window whatever their way there -- but WHERE the search stands when it
          // reaches n - 32 decides what the byte-wise loop behind it sees, so the hops are taken exactly (from one 64-bit field per pass).
          const uint64_t v = bits64(i);
          const uint32_t known = umin(64u, mEnd - (uint32_t)i);             // bits of v that are known (>= 16 here)
          uint32_t off = 0;
          bool found = false;
          while (off + 16u <= known && i + (int32_t)off < bulkEnd)
          {
            const uint32_t w = (uint32_t)(v >> off) & 0xFFFFu;
            if (w == 0xFFFFu) { found = true; break; }
            off += 32u - (uint32_t)__builtin_clz((~w) & 0xFFFFu);            // highest differing byte + 1
          }
          i += (int32_t)off;
          if (found)
          {
            runP = i;
            symbol = bytes16((uint32_t)runP);
            i = runP + 16;                                                  // set bits are known on [runP, i)
            inRun = true;
          }
          else if (!(i < bulkEnd)) { bulkDone = true; tailMode = 1; }       // the pair search goes on in the tail, from exactly here
        }
        if (inRun && !bulkDone)
        {
          if (i < bulkEnd && (uint32_t)i < mEnd)
          {
            // set bits from i on, among the known ones below bulkEnd
            const uint64_t v = bits64(i);
            const uint32_t known = umin(umin(64u, mEnd - (uint32_t)i), (uint32_t)(bulkEnd - i));
            const uint32_t t = (v == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~v);
            if (t >= known) i += (int32_t)known;                            // all set as far as known: the run goes on
            else
            {
              const int32_t L = i + (int32_t)t - runP;
              const int32_t e = runP + 16 + (TR::kAligned ? (L / 16) * 16 : L);
              count = e - runP;
              if (judge(e, count)) lastRLE = e;
              i = e;
              inRun = false;
              if (!(i < bulkEnd)) { bulkDone = true; tailMode = 1; }
            }
          }
          if (inRun && !(i < bulkEnd))
          {
            // the run reaches the tail's territory: hand over inside the extension loop, behind the last whole symbol that is known
            // to match (the reference stands at such a position: every trip of :205-230 that led there had i < n - 16)
            const int32_t at = runP + 16 + ((i - runP) / 16) * 16;
            count = at - runP; i = at;
            bulkDone = true; tailMode = 0;
          }
        }
      }
      coop_flush(false);
    }

    // ---- the last bytes: the reference's loop as it is (:205-300), from the state the scanner left ----
    if (!finished && bulkDone && avail >= n)
    {
      const int32_t nn = (int32_t)n;
      if (!haveSymbol) { symbol = (nn >= 16) ? ring_win(0u) : zero4; haveSymbol = true; }
      bool enterSearch = tailMode == 1;
      uint32_t guard = 4096u;
      while ((i < nn || enterSearch) && guard-- != 0u)
      {
        bool restart = true;
        while (restart)
        {
          restart = false;
          if (!enterSearch)
          {
            while (i < nn - 16)
            {
              const u32x4 x = bytes16((uint32_t)i);
              const uint32_t z0 = x.x ^ symbol.x, z1 = x.y ^ symbol.y, z2 = x.z ^ symbol.z, z3 = x.w ^ symbol.w;
              if ((z0 | z1 | z2 | z3) == 0u) { count += 16; i += 16; }
              else
              {
                if constexpr (!TR::kAligned)
                {
                  int32_t off;
                  if (z0) off = (int32_t)(__builtin_ctz(z0) >> 3);
                  else if (z1) off = 4 + (int32_t)(__builtin_ctz(z1) >> 3);
                  else if (z2) off = 8 + (int32_t)(__builtin_ctz(z2) >> 3);
                  else off = 12 + (int32_t)(__builtin_ctz(z3) >> 3);
                  i += off; count += off;
                }
                break;
              }
            }
            if (judge(i, count)) lastRLE = i;
          }
          enterSearch = false;
          while (i < nn - 32)
          {
            const u32x4 a = bytes16((uint32_t)i), c = bytes16((uint32_t)i + 16u);
            const uint32_t z0 = a.x ^ c.x, z1 = a.y ^ c.y, z2 = a.z ^ c.z, z3 = a.w ^ c.w;
            if ((z0 | z1 | z2 | z3) == 0u) { symbol = a; i += 32; count = 32; restart = true; break; }
            else if (z3 >> 24) i += 16;
            else
            {
              int32_t hbyte;
              if (z3) hbyte = 12 + ((31 - (int32_t)__builtin_clz(z3)) >> 3);
              else if (z2) hbyte = 8 + ((31 - (int32_t)__builtin_clz(z2)) >> 3);
              else if (z1) hbyte = 4 + ((31 - (int32_t)__builtin_clz(z1)) >> 3);
              else hbyte = (31 - (int32_t)__builtin_clz(z0)) >> 3;
              i += hbyte + 1;
            }
          }
        }
        // scalar step; bytes >= n never match
        symbol = (i + 16 <= nn) ? bytes16((uint32_t)i) : zero4;
        if (i + 32 <= nn && sym_eq4(symbol, bytes16((uint32_t)i + 16u))) { count = 32; i += 32; }
        else { count = 0; i += 1; }
      }
      // final block (:302-497): the pending run, then the terminator (the 128 bit end terminator always carries `00, u32 0`: q11)
      if (judge(i, count))
      {
        term_head(); hb(0); h32(0); hflush();
      }
      else
      {
        const uint32_t kLit = (uint32_t)(i - lastRLE);
        term_head();
        if constexpr (TR::kRange7) h32(((kLit + 1u) << 1) | 1u); else { hb(0); h32(kLit + 1u); }
        hflush();
        emit_literals((uint32_t)lastRLE, kLit);
      }
      if ((opos & 15u) != 0u)
        st128(slot + (opos & ~15u), oacc);
      st32(slot + 4, opos);
      sizes[b] = opos;
      finished = true;
    }

    coop_flush(false);
    wave_sync();
    land();
    wave_sync();
    extend_masks();
  }
  coop_flush(true);
#undef HS_MERGE128
}

} // namespace hsrle
