// hsrle_decode8pe.hip.h -- EXPERIMENT (round 6, -DHSRLE_DEC8_PE builds only): the 8 bit multi-symbol block decoder as PARSE (one lane per block) + EXPAND
// (the wave per output chunk).  VERDICT r5 item 2 (b) / (c) in one kernel.
//
// k_decode_blocks lets every lane copy its own packets' bytes: a trip is as long as the lane with the most 16-byte chunks, half the lanes idle, and the copy loops
// are what a trip mostly is.  Here the per-lane packet walk only PARSES -- header fields, stream / output positions -- and leaves one 32-bit entry per literal
// stretch / run piece of the step's 128 output bytes in LDS; then the wave builds the step's 64 x 8 output chunks, a lane per chunk (8 lanes = one row's 128 bytes,
// stored straight to HBM: no tile), merging the entries that overlap the chunk: literal bytes from the row's ring through the byte funnel, run bytes as the
// broadcast symbol, cut under two byte masks.  The ring is 256 bytes per lane (the literals an entry points at must stay until the expansion; a mid-step top-up
// protects everything from the step's first stream byte on).
//
// Same stream checks, same error bits, same container / top-up code as k_decode_blocks (hsrle_decode.hip.h).  rle8_multi / rle8_packed_multi, mode 0 blocks only.
#pragma once

#include "../hsrle_decode.hip.h"

namespace hsrle {

template <int FAM>
__global__ __launch_bounds__(64) void k_decode8_pe(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ payloadEnd,
                                                   uint8_t *__restrict__ out, uint64_t U, uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t *__restrict__ status,
                                                   const uint32_t *__restrict__ entries, uint32_t entryBase, const uint32_t *__restrict__ gate)
{
  static_assert(FAM == PLAIN || FAM == PACKED, "rle8_multi / rle8_packed_multi");
  using TR = Traits<FAM, 1, 0>;
  constexpr int Q = 128, R = 256, RS = R;
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t RMASK = (uint32_t)R - 1u;
  constexpr uint32_t MAXHDR = 1u + 4u + 1u + 4u + 2u;
  constexpr uint32_t E = 12u;                                           // entries per row and step
  (void)entries; (void)entryBase;
  if (gate != nullptr && (gate[0] | gate[1]) != 0u) return;

  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * RS];
  __shared__ __attribute__((aligned(16))) uint32_t rinfo[64];
  __shared__ __attribute__((aligned(16))) uint32_t ent[64 * E];
  __shared__ __attribute__((aligned(16))) uint32_t rowInfo[64];         // entries | produced bytes << 8
  __shared__ __attribute__((aligned(16))) uint8_t mlut[17 * 16];
  if (threadIdx.x < 17u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (c == 16u) ? ~0u : (uint32_t)part, p1 = (c == 16u) ? ~0u : (uint32_t)(part >> 32);
    lds_st128(mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = firstBlock + xcd_tile(blockIdx.x, gridDim.x) * 64u;
  const uint32_t lastBlockExcl = firstBlock + blockCount;
  const uint32_t b = wgFirst + lane;
  const bool active = b < lastBlockExcl;

  auto rsw_of = [](uint32_t r) -> uint32_t { return (r & 7u) << 4; };
  const uint32_t rowx = (lane * (uint32_t)RS) ^ rsw_of(lane);
  auto ring_win16 = [&](uint32_t p) -> u32x4 {
    const uint32_t a = p & ~7u;
    return funnel24(lds_ld64(ring + (rowx ^ (a & RMASK))), lds_ld64(ring + (rowx ^ ((a + 8u) & RMASK))), lds_ld64(ring + (rowx ^ ((a + 16u) & RMASK))), p & 7u);
  };

  // ---- per-lane stream state (hsrle_decode.hip.h) ----
  uint32_t slen = 0, blen = 0, sp = 0, E_ = 0, lim = 0, lit = 0, run = 0, o = 0, sym4 = 0, err = 0;
  bool last = false, done = true;
  uint32_t g0 = 0;
  uint64_t myBase0 = 0;
  {
    uint64_t base0 = 0;
    if (active)
    {
      uint64_t off0 = offsets[b], off1 = offsets[b + 1];
      const uint64_t payloadBytes = (uint64_t)(payloadEnd - payload);
      if (off0 > off1 || off1 > payloadBytes || off1 - off0 > 0xFFFFFF00ull) { off0 = 0; off1 = 0; }
      g0 = (uint32_t)((uintptr_t)(payload + off0) & (uintptr_t)(Q - 1));
      base0 = off0 - g0;
      slen = (uint32_t)(off1 - off0) + g0;
      const uint64_t start = (uint64_t)b * B;
      blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
      const uint64_t room = (uint64_t)(payloadEnd - payload) - base0;
      lim = (uint32_t)(room > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : room) & ~15u;
      lim = umin(lim, (slen + 15u) & ~15u);
      E_ = g0 & ~15u;
      sp = g0;
      done = false;
    }
    myBase0 = base0;
  }

  // ---- ring top-up (hsrle_decode.hip.h: every chunk requested exactly once; `prot`: stream bytes from here on must stay in the ring) ----
  u32x4 pf[LPR];
  uint32_t pfPos[LPR];
#pragma unroll
  for (int q = 0; q < LPR; q++) { pf[q] = u32x4{ 0, 0, 0, 0 }; pfPos[q] = 0xFFFFFFFFu; }
  uint64_t myBase[LPR];
  uint32_t limq[LPR];
  auto publish = [&](uint32_t v, int P) { rinfo[(lane % (64u / (uint32_t)P)) * (uint32_t)P + lane / (64u / (uint32_t)P)] = v; };
  const uint32_t serveBase = ((lane / LPR) * (uint32_t)RS) ^ rsw_of(lane / LPR);
  auto topup = [&](uint32_t prot) {
    if (!done) E_ += umin(umin((uint32_t)Q, lim - E_), (uint32_t)R - (E_ - (prot & ~15u)));
    uint32_t ri[LPR];
    publish(E_, LPR);
    wave_sync();
#pragma unroll
    for (int q = 0; q < LPR; q++) ri[q] = rinfo[(lane / LPR) * LPR + q];
    wave_sync();
    bool landed[LPR];
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      landed[q] = pfPos[q] < (ri[q] & ~15u);
      if (landed[q]) lds_st128(ring + ((serveBase ^ (pfPos[q] & RMASK)) + (uint32_t)q * RPL * RS), pf[q]);
    }
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      if (landed[q])
      {
        if (limq[q] - pfPos[q] > (uint32_t)Q) { pfPos[q] += (uint32_t)Q; pf[q] = ld128(payload + myBase[q] + pfPos[q]); }
        else pfPos[q] = 0xFFFFFFFFu;
      }
    }
  };
  wave_sync();
#pragma unroll
  for (int q = 0; q < LPR; q++)
  {
    const int r = (int)((uint32_t)q * RPL + lane / LPR);
    const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)myBase0, r, 64), hi32 = (uint32_t)__shfl((int)(uint32_t)(myBase0 >> 32), r, 64);
    myBase[q] = ((uint64_t)hi32 << 32) | lo32;
  }
#pragma unroll
  for (int q = 0; q < LPR; q++)
  {
    const int r = (int)((uint32_t)q * RPL + lane / LPR);
    const uint32_t rowLim = (uint32_t)__shfl((int)lim, r, 64);
    const uint32_t rowE0 = (uint32_t)__shfl((int)E_, r, 64);
    limq[q] = rowLim;
    pfPos[q] = (lane % LPR) * 16u;
    if (pfPos[q] < rowE0) pfPos[q] += (uint32_t)Q;
    if (pfPos[q] < rowLim) pf[q] = ld128(payload + myBase[q] + pfPos[q]);
    else pfPos[q] = 0xFFFFFFFFu;
  }
  for (int k = 0; k < R / Q; k++) { topup(sp); wave_sync(); }

  if (active)
  {
    sp = g0 + TR::kHeaderSize;
    const u32x4 hdv = ring_win16(g0);
    const uint32_t hd8 = hdv.z & 0xFFu;
    if (slen < g0 + TR::kHeaderSize + 2u || hdv.x != blen || hdv.y != slen - g0) { err |= DEC_ERR_HEADER; done = true; }
    else if (hd8 != 0u) { err |= DEC_ERR_MODE; done = true; }
  }

  uint32_t base = 0;
  uint32_t roundsLeft = B / 128u + B / 16u + 64u;
  const uint32_t rowLenAll = blen;                                      // (published per step below)

  while (__ballot(!done && o < blen) != 0ull)
  {
    if (roundsLeft-- == 0u) { err |= DEC_ERR_STREAM; break; }
    const uint32_t target = umin(base + 128u, blen);
    const uint32_t spStart = sp;                                        // entries of this step point at ring bytes from here on
    uint32_t nEnt = 0;
    constexpr uint32_t F_DONE = 1u, F_STALL = 2u, F_LAST = 4u;
    uint32_t fl = (done ? F_DONE : 0u) | (last ? F_LAST : 0u);

    for (int pass = 0; pass < 3; pass++)
    {
      const uint32_t avail0 = E_;
      const uint32_t spOK = umin((avail0 < lim) ? avail0 - MAXHDR : 0xFFFFFFFFu, slen - 2u);
      fl &= ~F_STALL;
      // ================= PARSE: one packet per trip, no data bytes touched =================
      for (;;)
      {
        const bool act = (fl & (F_DONE | F_STALL)) == 0u && o < target;
        if (__ballot(act) == 0ull) break;
        if (act && (lit | run) == 0u)
        {
          if ((fl & F_LAST) != 0u || sp > spOK)
          {
            if ((fl & F_LAST) != 0u) fl |= F_DONE;
            else if (sp + 2u > slen) { err |= DEC_ERR_STREAM; fl |= F_DONE; }
            else fl |= F_STALL;
          }
          else
          {
            const u32x4 hv = ring_win16(sp);
            [[maybe_unused]] const uint64_t lo = (uint64_t)hv.x | ((uint64_t)hv.y << 32), hi = (uint64_t)hv.z | ((uint64_t)hv.w << 32);
            uint32_t cnt, pos, nsym, range, used, endNow = 0;
            const uint32_t b0 = hv.x & 0xFFu;
            if constexpr (!TR::kPacked)
            {
              cnt = (hv.x >> 8) & 0xFFu;
              pos = 2u;
              nsym = __builtin_amdgcn_perm(hv.x, hv.x, 0u);
              const uint32_t c32 = ex32(lo, hi, pos);
              const bool longc = cnt == 0u;
              cnt = longc ? c32 : cnt;
              pos += longc ? 4u : 0u;
              const uint32_t w = ex32(lo, hi, pos);
              const uint32_t r0 = w & 0xFFu;
              const bool longr = r0 == 0u;
              const uint32_t r32 = ex32(lo, hi, pos + 1u);
              range = longr ? r32 : r0;
              used = pos + (longr ? 5u : 1u);
              endNow = (longr && range == 0u) ? 1u : 0u;
            }
            else
            {
              const uint32_t c7 = b0 & 0x7Fu;
              const bool longc = c7 == 0u;
              cnt = longc ? alignbyte(hv.y, hv.x, 1u) : c7;
              const uint32_t A = longc ? hv.y : hv.x, Bd = longc ? hv.z : hv.y;
              const bool newSym = !(b0 & 0x80u);
              nsym = newSym ? __builtin_amdgcn_perm(A, A, 0x01010101u) : sym4;
              const uint32_t posr = newSym ? 2u : 1u;
              pos = (longc ? 4u : 0u) + posr;
              const uint32_t w = alignbyte(Bd, A, posr);
              const uint32_t r0 = w & 0xFFu;
              const bool longr = (r0 & 1u) != 0u;
              range = longr ? (w >> 1) : (r0 >> 1);
              used = pos + (longr ? 4u : 1u);
              endNow = (longr && range == 0u) ? 1u : 0u;
            }
            const bool lastNow = endNow || cnt == 0u;
            sym4 = nsym;
            lit = (range == 0u || endNow) ? 0u : range - 1u;
            run = lastNow ? 0u : cnt + TR::SHORT - 1u;
            if (lastNow) fl |= F_LAST;
            sp += used;
            if (sp > slen || lit > slen - sp || (lit == 0u && run == 0u && !lastNow)) { err |= DEC_ERR_STREAM; fl |= F_DONE; }
          }
        }
        if (act && (fl & (F_DONE | F_STALL)) == 0u)
        {
          if (lit != 0u)
          {
            const uint32_t resident = (avail0 > sp) ? avail0 - sp : 0u;
            const uint32_t n = umin(umin(lit, target - o), resident);
            if (n == 0u) fl |= F_STALL;
            else
            {
              if (nEnt < E) ent[lane * E + nEnt] = (o - base) | ((n - 1u) << 7) | ((sp & RMASK) << 16);
              else err |= 0x100u;                                       // (prototype: a step with more pieces than the list holds)
              nEnt++;
              sp += n; lit -= n; o += n;
            }
          }
          if (lit == 0u && run != 0u && o < target)
          {
            const uint32_t m = umin(run, target - o);
            if (nEnt < E) ent[lane * E + nEnt] = (o - base) | ((m - 1u) << 7) | (1u << 14) | ((sym4 & 0xFFu) << 16);
            else err |= 0x100u;
            nEnt++;
            run -= m; o += m;
          }
        }
      }
      wave_sync();
      if (__ballot((fl & F_DONE) == 0u && o < target) == 0ull) break;     // every row has its step (the common case: one pass)
      topup(spStart);                                                   // starved rows: more stream bytes, nothing of this step overwritten
      wave_sync();
    }
    done = (fl & F_DONE) != 0u;
    last = (fl & F_LAST) != 0u;

    // ================= EXPAND: a lane per 16-byte output chunk, 8 rows per trip =================
    rowInfo[lane] = (nEnt < E ? nEnt : E) | (((active && o > base) ? o - base : 0u) << 8);
    wave_sync();
#pragma unroll 2
    for (uint32_t it = 0; it < 8u; it++)
    {
      const uint32_t r = it * 8u + (lane >> 3), c16 = (lane & 7u) << 4;
      const uint32_t ri = rowInfo[r];
      const uint32_t n = ri & 0xFFu, prod = ri >> 8;
      if (c16 < prod)
      {
        const uint32_t rrowx = (r * (uint32_t)RS) ^ rsw_of(r);
        u32x4 v = u32x4{ 0, 0, 0, 0 };
        for (uint32_t j = 0; j < n; j++)
        {
          const uint32_t e = ent[r * E + j];
          const uint32_t a = e & 127u, bnd = a + ((e >> 7) & 127u) + 1u;
          if (bnd <= c16) continue;
          if (a >= c16 + 16u) break;
          const uint32_t lo = a > c16 ? a - c16 : 0u, hi = bnd < c16 + 16u ? bnd - c16 : 16u;
          const u32x4 mh = lds_ld128(mlut + (hi << 4)), ml = lds_ld128(mlut + (lo << 4));
          u32x4 piece;
          if ((e >> 14) & 1u)
          {
            const uint32_t s4 = ((e >> 16) & 0xFFu) * 0x01010101u;
            piece = u32x4{ s4, s4, s4, s4 };
          }
          else
          {
            const uint32_t p = ((e >> 16) & RMASK) + c16 + (uint32_t)R - a;  // ring position of the byte that lands at chunk byte 0 (mod R)
            const uint32_t p8 = p & ~7u;
            piece = funnel24(lds_ld64(ring + (rrowx ^ (p8 & RMASK))), lds_ld64(ring + (rrowx ^ ((p8 + 8u) & RMASK))), lds_ld64(ring + (rrowx ^ ((p8 + 16u) & RMASK))), p & 7u);
          }
          v.x |= piece.x & mh.x & ~ml.x; v.y |= piece.y & mh.y & ~ml.y; v.z |= piece.z & mh.z & ~ml.z; v.w |= piece.w & mh.w & ~ml.w;
        }
        uint8_t *const dst = out + (uint64_t)(wgFirst + r) * B + base + c16;
        if (prod - c16 >= 16u) __builtin_nontemporal_store(v, (u32x4_unaligned *)dst);
        else
        {
          // (the last block of a buffer may end inside a chunk)
          const uint64_t w0 = (uint64_t)v.x | ((uint64_t)v.y << 32), w1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
          for (uint32_t k = 0; k < prod - c16; k++) dst[k] = (uint8_t)(k < 8u ? w0 >> (8u * k) : w1 >> (8u * (k - 8u)));
        }
      }
    }
    wave_sync();
    base += 128u;
    topup(sp);
    wave_sync();
  }

  (void)rowLenAll;
  if (active && o != blen) err |= DEC_ERR_STREAM;
  if (err != 0 && status != nullptr) atomicOr(status, err);
}

} // namespace hsrle
