// hsrle_encode8.hip.h -- the 8 bit multi-symbol encoders (rle8_multi, rle8_packed_multi, rle8_{3,7}symlut) with the block's
// input and output staged through LDS.
//
// Replaces: src/rle8_extreme_cpu.h:86-344 (wrapper, scalar tail, final block), :936-1099 (canonical AVX2 body),
//           src/rleX_Xsl.h:114-264 (process_symbol), :269-346, :421-485 (TYPE_SIZE 8 instantiation).
//
// Same execution model as everything else here: one lane = one block = one reference stream, 64 blocks per wavefront, the
// emit decisions run as the sequential state machine they are.  What this kernel adds over the generic
// k_encode_blocks is the DATA PATH (the generic kernel reads and writes global memory per lane, which costs 6-7x HBM traffic
// amplification, measured):
//   HBM --(top-up: 4 adjacent lanes read 64 contiguous, aligned input bytes of ONE block per step)--> LDS history ring [64][256]
//   ring --(run detection: 16 positions per step from one aligned 16-byte read: x ^ (x >> 8) == 0, the GPU form of the
//           reference's cmpeq + movemask + ctz scan, rle8_extreme_cpu.h:952-1084; events = run starts / run ends)-->
//   packets (header bytes + literal bytes copied ring -> row with a 128-bit byte funnel) --> LDS output row [64][160]
//   row --(flush: 4 adjacent lanes write whole 16-byte chunks of one row)--> HBM staging slot
// Literal gaps longer than the ring can serve (incompressible stretches) take a direct global-to-global path.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low, wave_sync

namespace hsrle {

template <int FAM>
__global__ __launch_bounds__(64) void k_encode8_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                       uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes)
{
  using TR = Traits<FAM, 1, 0>;
  constexpr int Q = 64;                      // input bytes per lane and step
  constexpr int H = 256;                     // history ring per lane (power of two)
  constexpr int HS = H + 48;                 // ring row stride (32 mirror bytes; odd multiple of 16)
  constexpr int OC = 160;                    // output row capacity
  constexpr int OS = OC + 16;                // output row stride (16 bytes of over-write slack; odd multiple of 16)
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr uint32_t LMAX = 96u;             // longest literal gap served from the ring
  constexpr uint32_t HDRMAX = 12u;           // longest packet header + slack
  constexpr int K = TR::K;

  __shared__ __attribute__((aligned(16))) uint8_t hist[64 * HS];
  __shared__ __attribute__((aligned(16))) uint8_t outr[64 * OS];

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = blockIdx.x * 64u;
  const uint32_t b = wgFirst + lane;
  const bool active = b < nBlocks;

  uint8_t *const hrow = hist + lane * HS;
  uint8_t *const orow = outr + lane * OS;

  uint32_t n = 0;
  if (active)
  {
    const uint64_t start = (uint64_t)b * B;
    n = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  }
  const uint8_t *const d = in + (uint64_t)b * B;
  uint8_t *const slot = slots + (uint64_t)b * slotStride;
  const uint8_t *const inEnd = in + U;

  // ---- per-lane encoder state ----
  uint32_t avail = 0;        // input bytes [.., avail) are (or were) in the ring; the ring holds [avail - H, avail)
  uint32_t cb = 0;           // base of the 16-position chunk scanned next (multiple of 16)
  bool inRun = false;
  uint32_t runStart = 0, sym = 0;
  uint32_t lastRLE = 0;
  uint32_t lastSym = 0;      // Packed: lastSymbol (starts 0, A.5 q5)
  [[maybe_unused]] uint64_t lutw = (K == 3) ? 0x0000000000FF7F00ull : 0x00FE807E01FF7F00ull; // LUT: MTF list, entry k in byte k
  bool ended = false;        // the end terminator has been written
  bool finished = !active;   // the whole stream is in the row / slot
  uint32_t pend = 0;         // bytes in the output row
  uint32_t rowBase = 0;      // stream offset of row byte 0 (multiple of 16)
  bool reported = false;     // the stream size has been written
  uint64_t winStarts = 0, pendingEnds = 0;   // current window: run-start bits, run-end bits not yet handled
  bool windowOpen = false, winOpenAtEnd = false;
  uint32_t winW = 0;

  // ---- output row primitives ----
  auto ob = [&](uint32_t v) { orow[pend] = (uint8_t)v; pend += 1; };
  auto o16 = [&](uint32_t v) { ob(v); ob(v >> 8); };
  auto o32 = [&](uint32_t v) { ob(v); ob(v >> 8); ob(v >> 16); ob(v >> 24); };

  // literal bytes [from, from + len) of the input, served from the ring
  auto lit_from_ring = [&](uint32_t from, uint32_t len) {
    if (len == 0) return;
    const uint32_t c = pend & 15u;
    uint8_t *dst = orow + (pend & ~15u);
    const uint32_t srcp = from - c;
    const uint32_t sh = srcp & 15u, a0 = srcp & ~15u;
    u32x4 x = lds_ld128(hrow + (a0 & HM)), y = lds_ld128(hrow + ((a0 + 16u) & HM));
    u32x4 w = merge_low(lds_ld128(dst), funnel16(x, y, sh), c);
    lds_st128(dst, w);
    const uint32_t total = c + len;
    for (uint32_t k = 16; k < total; k += 16)
    {
      x = y;
      y = lds_ld128(hrow + ((a0 + k + 16u) & HM));
      lds_st128(dst + k, funnel16(x, y, sh));
    }
    pend += len;
  };

  // long literal gap: everything for this packet goes straight to the slot in global memory
  auto direct_flush_row = [&]() {
    for (uint32_t k = 0; k < pend; k += 16)
      st128(slot + rowBase + k, lds_ld128(orow + k));
  };
  auto direct_literals = [&](uint32_t gpos, uint32_t from, uint32_t len) {
    copy_over(slot + gpos, d + from, len, inEnd);
  };
  // After a direct packet the row continues at stream offset `end`; its first chunk must hold the (end & 15) stream bytes
  // that precede it.  Those are the tail of [old row bytes .. header | literals]: bytes below `litAt` are still in the old
  // row (which started at oldBase), the others are input bytes starting at `from`.
  auto restart_row = [&](uint32_t end, uint32_t oldBase, uint32_t litAt, uint32_t from) {
    const uint32_t nb = end & ~15u, np = end & 15u;
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (uint32_t k = 0; k < np; k++)
    {
      const uint32_t sPos = nb + k;                                    // stream offset of this byte
      const uint32_t v = (sPos < litAt) ? (uint32_t)orow[sPos - oldBase] : (uint32_t)d[from + (sPos - litAt)];
      const uint32_t sh = 8u * (k & 3u);
      if (k < 4u) t0 |= v << sh; else if (k < 8u) t1 |= v << sh; else if (k < 12u) t2 |= v << sh; else t3 |= v << sh;
    }
    lds_st128(orow, u32x4{ t0, t1, t2, t3 });
    rowBase = nb;
    pend = np;
  };

  // ---- stream header ----
  if (active)
  {
    o32(n);
    o32(0);
    if constexpr (!TR::kLut) ob(0); // mode = multi
  }

  // ---- input top-up (4 lanes per row read 64 contiguous bytes; the loads fly during the step's scan) ----
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0;

  auto issue = [&]() {
    const uint32_t left = (n > avail) ? (n - avail + 15u) >> 4 : 0u;
    wantReq = umin((uint32_t)LPR, left);
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = lane % LPR;
      const uint32_t e = (uint32_t)__shfl((int)avail, (int)r, 64), nreq = (uint32_t)__shfl((int)wantReq, (int)r, 64);
      const bool valid = c < nreq;
      const uint64_t g = (uint64_t)(wgFirst + r) * B + e + c * 16u;
      u32x4 v = u32x4{ 0, 0, 0, 0 };
      if (valid)
      {
        if (g + 16u <= U)
          v = ld128(in + g);
        else
        {
          uint32_t t[4] = { 0, 0, 0, 0 };
          for (uint32_t k = 0; k < 16u && g + k < U; k++)
            t[k >> 2] |= (uint32_t)in[g + k] << (8u * (k & 3u));
          v = u32x4{ t[0], t[1], t[2], t[3] };
        }
      }
      pf[q] = v;
      pfAt[q] = r * (uint32_t)HS + ((e + c * 16u) & HM);
    }
  };

  auto land = [&]() {
    // the ring must keep the chunk being scanned (and one before it for the byte funnel)
    const uint32_t keep = (cb >= 16u) ? cb - 16u : 0u;
    const uint32_t fit = ((uint32_t)H - (avail - keep)) >> 4;
    const uint32_t take = umin(wantReq, fit);
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = lane % LPR;
      const uint32_t tk = (uint32_t)__shfl((int)take, (int)r, 64);
      if (c < tk)
      {
        lds_st128(hist + pfAt[q], pf[q]);
        if ((pfAt[q] - r * (uint32_t)HS) < 32u)
          lds_st128(hist + pfAt[q] + (uint32_t)H, pf[q]); // mirror of the first 32 ring bytes
      }
    }
    avail = umin(avail + (take << 4), n);
  };

  // ---- one finished run [p, e): decide, and if emitted write the packet.  Returns false if the row is full (retry later) ----
  auto handle_run = [&](uint32_t p, uint32_t e) -> bool {
    const uint32_t count = e - p;
    const uint32_t gap = p - lastRLE;
    bool same = false, body = false;
    int k;                                                             // 0 keep as literals, 1 short range field, 2 long range field
    [[maybe_unused]] uint32_t m = 0, c7 = 0, r7 = 0, cst = 0, rng = 0;

    if constexpr (TR::kLut)
    {
      // rleX_Xsl.h:116-132
      rng = gap + 2u;
      m = (uint32_t)K;
#pragma unroll
      for (int j = K - 1; j >= 0; j--)
        if (((lutw >> (8 * j)) & 0xFFull) == (uint64_t)sym) m = (uint32_t)j;
      cst = count - 3u + 2u;
      constexpr uint32_t MAXC = 127u, MAXR = (1u << TR::RB) - 1u;
      uint32_t pen = (rng <= 0xFFFFFu) ? (rng <= MAXR ? 0u : 2u) : 4u;  // 0xFFFFF vs 0xFFFF in the writer: A.5 q3
      pen += (cst <= 0xFFFFFu) ? (cst <= MAXC ? 0u : 2u) : 4u;
      pen += (m == (uint32_t)K) ? 1u : 0u;
      k = (count >= 1u + 10u || count >= 3u + pen) ? 1 : 0;
      c7 = (cst <= MAXC) ? cst : (cst <= 0xFFFFu ? 1u : 0u);
      r7 = (rng <= MAXR) ? rng : (rng <= 0xFFFFu ? 1u : 0u);
    }
    else if constexpr (TR::kPacked)
    {
      // body / tail split of the canonical AVX2 encoder
      rng = gap + 1u;
      const int32_t kk = (int32_t)(count - 1u) / 32;
      body = (e < n) && ((int32_t)p + 1 + 32 * kk < (int32_t)n - 32);
      if (body)
      {
        same = sym == lastSym;
        const bool emit = count >= 11u || (rng <= 127u && ((same && count >= 3u) || count >= 4u));
        k = emit ? (rng <= 127u ? 1 : 2) : 0;
      }
      else
        k = (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
    }
    else
    {
      rng = gap + 1u;
      k = (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;                   // rle8_extreme_cpu.h:974
    }

    if (!k)
      return true;

    const bool viaRing = gap <= LMAX && (gap == 0u || lastRLE + (uint32_t)H >= avail + 16u);
    if (pend + HDRMAX + (viaRing ? gap + 16u : 0u) > (uint32_t)OC)
      return false;                                                    // row full: flush first, then come back to this run end

    if (!viaRing)
      direct_flush_row();
    const uint32_t pend0 = pend;

    // ---- header ----
    if constexpr (TR::kLut)
    {
      const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
      const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
      const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
      lutw = keepHi | (low << 8) | (uint64_t)sym;
      o16((m << (K == 3 ? 14 : 13)) | (c7 << TR::RB) | r7);
      if (m == (uint32_t)K) ob(sym);
      if (cst != c7) { if (cst <= 0xFFFFu) o16(cst); else o32(cst); }
      if (rng != r7) { if (rng <= 0xFFFFu) o16(rng); else o32(rng); }
    }
    else if constexpr (TR::kPacked)
    {
      if (body) lastSym = sym;                                          // only the body rule tracks lastSymbol (A.3)
      const uint32_t c = count - 3u + 1u, sm = same ? 0x80u : 0u;
      if (c <= 127u) ob(c | sm); else { ob(sm); o32(c); }
      if (!same) ob(sym);
      if (k == 1) ob((rng << 1) & 0xFFu); else o32((rng << 1) | 1u);
    }
    else
    {
      const uint32_t c = count - 6u + 1u;
      ob(sym);
      if (c <= 255u) ob(c); else { ob(0); o32(c); }
      if (k == 1) ob(rng); else { ob(0); o32(rng); }
    }

    // ---- literals ----
    if (viaRing)
      lit_from_ring(lastRLE, gap);
    else
    {
      // header bytes were appended to the row behind the directly flushed part: move them out, then the literals
      const uint32_t hdr = pend - pend0;
      for (uint32_t j = 0; j < hdr; j++)
        slot[rowBase + pend0 + j] = orow[pend0 + j];
      direct_literals(rowBase + pend, lastRLE, gap);
      restart_row(rowBase + pend + gap, rowBase, rowBase + pend, lastRLE);
    }

    lastRLE = e;

    if (e >= n)
    {
      // end terminator (rle8_extreme_cpu.h:203-338; rleX_Xsl.h:319-338)
      if constexpr (TR::kLut) { o16((1u << TR::RB) | 1u); o16(0); o16(0); }
      else if constexpr (TR::kPacked) { ob(0x80); o32(0); o32(1); }
      else { ob(0); ob(0); o32(0); ob(0); o32(0); }
      ended = true;
    }
    return true;
  };

  // literal terminator carrying the bytes behind the last emitted run
  auto finish_literals = [&]() -> bool {
    const uint32_t kLit = n - lastRLE;
    const bool viaRing = kLit <= LMAX && (kLit == 0u || lastRLE + (uint32_t)H >= avail + 16u);
    if (pend + HDRMAX + (viaRing ? kLit + 16u : 0u) > (uint32_t)OC)
      return false;
    if (!viaRing)
      direct_flush_row();
    const uint32_t pend0 = pend;
    if constexpr (TR::kLut) { o16(1u << TR::RB); o16(0); o32(kLit + 2u); }
    else if constexpr (TR::kPacked) { ob(0x80); o32(0); o32(((kLit + 1u) << 1) | 1u); }
    else { ob(0); ob(0); o32(0); ob(0); o32(kLit + 1u); }
    if (viaRing)
      lit_from_ring(lastRLE, kLit);
    else
    {
      const uint32_t hdr = pend - pend0;
      for (uint32_t j = 0; j < hdr; j++)
        slot[rowBase + pend0 + j] = orow[pend0 + j];
      direct_literals(rowBase + pend, lastRLE, kLit);
      restart_row(rowBase + pend + kLit, rowBase, rowBase + pend, lastRLE);
    }
    return true;
  };

  // ---- main loop ----
  issue();
  land();
  wave_sync();

  uint32_t stepsLeft = 4u * (B / (uint32_t)Q) + B / 8u + 64u;           // bounded: every step scans or flushes something

  while (__ballot(!finished || (active && !reported)) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    issue();

    // ---------------- scan what is in the ring ----------------
    // Phase A (uniform): equality mask of up to 64 positions -> bit masks of run starts and run ends.
    // Phase B (per lane): one handle_run per run END -- a lane's trip count is the number of runs that end in its window.
    if (!finished)
    {
      if (pendingEnds == 0ull && !windowOpen)
      {
        // window [cb, cb + W): every position needs its successor byte (or the end of the input)
        const uint32_t lastStep = (avail >= n) ? 1u : 0u;
        uint32_t W = lastStep ? umin(64u, n - cb) : umin(64u, ((avail - 1u - cb) >> 4) << 4);
        if (cb >= n) W = 0;
        if (W != 0u)
        {
          uint64_t e64 = 0;
#pragma unroll
          for (uint32_t j = 0; j < 4u; j++)
          {
            if (j * 16u < W)
            {
              const u32x4 x = lds_ld128(hrow + ((cb + j * 16u) & HM));
              const uint32_t x4 = lds_ld32(hrow + ((cb + j * 16u + 16u) & HM));
              const uint32_t z0 = zero_bytes(x.x ^ alignbyte(x.y, x.x, 1)), z1 = zero_bytes(x.y ^ alignbyte(x.z, x.y, 1));
              const uint32_t z2 = zero_bytes(x.z ^ alignbyte(x.w, x.z, 1)), z3 = zero_bytes(x.w ^ alignbyte(x4, x.w, 1));
              // 0x80 flags -> 4 bits per dword
              const uint32_t b0 = (((z0 >> 7) * 0x00204081u) >> 21) & 0xFu, b1 = (((z1 >> 7) * 0x00204081u) >> 21) & 0xFu;
              const uint32_t b2 = (((z2 >> 7) * 0x00204081u) >> 21) & 0xFu, b3 = (((z3 >> 7) * 0x00204081u) >> 21) & 0xFu;
              e64 |= (uint64_t)(b0 | (b1 << 4) | (b2 << 8) | (b3 << 12)) << (16u * j);
            }
          }
          // position i is a match only if cb + i + 1 < n (bytes at or beyond n never match)
          const uint32_t validBits = (n - cb > W) ? W : (n - cb - 1u);
          e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
          const uint64_t wmask = (W >= 64u) ? ~0ull : ((1ull << W) - 1ull);
          const uint64_t prev = (e64 << 1) | (inRun ? 1ull : 0ull);      // "the position before me matched"
          winStarts = e64 & ~prev;
          pendingEnds = ~e64 & prev & wmask;                            // bit i: a run ends with position i (exclusive end cb + i + 1)
          winOpenAtEnd = ((e64 >> (W - 1u)) & 1ull) != 0ull;             // the last position matches its successor: run goes on
          winW = W;
          windowOpen = true;
        }
      }

      bool stall = false;

      while (pendingEnds != 0ull)
      {
        const uint32_t i = (uint32_t)__builtin_ctzll(pendingEnds);
        const uint64_t sBelow = winStarts & ((2ull << i) - 1ull);
        uint32_t st = runStart, sy = sym;
        if (sBelow != 0ull)
        {
          st = cb + (63u - (uint32_t)__builtin_clzll(sBelow));
          sy = hrow[st & HM];
        }
        sym = sy;
        if (!handle_run(st, cb + i + 1u)) { stall = true; break; }
        pendingEnds &= pendingEnds - 1ull;
      }

      if (!stall && windowOpen)
      {
        // the window is done: remember a run that is still open at its end
        if (winOpenAtEnd)
        {
          // the open run is the last one that started in this window; with no start at all the carried run goes on
          if (winStarts != 0ull)
          {
            runStart = cb + (63u - (uint32_t)__builtin_clzll(winStarts));
            sym = hrow[runStart & HM];
          }
          inRun = true;
        }
        else
          inRun = false;
        cb += winW;
        windowOpen = false;
      }

      if (!stall && !windowOpen && cb >= n && avail >= n)
      {
        // end of input: a run that reaches the end is judged now; then the literal terminator unless the stream ended
        bool ok = true;
        if (inRun) { ok = handle_run(runStart, n); if (ok) inRun = false; }
        if (ok && !ended) { ok = finish_literals(); if (ok) ended = true; }
        if (ok) finished = true;
      }
    }

    wave_sync();
    land();
    wave_sync();

    // ---------------- flush whole 16-byte chunks of every row (4 lanes per row; rows keep their ragged tail) ----------------
    {
      const uint32_t nch = reported ? 0u : (finished ? (pend + 15u) >> 4 : pend >> 4); // a finished stream flushes its last partial chunk too
      for (uint32_t pass = 0; pass < (uint32_t)(OS / 16 + LPR - 1) / LPR; pass++)
      {
        if (__ballot(nch > pass * LPR) == 0ull) break;
#pragma unroll
        for (int q = 0; q < LPR; q++)
        {
          const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = pass * LPR + lane % LPR;
          const uint32_t rn = (uint32_t)__shfl((int)nch, (int)r, 64);
          const uint32_t rb = (uint32_t)__shfl((int)rowBase, (int)r, 64);
          if (c < rn)
            st128(slots + (uint64_t)(wgFirst + r) * slotStride + rb + c * 16u, lds_ld128(outr + r * OS + c * 16u));
        }
      }
      wave_sync();
      if (!finished)
      {
        const uint32_t whole = pend & ~15u;
        if (whole != 0u)
          lds_st128(orow, lds_ld128(orow + whole));                    // the ragged tail moves to the row start
        rowBase += whole;
        pend &= 15u;
      }
      else if (active && !reported)
      {
        const uint32_t total = rowBase + pend;
        st32(slot + 4, total);                                         // compressedLength (patched behind the flushed header)
        sizes[b] = total;
        reported = true;
      }
      wave_sync();
    }
  }
}

} // namespace hsrle
