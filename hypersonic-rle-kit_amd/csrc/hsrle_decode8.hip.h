// hsrle_decode8.hip.h -- the 8 bit decoders (rle8_[packed_]decompress incl. the Single modes, rle8_{3,7}symlut_decompress)
// as a branch-free per-lane step machine.  This is the north-star kernel (8 bit Packed decode).
//
// Replaces: src/rle8_extreme_cpu.h:702-764 (dispatch), :1546-2006 (multi bodies), :2008-2434 (single bodies),
//           src/rleX_Xsl.h:530-1881 (TYPE_SIZE 8 instantiation), and the MEMCPY_*/MEMSET_* macros they expand
//           (src/rleX_extreme_common.h:32-312).
//
// One lane = one block = one reference stream (see hsrle_decode.hip.h for the general picture).  What is specific here:
//
//  * STEP MACHINE.  Every loop iteration ("step") is the same straight-line code for all 64 lanes: ONE unaligned 16-byte
//    LDS read at the lane's stream position, which is either a packet header (then the fields are picked out of the
//    register with shifts, and the bytes behind the header are already the first literal bytes) or the next 16 literal
//    bytes; then ONE 16-byte LDS store of either literal bytes or the byte-broadcast run symbol into the lane's tile row
//    (stores over-write up to 15 bytes, the next step repairs them -- the trick of the reference's MEMCPY/MEMSET macros).
//    No data-dependent branches: lanes that parse, copy and fill all run the same instructions with selects, so the
//    wavefront never serialises over "which phase is this lane in".
//  * COALESCED STREAM TOP-UP.  Streams enter the per-lane LDS rings through loads in which 8 adjacent lanes read 128
//    contiguous bytes of ONE stream (per-lane 16-byte loads at 64 different streams run at ~1 TB/s chip-wide on MI355X,
//    measured; 8-lane groups read whole lines).  The loads are issued one round ahead of their use.
//  * WHOLE-LINE FLUSH.  The tile [64][T] is written out by 8 adjacent lanes per row: every store instruction covers
//    whole 128-byte lines of the output.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h"

namespace hsrle {

// 32 bits at byte offset pos (0..12) of the 16-byte little-endian value hi:lo
__device__ __forceinline__ uint32_t ex32(uint64_t lo, uint64_t hi, uint32_t pos)
{
  const uint32_t sh = pos * 8u;
  const uint64_t a = (lo >> (sh & 63u)) | ((hi << 1) << (63u - (sh & 63u)));
  const uint64_t b = hi >> (sh & 63u);
  return (uint32_t)(sh < 64u ? a : b);
}

template <int FAM, int T, int R>
__global__ __launch_bounds__(64) void k_decode8_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets,
                                                       const uint8_t *__restrict__ payloadEnd, uint8_t *__restrict__ out, uint64_t U,
                                                       uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, 1, 0>;
  constexpr int TS = T + 16;                 // tile row stride: 16 bytes of over-write slack
  constexpr int RS = R + 16;                 // ring row stride: bytes [R, R+16) mirror [0, 16) so 16-byte reads never wrap
  constexpr int CPR = T / 16;                // 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR;              // tile rows covered by one flush instruction
  constexpr uint32_t RMASK = (uint32_t)R - 1u;
  constexpr uint32_t MAXHDR = 12u;           // longest 8 bit packet header (11) + 1
  constexpr uint32_t SHORT_MULTI = TR::kLut ? 3u : (TR::kPacked ? 3u : 6u);
  constexpr uint32_t SHORT_SINGLE = TR::kPacked ? 2u : 4u;
  static_assert((R & (R - 1)) == 0 && R >= 128, "ring size must be a power of two");
  static_assert(T == 64 || T == 128, "tile rows are flushed as whole 64/128-byte pieces");

  __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TS];
  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * RS + 64 * 16]; // + 64 dump slots where predicated-off stores go
  __shared__ uint64_t rowBase[64];                                  // stream start of every row, relative to `payload`
  __shared__ uint64_t rowReq[64];                                   // per round: (chunks to load << 32) | loaded end
  __shared__ uint32_t rowStart[64], rowLen[64];                     // slow flush path only

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = firstBlock + blockIdx.x * 64u;
  const uint32_t lastBlockExcl = firstBlock + blockCount;
  const uint32_t b = wgFirst + lane;
  const bool active = b < lastBlockExcl;

  uint8_t *const row = tile + lane * TS;
  uint8_t *const rng = ring + lane * RS;

  // ---- per-lane stream state ----
  uint32_t slen = 0, blen = 0;
  uint32_t sp = 0;        // read position in the stream
  uint32_t E = 0;         // stream bytes [.., E) are in the ring (multiple of 16)
  uint32_t lim = 0;       // E never exceeds lim (loadable bytes of this stream incl. the payload tail pad)
  uint32_t lit = 0;       // literal bytes of the current packet still to copy
  uint32_t run = 0;       // run bytes of the current packet still to write
  uint32_t o = 0;         // bytes of this block produced so far
  uint32_t sym4 = 0;      // current symbol, byte-broadcast
  uint32_t last = 0;      // the stream ends after the current packet's literals
  uint32_t done = 1;
  uint32_t single = 0;
  uint32_t err = 0;
  [[maybe_unused]] uint64_t lutw = 0; // LUT: move-to-front list, entry k in byte k

  {
    uint64_t off0 = 0;
    if (active)
    {
      off0 = offsets[b];
      slen = (uint32_t)(offsets[b + 1] - off0);
      const uint64_t start = (uint64_t)b * B;
      blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
      const uint64_t room = (uint64_t)(payloadEnd - payload) - off0;
      lim = (uint32_t)(room > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : room) & ~15u;
      done = 0;
    }
    rowBase[lane] = off0;
  }

  // ---- ring top-up.  issue(): 8 loads, in load q lanes 8g..8g+7 read 128 contiguous bytes of row 8q+g's stream ----
  u32x4 pf[8];
  uint32_t pfAt[8], pfMirror[8];

  auto issue = [&]() {
    const uint32_t resident = E - (sp & ~15u);
    uint32_t want = umin(umin(((uint32_t)R - resident) >> 4, 8u), (lim - E) >> 4);
    if (done) want = 0;
    rowReq[lane] = ((uint64_t)want << 32) | E;
    E += want << 4;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
      const uint32_t r = (uint32_t)q * 8u + (lane >> 3), c = lane & 7u;
      const uint64_t req = rowReq[r];
      const uint32_t e = (uint32_t)req, n = (uint32_t)(req >> 32);
      const bool valid = c < n;
      const uint32_t pos = e + c * 16u;
      const uint32_t ro = pos & RMASK;
      pf[q] = ld128(payload + rowBase[r] + (valid ? pos : 0u));        // predicated-off lanes re-read the stream start
      pfAt[q] = valid ? (r * (uint32_t)RS + ro) : (64u * (uint32_t)RS + lane * 16u);          // ring slot or dump slot
      pfMirror[q] = (valid && ro == 0u) ? (r * (uint32_t)RS + (uint32_t)R) : (64u * (uint32_t)RS + lane * 16u);
    }
  };

  // the dump slots sit behind the ring rows in the same array, so a plain offset selects either
  auto land = [&]() {
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
      st128(ring + pfAt[q], pf[q]);
      st128(ring + pfMirror[q], pf[q]);
    }
  };

  // prologue: fill the ring (R / 128 top-ups), then read the stream header from it
  __syncthreads();
  for (int k = 0; k < R / 128; k++)
  {
    issue();
    land();
    __syncthreads();
  }

  if (active)
  {
    sp = TR::kHeaderSize;

    if (slen < TR::kHeaderSize + 2u || ld32(rng) != blen || ld32(rng + 4) != slen)
    {
      err |= DEC_ERR_HEADER;
      done = 1;
    }
    else if constexpr (!TR::kLut)
    {
      const uint32_t mode = rng[8];
      if (mode == 1u) { single = 1; sym4 = (uint32_t)rng[9] * 0x01010101u; sp = 10; }
      else if (mode != 0u) { err |= DEC_ERR_MODE; done = 1; }
    }
  }

  if constexpr (TR::kLut)
    lutw = (TR::K == 3) ? 0x0000000000FF7F00ull : 0x00FE807E01FF7F00ull; // 0x00,0x7F,0xFF(,0x01,0x7E,0x80,0xFE): rleX_Xsl.h:533-543

  // every spin is bounded: a malformed stream (or a bug) ends as DEC_ERR_STREAM, never as a hang
  uint32_t roundsLeft = B / (uint32_t)T + B / 16u + 64u; // output rounds + worst-case starved rounds (>= 16 stream bytes each)

  while (__ballot(!done && o < blen) != 0ull)
  {
    if (roundsLeft-- == 0u) { err |= DEC_ERR_STREAM | (0x100u << 0); break; }
    // ---- top-up for the NEXT round: the loads fly while this round decodes from the ring ----
    const uint32_t avail0 = E;                                         // bytes [.., avail0) are readable during this round
    issue();

    const uint32_t base = o;                                           // this round's tile row holds block bytes [base, ...)
    const uint32_t target = umin((o / (uint32_t)T + 1u) * (uint32_t)T, blen);
    uint32_t fin = (done || o >= target) ? 1u : 0u;
    uint32_t stepsLeft = 2u * (uint32_t)T + 64u;

    while (__ballot(!fin) != 0ull)
    {
      if (stepsLeft-- == 0u) { err |= DEC_ERR_STREAM | (0x100u << 1); done = 1; break; }
      // ================= one step, identical straight-line code for every lane =================
      const uint32_t resident = (avail0 > sp) ? avail0 - sp : 0u;
      const bool idle = (lit | run) == 0u;
      if (idle && last) done = 1;
      const bool streamOk = sp + 2u <= slen;
      if (idle && !done && !streamOk) { err |= DEC_ERR_STREAM | (0x100u << 2); done = 1; }
      const bool parse = idle && !done && !fin && (resident >= MAXHDR || avail0 >= lim);
      bool starved = idle && !done && !parse;

      const u32x4 v = ld128(rng + (sp & RMASK));
      const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);

      // ---- header fields, computed unconditionally, committed with selects ----
      uint32_t cnt, pos, nsym = sym4, range, used, endNow = 0;
      [[maybe_unused]] uint64_t nlut = lutw;

      if constexpr (TR::kLut)
      {
        const uint32_t w16 = v.x & 0xFFFFu;
        const uint32_t idx = w16 >> (FAM == LUT3 ? 14 : 13);
        const uint32_t c7 = (w16 >> TR::RB) & 0x7Fu;
        const uint32_t r7 = w16 & ((1u << TR::RB) - 1u);
        const bool isNew = idx == (uint32_t)TR::K;
        pos = isNew ? 3u : 2u;
        const uint32_t sb = isNew ? ((v.x >> 16) & 0xFFu) : (uint32_t)((lutw >> (8u * idx)) & 0xFFu);
        const uint32_t limit = isNew ? (uint32_t)TR::K - 1u : idx;     // entries [0, limit) move down by one
        const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
        const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
        nlut = keepHi | (low << 8) | (uint64_t)sb;
        nsym = sb * 0x01010101u;

        const uint32_t cext = ex32(lo, hi, pos);
        cnt = (c7 == 0u) ? cext : (c7 == 1u ? (cext & 0xFFFFu) : c7);
        pos += (c7 == 0u) ? 4u : (c7 == 1u ? 2u : 0u);
        const uint32_t rext = ex32(lo, hi, pos);
        range = (r7 == 0u) ? rext : (r7 == 1u ? (rext & 0xFFFFu) : r7);
        used = pos + ((r7 == 0u) ? 4u : (r7 == 1u ? 2u : 0u));
        endNow = (r7 == 1u && range == 0u) ? 1u : 0u;
        if (parse && !endNow && range < 2u) { err |= DEC_ERR_STREAM | (0x100u << 3); done = 1; }
        range = (range >= 2u) ? range - 1u : 0u;                        // literal count + 1, like the other families
      }
      else
      {
        const uint32_t b0 = v.x & 0xFFu;

        if constexpr (!TR::kPacked)
        {
          // multi: sym, cnt ...   single: cnt ...
          cnt = single ? b0 : ((v.x >> 8) & 0xFFu);
          pos = single ? 1u : 2u;
          if (!single) nsym = b0 * 0x01010101u;
          const uint32_t c32 = ex32(lo, hi, pos);
          const bool longc = cnt == 0u;
          cnt = longc ? c32 : cnt;
          pos += longc ? 4u : 0u;
        }
        else
        {
          cnt = single ? b0 : (b0 & 0x7Fu);
          pos = 1u;
          const uint32_t c32 = ex32(lo, hi, 1u);
          const bool longc = cnt == 0u;
          cnt = longc ? c32 : cnt;
          pos += longc ? 4u : 0u;
          const bool newSym = !single && !(b0 & 0x80u);
          const uint32_t sb = ex32(lo, hi, pos) & 0xFFu;
          nsym = newSym ? sb * 0x01010101u : sym4;
          pos += newSym ? 1u : 0u;
        }

        const uint32_t w = ex32(lo, hi, pos);
        const uint32_t r0 = w & 0xFFu;

        if (TR::kPacked && !single)
        {
          // 7-bit-or-4-byte range (rle8_extreme_cpu.h:1883-1899)
          const bool longr = (r0 & 1u) != 0u;
          range = longr ? (w >> 1) : (r0 >> 1);
          used = pos + (longr ? 4u : 1u);
          endNow = (longr && range == 0u) ? 1u : 0u;
        }
        else
        {
          const bool longr = r0 == 0u;
          const uint32_t r32 = ex32(lo, hi, pos + 1u);
          range = longr ? r32 : r0;
          used = pos + (longr ? 5u : 1u);
          endNow = (longr && range == 0u) ? 1u : 0u;
        }
      }

      if (parse)
      {
        const uint32_t shortv = single ? SHORT_SINGLE : SHORT_MULTI;
        sym4 = nsym;
        if constexpr (TR::kLut) lutw = nlut;
        lit = (range == 0u || endNow) ? 0u : range - 1u;               // a 7 bit range byte of 0x00 carries no literals (q11)
        run = (cnt == 0u || endNow) ? 0u : cnt + shortv - (TR::kLut ? 2u : 1u);
        last = (endNow || cnt == 0u) ? 1u : 0u;
        const uint32_t nsp = sp + used;
        if (nsp > slen || lit > slen - nsp) { err |= DEC_ERR_STREAM | (0x100u << 4); done = 1; }
        if (lit == 0u && run == 0u && !last) { err |= DEC_ERR_STREAM | (0x100u << 5); done = 1; }
      }

      const uint32_t usedNow = parse ? used : 0u;

      // ---- the bytes behind the header are the first literal bytes: shift the 16-byte register right by `used` ----
      u32x4 data;
      {
        const uint32_t sh = usedNow * 8u;
        const uint64_t dlo = (sh < 64u) ? ((lo >> (sh & 63u)) | ((hi << 1) << (63u - (sh & 63u)))) : (hi >> (sh & 63u));
        const uint64_t dhi = (sh < 64u) ? (hi >> (sh & 63u)) : 0ull;
        data = u32x4{ (uint32_t)dlo, (uint32_t)(dlo >> 32), (uint32_t)dhi, (uint32_t)(dhi >> 32) };
      }

      const bool live = !done && !fin && !starved;
      const bool isLit = lit != 0u;
      const uint32_t room = target - o;
      const uint32_t residentAfter = (resident > usedNow) ? resident - usedNow : 0u;
      uint32_t n = isLit ? umin(umin(lit, 16u - usedNow), umin(room, residentAfter)) : umin(umin(run, 16u), room);
      if (!live) n = 0;
      if (live && isLit && n == 0u && room != 0u) starved = true;       // literal bytes not resident yet

      st128(row + (o - base), isLit ? data : u32x4{ sym4, sym4, sym4, sym4 });

      sp += usedNow + (isLit ? n : 0u);
      lit -= isLit ? n : 0u;
      run -= isLit ? 0u : n;
      o += n;
      if ((lit | run) == 0u && last && !done && !starved) done = 1;
      fin |= (done || o >= target || starved) ? 1u : 0u;                // sticky for the rest of the round
    }

    // ---- flush ----
    const uint32_t produced = o - base;
    const bool uniform = __ballot(active && produced == (uint32_t)T && base == __builtin_amdgcn_readfirstlane(base)) == ~0ull;
    __syncthreads();

    if (uniform)
    {
      // fast path: all 64 rows hold T bytes of the same round: every store instruction writes RPI x T bytes = whole lines
      const uint32_t ubase = __builtin_amdgcn_readfirstlane(base);
      u32x4 fv[CPR];
#pragma unroll
      for (int q = 0; q < CPR; q++)
        fv[q] = ld128(tile + ((uint32_t)q * RPI + lane / CPR) * TS + (lane % CPR) * 16u);
#pragma unroll
      for (int q = 0; q < CPR; q++)
        st128(out + (uint64_t)(wgFirst + (uint32_t)q * RPI + lane / CPR) * B + ubase + (lane % CPR) * 16u, fv[q]);
    }
    else
    {
      rowStart[lane] = base;
      rowLen[lane] = produced;
      __syncthreads();
#pragma unroll 1
      for (int q = 0; q < CPR; q++)
      {
        const uint32_t r = (uint32_t)q * RPI + lane / CPR, c = lane % CPR;
        const uint32_t rb = wgFirst + r;
        const uint32_t valid = rowLen[r];
        const uint32_t co = c * 16u;
        if (rb >= lastBlockExcl || co >= valid)
          continue;

        uint8_t *g = out + (uint64_t)rb * B + rowStart[r] + co;
        const uint8_t *l = tile + r * TS + co;

        if (co + 16u <= valid)
          st128(g, ld128(l));
        else
          for (uint32_t k = 0; k < valid - co; k++)
            g[k] = l[k];
      }
    }

    // ---- the loads issued before the decode steps have had the whole round to arrive ----
    land();
    __syncthreads();
  }

  if (active && o != blen)
    err |= DEC_ERR_STREAM | (0x100u << 6);

  if (err != 0 && status != nullptr)
    atomicOr(status, err);
}

} // namespace hsrle
