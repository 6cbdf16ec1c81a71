// hsrle_decode.hip.h -- block-parallel decoder for every rleX_extreme packet grammar.
//
// Replaces the reference's decode bodies:
//   8 bit multi/single      src/rle8_extreme_cpu.h:702-764 (dispatch), :1546-2006, :2008-2434
//   16/32/64 bit            src/rleX_extreme_cpu.h:82-111, src/rleX_extreme_cpu_decode.h:27-164
//   24/48 bit               src/rle{24,48}_extreme_cpu_decode.h
//   128 bit                 src/rle128_extreme_cpu.h:499-802
//   3/7 symbol LUT          src/rleX_Xsl.h:530-1881
//   MEMCPY_* / MEMSET_*     src/rleX_extreme_common.h:32-312  (-> ring-to-tile copies / fill_run below)
//
// One lane decodes one block (= one complete reference stream); a 64-lane workgroup owns 64 consecutive blocks, i.e.
// one contiguous 64 * blockSize slice of the output.  Data path per workgroup:
//
//   HBM --(16-byte vector loads, every lane streaming its own block, issued one round ahead)--> LDS ring [64][R + 16]
//   LDS ring --(per-lane packet walk: header fields, 16-byte literal vectors)--> LDS tile [64][T + 16]
//   LDS tile --(8 lanes x 16 B per row: whole 128-byte lines)--> HBM
//
// so the packet-to-packet dependency chain (the next header's position is known only after the previous packet's literal
// length, reference: src/rleX_extreme_cpu_decode.h:129-162) only ever waits on LDS, never on HBM, and it is walked
// by 64 lanes at once.  Nothing outside [0, uncompressedSize) is written (the reference scribbles up to 128 bytes past
// the end).
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

enum DecodeError : uint32_t
{
  DEC_ERR_HEADER = 1u,   // block stream header does not match the container table
  DEC_ERR_STREAM = 2u,   // packet chain leaves the stream / ends early
  DEC_ERR_MODE = 4u      // unknown 8 bit mode byte (reference: rle8_extreme_cpu.h:759-760)
};

// 32-byte per-lane fill pattern in LDS: pat[k] = sym[k % S].  `v` holds the symbol in its low S bytes.
template <int S>
__device__ __forceinline__ void set_pattern(uint8_t *pat, u32x4 v)
{
  u32x4 a, b;

  if constexpr (S == 1) { const uint32_t d = (v.x & 0xFFu) * 0x01010101u; a = u32x4{ d, d, d, d }; b = a; }
  else if constexpr (S == 2) { const uint32_t h = v.x & 0xFFFFu; const uint32_t d = h | (h << 16); a = u32x4{ d, d, d, d }; b = a; }
  else if constexpr (S == 4) { a = u32x4{ v.x, v.x, v.x, v.x }; b = a; }
  else if constexpr (S == 8) { a = u32x4{ v.x, v.y, v.x, v.y }; b = a; }
  else if constexpr (S == 16) { a = v; b = v; }
  else if constexpr (S == 3)
  {
    const uint32_t t = v.x & 0xFFFFFFu;
    const uint32_t d0 = t | (t << 24), d1 = (t >> 8) | (t << 16), d2 = (t >> 16) | (t << 8);
    a = u32x4{ d0, d1, d2, d0 };
    b = u32x4{ d1, d2, d0, d1 };
  }
  else // S == 6
  {
    const uint32_t lo = v.x, hi = v.y & 0xFFFFu;
    const uint32_t d0 = lo, d1 = hi | (lo << 16), d2 = (lo >> 16) | (hi << 16);
    a = u32x4{ d0, d1, d2, d0 };
    b = u32x4{ d1, d2, d0, d1 };
  }

  st128(pat, a);
  st128(pat + 16, b);
}

// Write m bytes of the run pattern starting at pattern phase `phase` (0..S-1); may write 15 bytes beyond dst + m.
template <int S>
__device__ __forceinline__ void fill_run(uint8_t *dst, const uint8_t *pat, uint32_t m, uint32_t phase)
{
  if constexpr (16 % S == 0)
  {
    const u32x4 v = ld128(pat + phase);
    for (uint32_t k = 0; k < m; k += 16)
      st128(dst + k, v);
  }
  else
  {
    for (uint32_t k = 0; k < m; k += 16)
    {
      st128(dst + k, ld128(pat + phase));
      phase = (phase + 16u) % (uint32_t)S;
    }
  }
}

template <int S>
__device__ __forceinline__ u32x4 mask_symbol(u32x4 v)
{
  if constexpr (S == 1) return u32x4{ v.x & 0xFFu, 0, 0, 0 };
  else if constexpr (S == 2) return u32x4{ v.x & 0xFFFFu, 0, 0, 0 };
  else if constexpr (S == 3) return u32x4{ v.x & 0xFFFFFFu, 0, 0, 0 };
  else if constexpr (S == 4) return u32x4{ v.x, 0, 0, 0 };
  else if constexpr (S == 6) return u32x4{ v.x, v.y & 0xFFFFu, 0, 0 };
  else if constexpr (S == 8) return u32x4{ v.x, v.y, 0, 0 };
  else return v;
}

// FAM in {PLAIN, PACKED, LUT3, LUT7}; the 8 bit PLAIN / PACKED kernels also decode the Single modes (mode byte 1),
// exactly like rle8_decompress / rle8_packed_decompress do.
//   T = output bytes per lane and round (LDS tile row), R = per-lane stream ring size in LDS (power of two).
template <int FAM, int S, int AL, int T, int R>
__global__ __launch_bounds__(64) void k_decode_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets,
                                                      const uint8_t *__restrict__ payloadEnd, uint8_t *__restrict__ out, uint64_t U,
                                                      uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int TS = T + 16;                 // tile row stride: 16 bytes of over-write slack
  constexpr int RS = R + 16;                 // ring row stride: bytes [R, R+16) mirror [0, 16) so 16-byte reads never wrap
  constexpr int CPR = T / 16;                // 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR;              // tile rows covered by one flush instruction
  constexpr uint32_t RMASK = (uint32_t)R - 1u;
  constexpr uint32_t MAXHDR = 1u + 4u + (uint32_t)S + 4u + 2u; // longest packet header of any family
  constexpr int NPF = T / 16 + 1;            // 16-byte chunks one top-up can carry per lane
  constexpr bool kPatInLds = (S != 1);       // 8 bit: the fill pattern is one broadcast register
  static_assert((R & (R - 1)) == 0 && R >= 128, "ring size must be a power of two");
  static_assert(T % 64 == 0 && T <= 256, "tile rows are flushed as 64/128-byte pieces");

  __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TS];
  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * RS];
  __shared__ __attribute__((aligned(16))) uint8_t pats[kPatInLds ? 64 * 32 : 16];
  __shared__ uint32_t rowStart[64], rowLen[64]; // slow flush path: block offset / length of what the row holds in the tile

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = firstBlock + blockIdx.x * 64u;
  const uint32_t lastBlockExcl = firstBlock + blockCount;
  const uint32_t b = wgFirst + lane;
  const bool active = b < lastBlockExcl;

  uint8_t *const row = tile + lane * TS;
  uint8_t *const rng = ring + lane * RS;
  uint8_t *const pat = pats + (kPatInLds ? lane * 32 : 0);

  // ---- per-lane stream state ----
  const uint8_t *s = payload;   // this lane's stream (global)
  uint32_t slen = 0, blen = 0;
  uint32_t sp = 0;        // read position in the stream
  uint32_t E = 0;         // stream bytes [.., E) are in the ring (multiple of 16)
  uint32_t lim = 0;       // E never exceeds lim (loadable bytes of this stream incl. the payload tail pad)
  uint32_t lit = 0;       // literal bytes of the current packet still to copy
  uint32_t run = 0;       // run bytes of the current packet still to write
  uint32_t phase = 0;     // pattern phase of the next run byte
  uint32_t o = 0;         // bytes of this block produced so far
  uint32_t sym1 = 0;      // S == 1: current symbol, byte-broadcast
  bool last = false;      // the stream ends after the current packet's literals
  bool done = true;
  bool single = false;
  uint32_t err = 0;
  [[maybe_unused]] uint32_t lut[TR::kLut ? TR::K : 1][TR::SW];

  if (active)
  {
    const uint64_t off0 = offsets[b];
    s = payload + off0;
    slen = (uint32_t)(offsets[b + 1] - off0);
    const uint64_t start = (uint64_t)b * B;
    blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
    const uint64_t room = (uint64_t)(payloadEnd - s);
    lim = (uint32_t)(room > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : room) & ~15u;
    done = false;
  }

  auto set_sym = [&](u32x4 v) {
    if constexpr (S == 1) sym1 = (v.x & 0xFFu) * 0x01010101u;
    else set_pattern<S>(pat, v);
  };

  // ---- ring top-up: every lane streams its own block; loads are issued one round before their bytes are needed ----
  u32x4 pf[NPF];
  uint32_t want = 0;

  auto issue = [&]() {
    const uint32_t resident = E - (sp & ~15u);
    want = umin(umin(((uint32_t)R - resident) >> 4, (uint32_t)NPF), (lim - E) >> 4);
    if (done) want = 0;
#pragma unroll
    for (int q = 0; q < NPF; q++)
      if ((uint32_t)q < want)
        pf[q] = ld128(s + E + 16u * q);
  };

  auto land = [&]() {
#pragma unroll
    for (int q = 0; q < NPF; q++)
      if ((uint32_t)q < want)
      {
        const uint32_t ro = (E + 16u * q) & RMASK;
        st128(rng + ro, pf[q]);
        if (ro == 0u)
          st128(rng + R, pf[q]); // mirror of the first 16 ring bytes
      }
    E += want << 4;
  };

  // prologue: fill the ring, then read the stream header from it
  for (int k = 0; k < (R / 16 + NPF - 1) / NPF; k++)
  {
    issue();
    land();
  }
  __syncthreads();

  if (active)
  {
    sp = TR::kHeaderSize;

    if (slen < TR::kHeaderSize + 2u || ld32(rng) != blen || ld32(rng + 4) != slen)
    {
      err |= DEC_ERR_HEADER;
      done = true;
    }
    else if constexpr (S == 1 && !TR::kLut)
    {
      const uint32_t mode = rng[8];
      if (mode == 1u) { single = true; set_sym(u32x4{ rng[9], 0, 0, 0 }); sp = 10; }
      else if (mode != 0u) { err |= DEC_ERR_MODE; done = true; }
    }
  }

  if constexpr (TR::kLut)
  {
    // initial move-to-front list (reference: rleX_Xsl.h:533-543)
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
    for (int k = 0; k < TR::K; k++)
    {
      const u32x4 m = mask_symbol<S>(u32x4{ init[k] * 0x01010101u, init[k] * 0x01010101u, init[k] * 0x01010101u, init[k] * 0x01010101u });
#pragma unroll
      for (int w = 0; w < TR::SW; w++)
        lut[k][w] = m[w];
    }
    set_sym(u32x4{ 0, 0, 0, 0 });
  }
  else if (!single)
  {
    set_sym(u32x4{ 0, 0, 0, 0 }); // Packed decoders start with symbol 0 (rleX_extreme_cpu_decode.h:31-37, q5)
  }

#ifdef HSRLE_STAMPS
  unsigned long long tIssue = 0, tDecode = 0, tFlush = 0, tLand = 0, nRounds = 0, nIter = 0, t0, t1;
#define HS_STAMP(acc) { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; }
#else
#define HS_STAMP(acc)
#endif

  while (__ballot(!done && o < blen) != 0ull)
  {
#ifdef HSRLE_STAMPS
    t0 = __builtin_readcyclecounter(); nRounds++;
#endif
    // ---- top-up for the NEXT round: the loads fly while this round decodes from the ring ----
    const uint32_t avail0 = E;                                         // bytes [.., avail0) are readable during this round
    issue();
    HS_STAMP(tIssue)

    const uint32_t base = o;                                           // this round's tile row holds block bytes [base, ...)
    const uint32_t target = umin((o / (uint32_t)T + 1u) * (uint32_t)T, blen);

    while (!done && o < target)
    {
#ifdef HSRLE_STAMPS
      nIter++;
#endif
      if (lit == 0 && run == 0)
      {
        if (last) { done = true; break; }
        if (sp + 2u > slen) { err |= DEC_ERR_STREAM; done = true; break; }
        if (avail0 - sp < MAXHDR && avail0 < lim) break;               // header not resident yet: continue next round

        // ---------------- packet header ----------------
        uint32_t cnt, range, used;
        bool endNow = false;
#define HS_H(off) (rng + ((sp + (off)) & RMASK))

        if constexpr (TR::kLut)
        {
          const uint32_t v = ld16(HS_H(0));
          used = 2;
          const uint32_t idx = v >> (FAM == LUT3 ? 14 : 13);
          cnt = (v >> TR::RB) & 0x7Fu;
          range = v & ((1u << TR::RB) - 1u);

          if (idx != 0)
          {
            uint32_t tmp[TR::SW];
            if (idx == (uint32_t)TR::K)
            {
              const u32x4 nv = mask_symbol<S>(ld128(HS_H(used)));
              used += S;
#pragma unroll
              for (int w = 0; w < TR::SW; w++) tmp[w] = nv[w];
            }
            else
            {
#pragma unroll
              for (int w = 0; w < TR::SW; w++) tmp[w] = lut[0][w];
#pragma unroll
              for (int k = 1; k < TR::K; k++)
                if (idx == (uint32_t)k)
                {
#pragma unroll
                  for (int w = 0; w < TR::SW; w++) tmp[w] = lut[k][w];
                }
            }
            const uint32_t limit = (idx == (uint32_t)TR::K) ? (uint32_t)TR::K - 1u : idx;
#pragma unroll
            for (int k = TR::K - 1; k >= 1; k--)
              if ((uint32_t)k <= limit)
              {
#pragma unroll
                for (int w = 0; w < TR::SW; w++) lut[k][w] = lut[k - 1][w];
              }
            u32x4 pv = u32x4{ 0, 0, 0, 0 };
#pragma unroll
            for (int w = 0; w < TR::SW; w++) { lut[0][w] = tmp[w]; pv[w] = tmp[w]; }
            set_sym(pv);
          }

          if (cnt == 0) { cnt = ld32(HS_H(used)); used += 4; }
          else if (cnt == 1) { cnt = ld16(HS_H(used)); used += 2; }

          if (range == 0) { range = ld32(HS_H(used)); used += 4; }
          else if (range == 1) { range = ld16(HS_H(used)); used += 2; endNow = (range == 0); }

          if (!endNow && range < 2u) { err |= DEC_ERR_STREAM; done = true; break; }
          lit = endNow ? 0u : range - 2u;
          run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + 3u / (uint32_t)S - 2u) * (uint32_t)S : cnt + 1u);
        }
        else if constexpr (S == 1)
        {
          // 8 bit: the whole header (<= 10 bytes) comes from ONE 16-byte ring read; fields are picked with shifts
          const u32x4 hv = ld128(HS_H(0));
          const uint64_t lo = (uint64_t)hv.x | ((uint64_t)hv.y << 32), hi = (uint64_t)hv.z | ((uint64_t)hv.w << 32);
          uint32_t pos;

          if (single)
          {
            cnt = hv.x & 0xFFu;
            pos = 1;
            if (cnt == 0) { cnt = (uint32_t)(lo >> 8); pos = 5; }
          }
          else if constexpr (!TR::kPacked)
          {
            sym1 = (hv.x & 0xFFu) * 0x01010101u;
            cnt = (hv.x >> 8) & 0xFFu;
            pos = 2;
            if (cnt == 0) { cnt = (uint32_t)(lo >> 16); pos = 6; }
          }
          else
          {
            const uint32_t x = hv.x & 0xFFu;
            cnt = x & 0x7Fu;
            pos = 1;
            if (cnt == 0) { cnt = (uint32_t)(lo >> 8); pos = 5; }
            if (!(x & 0x80u)) { sym1 = ((uint32_t)(lo >> (8u * pos)) & 0xFFu) * 0x01010101u; pos += 1; }
          }

          const uint32_t w = (uint32_t)((lo >> (8u * pos)) | (hi << (64u - 8u * pos))); // pos in 1..7
          const uint32_t r0 = w & 0xFFu;

          if (TR::kRange7 && !single)
          {
            if (r0 & 1u) { range = w >> 1; used = pos + 4; endNow = (range == 0); }
            else { range = r0 >> 1; used = pos + 1; }
          }
          else
          {
            range = r0; used = pos + 1;
            if (r0 == 0)
            {
              const uint32_t pos2 = pos + 1; // 2..8
              range = (pos2 < 8u) ? (uint32_t)((lo >> (8u * pos2)) | (hi << (64u - 8u * pos2))) : (uint32_t)hi;
              used = pos + 5;
              endNow = (range == 0);
            }
          }

          lit = (range == 0) ? 0u : range - 1u; // a 7 bit range byte of 0x00 carries no literals (A.5 q11)

          if (cnt == 0) run = 0;
          else if (single) run = cnt + ((FAM == PACKED) ? 2u : 4u) - 1u;
          else run = cnt + TR::SHORT - 1u;
        }
        else
        {
          used = 0;

          if constexpr (!TR::kPacked)
          {
            set_sym(ld128(HS_H(0)));
            used = S;
            cnt = *HS_H(used); used += 1;
            if (cnt == 0) { cnt = ld32(HS_H(used)); used += 4; }
          }
          else
          {
            const uint32_t x = *HS_H(0);
            used = 1;
            cnt = x & 0x7Fu;
            if (cnt == 0) { cnt = ld32(HS_H(used)); used += 4; }
            if (!(x & 0x80u)) { set_sym(ld128(HS_H(used))); used += S; }
          }

          if constexpr (TR::kRange7)
          {
            const uint32_t r0 = *HS_H(used);
            if (r0 & 1u) { range = ld32(HS_H(used)) >> 1; used += 4; endNow = (range == 0); }
            else { range = r0 >> 1; used += 1; }
          }
          else
          {
            range = *HS_H(used); used += 1;
            if (range == 0) { range = ld32(HS_H(used)); used += 4; endNow = (range == 0); }
          }

          lit = (range == 0) ? 0u : range - 1u;
          run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + TR::SHORT / (uint32_t)S - 1u) * (uint32_t)S : cnt + TR::SHORT - 1u);
        }
#undef HS_H

        sp += used;
        phase = 0;
        last = endNow || (cnt == 0);
        if (endNow) { lit = 0; run = 0; }

        if (sp > slen || lit > slen - sp) { err |= DEC_ERR_STREAM; done = true; break; }
        if (lit == 0 && run == 0 && !last) { err |= DEC_ERR_STREAM; done = true; break; }
        continue;
      }

      if (lit != 0)
      {
        const uint32_t resident = (avail0 > sp) ? avail0 - sp : 0u;
        const uint32_t n = umin(umin(lit, target - o), resident);
        if (n == 0) break;                                             // literals not resident yet: continue next round

        // loads may over-read (harmless, stays inside this lane's ring row); stores are bounded by n
        uint8_t *dst = row + (o - base);
        for (uint32_t k = 0; k < n; k += 64)
        {
          const u32x4 v0 = ld128(rng + ((sp + k) & RMASK));
          const u32x4 v1 = ld128(rng + ((sp + k + 16) & RMASK));
          const u32x4 v2 = ld128(rng + ((sp + k + 32) & RMASK));
          const u32x4 v3 = ld128(rng + ((sp + k + 48) & RMASK));
          st128(dst + k, v0);
          if (k + 16 < n) st128(dst + k + 16, v1);
          if (k + 32 < n) st128(dst + k + 32, v2);
          if (k + 48 < n) st128(dst + k + 48, v3);
        }
        sp += n;
        lit -= n;
        o += n;
      }

      if (lit == 0 && run != 0 && o < target)
      {
        const uint32_t m = umin(run, target - o);
        if constexpr (S == 1)
        {
          const u32x4 v = u32x4{ sym1, sym1, sym1, sym1 };
          uint8_t *dst = row + (o - base);
          for (uint32_t k = 0; k < m; k += 64)
          {
            st128(dst + k, v);
            if (k + 16 < m) st128(dst + k + 16, v);
            if (k + 32 < m) st128(dst + k + 32, v);
            if (k + 48 < m) st128(dst + k + 48, v);
          }
        }
        else
        {
          fill_run<S>(row + (o - base), pat, m, phase);
          phase = (phase + m) % (uint32_t)S;
        }
        run -= m;
        o += m;
      }
    }

    HS_STAMP(tDecode)
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

    HS_STAMP(tFlush)
    // ---- the loads issued before the decode step have had the whole round to arrive ----
    land();
    __syncthreads();
    HS_STAMP(tLand)
  }

#ifdef HSRLE_STAMPS
  // diagnostic build only: per-phase cycle sums of every workgroup's lane 0, appended behind the status word
  if (status != nullptr)
  {
    unsigned long long *dbg = (unsigned long long *)(status + 16);
    const unsigned long long iters = __builtin_amdgcn_readfirstlane((uint32_t)nIter);
    if (lane == 0)
    {
      atomicAdd(dbg + 0, tIssue); atomicAdd(dbg + 1, tDecode); atomicAdd(dbg + 2, tFlush); atomicAdd(dbg + 3, tLand);
      atomicAdd(dbg + 4, nRounds); atomicAdd(dbg + 5, iters); atomicAdd(dbg + 6, 1ull);
    }
  }
#endif

  if (active && o != blen)
    err |= DEC_ERR_STREAM;

  if (err != 0 && status != nullptr)
    atomicOr(status, err);
}

} // namespace hsrle
