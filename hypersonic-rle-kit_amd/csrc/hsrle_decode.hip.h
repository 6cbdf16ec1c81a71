// hsrle_decode.hip.h -- block-parallel decoder for every rleX_extreme packet grammar.
//
// Replaces the reference's decode bodies:
//   8 bit multi/single      src/rle8_extreme_cpu.h:702-764 (dispatch), :1546-2006, :2008-2434
//   16/32/64 bit            src/rleX_extreme_cpu.h:82-111, src/rleX_extreme_cpu_decode.h:27-164
//   24/48 bit               src/rle{24,48}_extreme_cpu_decode.h
//   128 bit                 src/rle128_extreme_cpu.h:499-802
//   3/7 symbol LUT          src/rleX_Xsl.h:530-1881
//   MEMCPY_* / MEMSET_*     src/rleX_extreme_common.h:32-312  (-> copy_over / fill_run below)
//
// One lane decodes one block (= one complete reference stream); a 64-lane workgroup therefore owns 64 consecutive
// blocks, i.e. one contiguous 64 * blockSize slice of the output.  The output is produced in rounds of T bytes per
// lane into an LDS tile [64][T + 16]; after every round the wave flushes the tile with 16-byte-per-lane stores so
// that every global store instruction writes whole 128-byte lines.  Nothing outside [0, uncompressedSize) is written
// (the reference scribbles up to 128 bytes past the end).
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
template <int FAM, int S, int AL, int T>
__global__ __launch_bounds__(64) void k_decode_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets,
                                                      const uint8_t *__restrict__ payloadEnd, uint8_t *__restrict__ out, uint64_t U,
                                                      uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int TS = T + 16;      // row stride: 16 bytes of over-write slack
  constexpr int CPR = T / 16;     // 16-byte chunks per row
  static_assert(T % 128 == 0, "rows are flushed as whole 128-byte lines");

  __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TS];
  __shared__ __attribute__((aligned(16))) uint8_t pats[64 * 32];

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = firstBlock + blockIdx.x * 64u;            // first block of this workgroup
  const uint32_t lastBlockExcl = firstBlock + blockCount;
  const uint32_t b = wgFirst + lane;
  const bool active = b < lastBlockExcl;

  uint8_t *const row = tile + lane * TS;
  uint8_t *const pat = pats + lane * 32;

  // ---- per-lane stream state ----
  const uint8_t *s = payload;
  uint32_t slen = 0, blen = 0;
  uint32_t sp = 0;        // read position in the stream
  uint32_t lit = 0;       // literal bytes of the current packet still to copy
  uint32_t run = 0;       // run bytes of the current packet still to write
  uint32_t phase = 0;     // pattern phase of the next run byte
  uint32_t o = 0;         // bytes of this block produced so far
  bool last = false;      // the stream ends after the current packet's literals
  bool done = true;
  bool single = false;
  uint32_t err = 0;
  [[maybe_unused]] uint32_t lut[TR::kLut ? TR::K : 1][TR::SW];

  if (active)
  {
    const uint64_t off0 = offsets[b], off1 = offsets[b + 1];
    s = payload + off0;
    slen = (uint32_t)(off1 - off0);
    const uint64_t start = (uint64_t)b * B;
    blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
    done = false;
    sp = TR::kHeaderSize;

    if (slen < TR::kHeaderSize + 2u || ld32(s) != blen || ld32(s + 4) != slen)
    {
      err |= DEC_ERR_HEADER;
      done = true;
    }
    else if constexpr (S == 1 && !TR::kLut)
    {
      const uint32_t mode = s[8];
      if (mode == 1u) { single = true; set_pattern<1>(pat, u32x4{ s[9], 0, 0, 0 }); sp = 10; }
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
    set_pattern<S>(pat, u32x4{ 0, 0, 0, 0 });
  }
  else if (!single)
  {
    set_pattern<S>(pat, u32x4{ 0, 0, 0, 0 }); // Packed decoders start with symbol 0 (rleX_extreme_cpu_decode.h:31-37, q5)
  }

  const uint32_t nRounds = (B + T - 1) / T;

  for (uint32_t round = 0; round < nRounds; round++)
  {
    const uint32_t base = round * T;
    const uint32_t target = umin(base + T, blen);

    while (!done && o < target)
    {
      if (lit == 0 && run == 0)
      {
        if (last) { done = true; break; }
        if (sp + 2u > slen) { err |= DEC_ERR_STREAM; done = true; break; }

        // ---------------- packet header ----------------
        const uint8_t *h = s + sp;
        uint32_t cnt, range, used;
        bool endNow = false;

        if constexpr (TR::kLut)
        {
          const uint32_t v = ld16(h);
          used = 2;
          const uint32_t idx = v >> (FAM == LUT3 ? 14 : 13);
          cnt = (v >> TR::RB) & 0x7Fu;
          range = v & ((1u << TR::RB) - 1u);

          if (idx != 0)
          {
            uint32_t tmp[TR::SW];
            if (idx == (uint32_t)TR::K)
            {
              const u32x4 nv = mask_symbol<S>(ld128(h + used));
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
            set_pattern<S>(pat, pv);
          }

          if (cnt == 0) { cnt = ld32(h + used); used += 4; }
          else if (cnt == 1) { cnt = ld16(h + used); used += 2; }

          if (range == 0) { range = ld32(h + used); used += 4; }
          else if (range == 1) { range = ld16(h + used); used += 2; endNow = (range == 0); }

          if (!endNow && range < 2u) { err |= DEC_ERR_STREAM; done = true; break; }
          lit = endNow ? 0u : range - 2u;
          run = (cnt == 0) ? 0u : (TR::kAligned ? (cnt + 3u / (uint32_t)S - 2u) * (uint32_t)S : cnt + 1u);
        }
        else
        {
          used = 0;

          if (single)
          {
            cnt = h[0]; used = 1;
            if (cnt == 0) { cnt = ld32(h + used); used += 4; }
          }
          else if constexpr (!TR::kPacked)
          {
            set_pattern<S>(pat, ld128(h));
            used = S;
            cnt = h[used]; used += 1;
            if (cnt == 0) { cnt = ld32(h + used); used += 4; }
          }
          else
          {
            const uint32_t x = h[0];
            used = 1;
            cnt = x & 0x7Fu;
            if (cnt == 0) { cnt = ld32(h + used); used += 4; }
            if (!(x & 0x80u)) { set_pattern<S>(pat, ld128(h + used)); used += S; }
          }

          if (TR::kRange7 && !single)
          {
            const uint32_t r0 = h[used];
            if (r0 & 1u) { range = ld32(h + used) >> 1; used += 4; endNow = (range == 0); }
            else { range = r0 >> 1; used += 1; }
          }
          else
          {
            range = h[used]; used += 1;
            if (range == 0) { range = ld32(h + used); used += 4; endNow = (range == 0); }
          }

          lit = (range == 0) ? 0u : range - 1u; // a 7 bit range byte of 0x00 carries no literals (A.5 q11)

          if (cnt == 0) run = 0;
          else if (single) run = cnt + ((FAM == PACKED) ? 2u : 4u) - 1u;
          else run = TR::kAligned ? (cnt + TR::SHORT / (uint32_t)S - 1u) * (uint32_t)S : cnt + TR::SHORT - 1u;
        }

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
        const uint32_t n = umin(lit, target - o);
        copy_over(row + (o - base), s + sp, n, payloadEnd);
        sp += n;
        lit -= n;
        o += n;
      }

      if (lit == 0 && run != 0 && o < target)
      {
        const uint32_t m = umin(run, target - o);
        fill_run<S>(row + (o - base), pat, m, phase);
        run -= m;
        o += m;
        phase = (phase + m) % (uint32_t)S;
      }
    }

    __syncthreads();

    // ---- flush: every store instruction covers 8 rows x 128 bytes = 8 whole lines ----
    for (uint32_t j = lane; j < 64u * CPR; j += 64u)
    {
      const uint32_t r = j / CPR, c = j % CPR;
      const uint32_t rb = wgFirst + r;
      if (rb >= lastBlockExcl)
        continue;

      const uint64_t rstart = (uint64_t)rb * B;
      const uint64_t rlen64 = (U - rstart) < (uint64_t)B ? (U - rstart) : (uint64_t)B;
      const uint32_t rlen = (uint32_t)rlen64;
      if (base >= rlen)
        continue;

      const uint32_t valid = umin((uint32_t)T, rlen - base);
      const uint32_t co = c * 16u;
      if (co >= valid)
        continue;

      uint8_t *g = out + rstart + base + co;
      const uint8_t *l = tile + r * TS + co;

      if (co + 16u <= valid)
        st128(g, ld128(l));
      else
        for (uint32_t k = 0; k < valid - co; k++)
          g[k] = l[k];
    }

    __syncthreads();
  }

  if (active && o != blen)
    err |= DEC_ERR_STREAM;

  if (err != 0 && status != nullptr)
    atomicOr(status, err);
}

} // namespace hsrle
