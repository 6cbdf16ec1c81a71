// hsrle_encode.hip.h -- block-parallel encoders for every rleX_extreme codec.
//
// Replaces the reference's encode bodies:
//   8 bit multi (plain/Packed)   src/rle8_extreme_cpu.h:86-344, :768-931 (sse2), :936-1099 (avx2; canonical, A.5 q1)
//   8 bit single                 src/rle8_extreme_cpu.h:346-700, :1103-1321; symbol pick src/rle8_extreme_cpu.c:53-153
//   16/32/64 bit                 src/rleX_extreme_cpu.h:47-60, src/rleX_extreme_cpu_encode.h:14-609
//   24/48 bit                    src/rle{24,48}_extreme_cpu_encode.h
//   128 bit                      src/rle128_extreme_cpu.h:32-497
//   3/7 symbol LUT               src/rleX_Xsl.h:93-486, src/rleX_Xsl_multibyte_encoder.h:18-370
//
// One lane encodes one block into its own staging slot (slot stride >= rle_compress_bounds(blockSize)); a second
// kernel (hsrle_container.hip) scans the sizes and compacts the slots into the container payload.  Every block stream
// is byte-identical to what the reference encoder writes for the same block ("bytes at or beyond the block end never
// match").  The CPU's movemask/ctz scans (rle8_extreme_cpu.h:952-1084) become per-lane 16-byte vector
// loads with SWAR byte compares; the emit decisions stay a sequential state machine per block, which
// is exactly what one-lane-per-block gives for free.
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

struct Sink
{
  uint8_t *o;
  uint32_t at;
  const uint8_t *inEnd; // one past the last readable input byte (global)

  __device__ __forceinline__ void put8(uint32_t v) { o[at] = (uint8_t)v; at += 1; }
  __device__ __forceinline__ void put16(uint32_t v) { st16(o + at, v); at += 2; }
  __device__ __forceinline__ void put32(uint32_t v) { st32(o + at, v); at += 4; }
  __device__ __forceinline__ void putn(const uint8_t *src, uint32_t n) { copy_over(o + at, src, n, inEnd); at += n; }
  __device__ __forceinline__ void patch32(uint32_t pos, uint32_t v) { st32(o + pos, v); }

  template <int S>
  __device__ __forceinline__ void put_sym(u32x4 v)
  {
    if constexpr (S == 1) put8(v.x);
    else if constexpr (S == 2) put16(v.x);
    else if constexpr (S == 3) { put16(v.x); put8(v.x >> 16); }
    else if constexpr (S == 4) put32(v.x);
    else if constexpr (S == 6) { put32(v.x); put16(v.y); }
    else if constexpr (S == 8) { put32(v.x); put32(v.y); }
    else { st128(o + at, v); at += 16; }
  }
};

// exactly S bytes at p, zero extended
template <int S>
__device__ __forceinline__ u32x4 load_sym(const uint8_t *p)
{
  if constexpr (S == 1) return u32x4{ ld8(p), 0, 0, 0 };
  else if constexpr (S == 2) return u32x4{ ld16(p), 0, 0, 0 };
  else if constexpr (S == 3) return u32x4{ ld16(p) | (ld8(p + 2) << 16), 0, 0, 0 };
  else if constexpr (S == 4) return u32x4{ ld32(p), 0, 0, 0 };
  else if constexpr (S == 6) return u32x4{ ld32(p), ld16(p + 4), 0, 0 };
  else if constexpr (S == 8) { const uint64_t v = ld64(p); return u32x4{ (uint32_t)v, (uint32_t)(v >> 32), 0, 0 }; }
  else return ld128(p);
}

__device__ __forceinline__ bool sym_eq(u32x4 a, u32x4 b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) == 0; }

// ------------------------------------------------------------------------------------------------------------------
// run discovery.  Returns false when no further run exists; [p, e) is the run.

// 8 bit: maximal runs of equal bytes of length >= 2 (rle8_extreme_cpu.h:1062-1089).
__device__ __forceinline__ bool next_run8(const uint8_t *d, uint32_t n, uint32_t &i, uint32_t &p, uint32_t &e)
{
  uint32_t q = i;
  bool found = false;

  // 16 positions per step: z = x ^ (x shifted by one byte); a zero byte in z is an adjacent equal pair
  while (q + 20u <= n)
  {
    const u32x4 x = ld128(d + q);
    const uint32_t x4 = ld32(d + q + 16);
    const uint32_t m0 = zero_bytes(x.x ^ alignbyte(x.y, x.x, 1));
    const uint32_t m1 = zero_bytes(x.y ^ alignbyte(x.z, x.y, 1));
    const uint32_t m2 = zero_bytes(x.z ^ alignbyte(x.w, x.z, 1));
    const uint32_t m3 = zero_bytes(x.w ^ alignbyte(x4, x.w, 1));

    if ((m0 | m1 | m2 | m3) != 0)
    {
      if (m0) q += first_set_byte(m0);
      else if (m1) q += 4u + first_set_byte(m1);
      else if (m2) q += 8u + first_set_byte(m2);
      else q += 12u + first_set_byte(m3);
      found = true;
      break;
    }
    q += 16u;
  }

  if (!found)
  {
    for (; q + 1u < n; q++)
      if (d[q] == d[q + 1]) { found = true; break; }

    if (!found) { i = n; return false; }
  }

  const uint32_t sym = d[q];
  const uint32_t bs = sym * 0x01010101u;
  uint32_t r = q + 2u;
  bool stop = false;

  while (r + 16u <= n)
  {
    const u32x4 x = ld128(d + r);
    const uint32_t w0 = x.x ^ bs, w1 = x.y ^ bs, w2 = x.z ^ bs, w3 = x.w ^ bs;

    if ((w0 | w1 | w2 | w3) != 0)
    {
      if (w0) r += first_set_byte(w0);
      else if (w1) r += 4u + first_set_byte(w1);
      else if (w2) r += 8u + first_set_byte(w2);
      else r += 12u + first_set_byte(w3);
      stop = true;
      break;
    }
    r += 16u;
  }

  if (!stop)
    while (r < n && d[r] == sym)
      r++;

  p = q;
  e = r;
  i = r;
  return true;
}

template <int S>
__device__ __forceinline__ bool eq_sym_at(const uint8_t *a, const uint8_t *b)
{
  if constexpr (S == 2) return ld16(a) == ld16(b);
  else if constexpr (S == 3) return ld16(a) == ld16(b) && a[2] == b[2];
  else if constexpr (S == 4) return ld32(a) == ld32(b);
  else if constexpr (S == 6) return ld32(a) == ld32(b) && ld16(a + 4) == ld16(b + 4);
  else if constexpr (S == 8) return ld64(a) == ld64(b);
  else return sym_eq(ld128(a), ld128(b));
}

// S > 1: first p with d[p..p+S) == d[p+S..p+2S), extended by whole symbols and (byte-aligned variants) by the matching
// leading bytes of the next partial symbol (rleX_extreme_cpu_encode.h:79-163, :315-371).
template <int S, bool ALIGNED>
__device__ __forceinline__ bool next_runS(const uint8_t *d, uint32_t n, uint32_t &i, uint32_t &p, uint32_t &e)
{
  uint32_t q = i;

  if constexpr (S == 2 || S == 4)
  {
    // window of dwords: position q+k matches iff bytes [q+k, q+k+S) == [q+k+S, q+k+2S)
    while (q + 12u <= n)
    {
      const uint32_t w0 = ld32(d + q), w1 = ld32(d + q + 4), w2 = ld32(d + q + 8);
      bool hit = false;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
      {
        const uint32_t a = (k == 0) ? w0 : alignbyte(w1, w0, k);
        const uint32_t bb = (k == 0) ? w1 : alignbyte(w2, w1, k);
        const bool m = (S == 4) ? (a == bb) : (((a ^ (a >> 16)) & 0xFFFFu) == 0);
        if (m && !hit) { hit = true; q += k; }
      }
      if (hit) goto found;
      q += 4u;
    }
  }

  for (;; q++)
  {
    if (q + 2u * S > n) { i = n; return false; }
    if (eq_sym_at<S>(d + q, d + q + S)) break;
  }

found:
  uint32_t r = q + 2u * S;
  while (r + S <= n && eq_sym_at<S>(d + r, d + q))
    r += S;

  if constexpr (!ALIGNED)
  {
    if (r + S <= n)
    {
      uint32_t j = 0;
      while (j < (uint32_t)S && d[r + j] == d[q + j])
        j++;
      r += j;
    }
  }

  p = q;
  e = r;
  i = r;
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// plain / Packed packet writer and terminators (rleX_extreme_cpu_encode.h:174-311, :384-603; 8 bit: rle8_extreme_cpu.h
// :976-1059, :203-338).  The 8 bit codecs use the same shapes with S = 1 plus the mode byte in the stream header.

template <int FAM, int S, int AL>
struct RunWriter
{
  using TR = Traits<FAM, S, AL>;

  // returns the stored count field for a run of `count` bytes
  static __device__ __forceinline__ uint32_t stored(uint32_t count)
  {
    if constexpr (TR::kAligned) return count / (uint32_t)S - TR::SHORT / (uint32_t)S + 1u;
    else return count - TR::SHORT + 1u;
  }

  static __device__ __forceinline__ void put_run(Sink &s, u32x4 sym, bool same, uint32_t count, uint32_t range, bool longForm)
  {
    const uint32_t c = stored(count);

    if constexpr (!TR::kPacked)
    {
      s.template put_sym<S>(sym);
      if (c <= 255u) s.put8(c); else { s.put8(0); s.put32(c); }
    }
    else
    {
      const uint32_t sm = same ? 0x80u : 0u;
      if (c <= 127u) s.put8(c | sm); else { s.put8(sm); s.put32(c); }
      if (!same) s.template put_sym<S>(sym);
    }

    if constexpr (TR::kRange7)
    {
      if (!longForm) s.put8((range << 1) & 0xFFu); else s.put32((range << 1) | 1u);
    }
    else
    {
      if (!longForm) s.put8(range); else { s.put8(0); s.put32(range); }
    }
  }

  static __device__ __forceinline__ void put_term_head(Sink &s)
  {
    if constexpr (!TR::kPacked) { for (int k = 0; k < S; k++) s.put8(0); s.put8(0); s.put32(0); }
    else { s.put8(0x80); s.put32(0); }
  }

  // plainEnd: the 128 bit encoder always writes `00, u32 0` (A.5 q11)
  static __device__ __forceinline__ void put_term_end(Sink &s, bool plainEnd)
  {
    put_term_head(s);
    if (TR::kRange7 && !plainEnd) s.put32(1); else { s.put8(0); s.put32(0); }
  }

  static __device__ __forceinline__ void put_term_literals(Sink &s, const uint8_t *lit, uint32_t k)
  {
    put_term_head(s);
    if constexpr (TR::kRange7) s.put32(((k + 1u) << 1) | 1u); else { s.put8(0); s.put32(k + 1u); }
    s.putn(lit, k);
  }

  // 0 = literals, 1 = short-range packet, 2 = long-range packet
  static __device__ __forceinline__ int decide(bool same, uint32_t count, uint32_t range)
  {
    bool shortOk;
    if constexpr (!TR::kPacked) shortOk = range <= TR::MAXRANGE && count >= TR::SHORT;
    else shortOk = range <= TR::MAXRANGE && ((count >= TR::SHORT && same) || count >= TR::MEDIUM);
    if (shortOk) return 1;
    if (count >= TR::LONG) return 2;
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// per-block encoders.  d = block start, n = block length, s = sink positioned at the slot start.  Return stream size.

// plain / Packed for S in {1,2,3,4,6,8}
template <int FAM, int S, int AL>
__device__ __forceinline__ uint32_t encode_block_multi(const uint8_t *d, uint32_t n, Sink &s)
{
  using TR = Traits<FAM, S, AL>;
  using RW = RunWriter<FAM, S, AL>;

  s.put32(n);
  s.put32(0);
  if constexpr (S == 1) s.put8(0); // mode = multi

  uint32_t lastRLE = 0, i = 0, p = 0, e = 0;
  u32x4 last = u32x4{ 0, 0, 0, 0 };
  bool ended = false;

  for (;;)
  {
    bool have;
    if constexpr (S == 1) have = next_run8(d, n, i, p, e);
    else have = next_runS<S, TR::kAligned>(d, n, i, p, e);
    if (!have)
      break;

    const u32x4 sym = load_sym<S>(d + p);
    const uint32_t count = e - p;
    const uint32_t range = p - lastRLE + 1u;
    bool same = false;
    int k;

    if constexpr (S == 1 && TR::kPacked)
    {
      // body / tail split of the canonical AVX2 encoder
      const int32_t kk = (int32_t)(e - p - 1u) / 32;
      const bool body = (e < n) && ((int32_t)p + 1 + 32 * kk < (int32_t)n - 32);

      if (body)
      {
        same = sym.x == last.x;
        const bool emit = count >= 11u || (range <= 127u && ((same && count >= 3u) || count >= 4u));
        k = emit ? (range <= 127u ? 1 : 2) : 0;
        if (emit) last = sym;
      }
      else
      {
        k = (count >= 11u) ? (range <= 127u ? 1 : 2) : 0;
      }
    }
    else if constexpr (S == 1)
    {
      k = (count >= 6u) ? (range <= 255u ? 1 : 2) : 0; // rle8_extreme_cpu.h:974
    }
    else
    {
      if constexpr (TR::kPacked) same = sym_eq(sym, last);
      k = RW::decide(same, count, range);
      if (TR::kPacked && k) last = sym;
    }

    if (!k)
      continue;

    RW::put_run(s, sym, same, count, range, k == 2);
    s.putn(d + lastRLE, p - lastRLE);
    lastRLE = e;

    if (e >= n)
    {
      RW::put_term_end(s, false);
      ended = true;
    }
  }

  if (!ended)
    RW::put_term_literals(s, d + lastRLE, n - lastRLE);

  s.patch32(4, s.at);
  return s.at;
}

// 128 bit: literal restatement (rle128_extreme_cpu.h:32-497)
template <int FAM, int AL>
__device__ __forceinline__ uint32_t encode_block_128(const uint8_t *d, uint32_t n32, Sink &s)
{
  using TR = Traits<FAM, 16, AL>;
  using RW = RunWriter<FAM, 16, AL>;

  s.put32(n32);
  s.put32(0);

  const int32_t n = (int32_t)n32;
  int32_t i = 0, lastRLE = 0, count = 0;
  u32x4 symbol = (n >= 16) ? ld128(d) : u32x4{ 0, 0, 0, 0 }; // NOT inverted (q4)
  u32x4 last = u32x4{ 0, 0, 0, 0 };

  auto judge = [&](bool final) -> bool {
    const uint32_t range = (uint32_t)(i - lastRLE - count + 1);
    const bool same = TR::kPacked ? sym_eq(symbol, last) : false;
    const int k = RW::decide(same, (uint32_t)count, range);
    if (k)
    {
      if (TR::kPacked) last = symbol;
      RW::put_run(s, symbol, same, (uint32_t)count, range, k == 2);
      s.putn(d + lastRLE, (uint32_t)(i - count - lastRLE));
      if (!final) lastRLE = i;
    }
    return k != 0;
  };

  while (i < n)
  {
    bool restart = true;

    while (restart)
    {
      restart = false;

      while (i < n - 16)
      {
        const u32x4 x = ld128(d + i);
        const uint32_t z0 = x.x ^ symbol.x, z1 = x.y ^ symbol.y, z2 = x.z ^ symbol.z, z3 = x.w ^ symbol.w;

        if ((z0 | z1 | z2 | z3) == 0)
        {
          count += 16;
          i += 16;
        }
        else
        {
          if constexpr (!TR::kAligned)
          {
            int32_t off;
            if (z0) off = (int32_t)(__builtin_ctz(z0) >> 3);
            else if (z1) off = 4 + (int32_t)(__builtin_ctz(z1) >> 3);
            else if (z2) off = 8 + (int32_t)(__builtin_ctz(z2) >> 3);
            else off = 12 + (int32_t)(__builtin_ctz(z3) >> 3);
            i += off;
            count += off;
          }
          break;
        }
      }

      judge(false);

      while (i < n - 32)
      {
        const u32x4 a = ld128(d + i), b = ld128(d + i + 16);
        const uint32_t z0 = a.x ^ b.x, z1 = a.y ^ b.y, z2 = a.z ^ b.z, z3 = a.w ^ b.w;

        if ((z0 | z1 | z2 | z3) == 0)
        {
          symbol = a;
          i += 32;
          count = 32;
          restart = true;
          break;
        }
        else if (z3 >> 24)
        {
          i += 16;
        }
        else
        {
          int32_t hb; // highest mismatching byte
          if (z3) hb = 12 + ((31 - (int32_t)__builtin_clz(z3)) >> 3);
          else if (z2) hb = 8 + ((31 - (int32_t)__builtin_clz(z2)) >> 3);
          else if (z1) hb = 4 + ((31 - (int32_t)__builtin_clz(z1)) >> 3);
          else hb = (31 - (int32_t)__builtin_clz(z0)) >> 3;
          i += hb + 1;
        }
      }
    }

    // scalar step; bytes >= n never match
    symbol = (i + 16 <= n) ? ld128(d + i) : u32x4{ 0, 0, 0, 0 };

    if (i + 32 <= n && sym_eq(symbol, ld128(d + i + 16)))
    {
      count = 32;
      i += 32;
    }
    else
    {
      count = 0;
      i += 1;
    }
  }

  if (judge(true))
    RW::put_term_end(s, true);
  else
    RW::put_term_literals(s, d + lastRLE, (uint32_t)(i - lastRLE));

  s.patch32(4, s.at);
  return s.at;
}

// One CHUNK of a monolithic 128 bit stream (hsrle_mono_encode.hip.h: the input is cut behind runs of >= LONG bytes, which every state of
// the encoder stores).  d = the chunk's first byte, nChunk its length, nTrue the bytes from there to the END OF THE INPUT (the loops'
// bounds look at the true end: A.5 q4).  The first chunk starts as the stream does (the first 16 bytes match themselves); every other one
// starts where the encoder stands behind a stored run: at the pair search, lastRLE = the chunk's start, lastSymbol = the boundary run's
// symbol (16 bytes at lastSymAt).  A chunk that is not the last ends with its boundary run's packet: no terminator.  No stream header.
template <int FAM, int AL>
__device__ __forceinline__ uint32_t encode_chunk_128(const uint8_t *d, uint32_t nChunk, uint32_t nTrue32, bool first, const uint8_t *lastSymAt, Sink &s)
{
  using TR = Traits<FAM, 16, AL>;
  using RW = RunWriter<FAM, 16, AL>;

  const int32_t n = (int32_t)nTrue32, stopAt = (int32_t)nChunk;
  const bool finalChunk = nChunk == nTrue32;
  int32_t i = 0, lastRLE = 0, count = 0;
  u32x4 symbol = (first && n >= 16) ? ld128(d) : u32x4{ 0, 0, 0, 0 };
  u32x4 last = first ? u32x4{ 0, 0, 0, 0 } : ld128(lastSymAt);
  bool skipExtend = !first;

  auto judge = [&](bool final) -> bool {
    const uint32_t range = (uint32_t)(i - lastRLE - count + 1);
    const bool same = TR::kPacked ? sym_eq(symbol, last) : false;
    const int k = RW::decide(same, (uint32_t)count, range);
    if (k)
    {
      if (TR::kPacked) last = symbol;
      RW::put_run(s, symbol, same, (uint32_t)count, range, k == 2);
      s.putn(d + lastRLE, (uint32_t)(i - count - lastRLE));
      if (!final) lastRLE = i;
    }
    return k != 0;
  };

  while (i < n)
  {
    bool restart = true;

    while (restart)
    {
      restart = false;

      if (!skipExtend)
      {
        while (i < n - 16)
        {
          const u32x4 x = ld128(d + i);
          const uint32_t z0 = x.x ^ symbol.x, z1 = x.y ^ symbol.y, z2 = x.z ^ symbol.z, z3 = x.w ^ symbol.w;

          if ((z0 | z1 | z2 | z3) == 0)
          {
            count += 16;
            i += 16;
          }
          else
          {
            if constexpr (!TR::kAligned)
            {
              int32_t off;
              if (z0) off = (int32_t)(__builtin_ctz(z0) >> 3);
              else if (z1) off = 4 + (int32_t)(__builtin_ctz(z1) >> 3);
              else if (z2) off = 8 + (int32_t)(__builtin_ctz(z2) >> 3);
              else off = 12 + (int32_t)(__builtin_ctz(z3) >> 3);
              i += off;
              count += off;
            }
            break;
          }
        }

        const bool stored = judge(false);
        if (!finalChunk && i >= stopAt)
          return stored && i == stopAt ? s.at : 0u;                     // the boundary run's packet ends the chunk (anything else: the cut was wrong)
      }
      skipExtend = false;

      while (i < n - 32)
      {
        const u32x4 a = ld128(d + i), b = ld128(d + i + 16);
        const uint32_t z0 = a.x ^ b.x, z1 = a.y ^ b.y, z2 = a.z ^ b.z, z3 = a.w ^ b.w;

        if ((z0 | z1 | z2 | z3) == 0)
        {
          symbol = a;
          i += 32;
          count = 32;
          restart = true;
          break;
        }
        else if (z3 >> 24)
        {
          i += 16;
        }
        else
        {
          int32_t hb; // highest mismatching byte
          if (z3) hb = 12 + ((31 - (int32_t)__builtin_clz(z3)) >> 3);
          else if (z2) hb = 8 + ((31 - (int32_t)__builtin_clz(z2)) >> 3);
          else if (z1) hb = 4 + ((31 - (int32_t)__builtin_clz(z1)) >> 3);
          else hb = (31 - (int32_t)__builtin_clz(z0)) >> 3;
          i += hb + 1;
        }
      }
    }

    // scalar step; bytes >= n never match
    symbol = (i + 16 <= n) ? ld128(d + i) : u32x4{ 0, 0, 0, 0 };

    if (i + 32 <= n && sym_eq(symbol, ld128(d + i + 16)))
    {
      count = 32;
      i += 32;
    }
    else
    {
      count = 0;
      i += 1;
    }
  }

  if (!finalChunk)
    return 0u;                                                          // (a chunk in front of the last one ends above)
  if (judge(true))
    RW::put_term_end(s, true);
  else
    RW::put_term_literals(s, d + lastRLE, (uint32_t)(i - lastRLE));
  return s.at;
}

// 3 / 7 symbol LUT (rleX_Xsl.h:114-264 process_symbol)
template <int FAM, int S, int AL>
__device__ __forceinline__ uint32_t encode_block_lut(const uint8_t *d, uint32_t n, Sink &s)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int K = TR::K;
  constexpr uint32_t RB = TR::RB, MAXC = 127u, MAXR = (1u << RB) - 1u;

  s.put32(n);
  s.put32(0);

  u32x4 lut[K];
  {
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const uint32_t b4 = init[k] * 0x01010101u;
      u32x4 v = u32x4{ b4, b4, b4, b4 };
      if constexpr (S == 1) v = u32x4{ v.x & 0xFFu, 0, 0, 0 };
      else if constexpr (S == 2) v = u32x4{ v.x & 0xFFFFu, 0, 0, 0 };
      else if constexpr (S == 3) v = u32x4{ v.x & 0xFFFFFFu, 0, 0, 0 };
      else if constexpr (S == 4) v = u32x4{ v.x, 0, 0, 0 };
      else if constexpr (S == 6) v = u32x4{ v.x, v.y & 0xFFFFu, 0, 0 };
      else v = u32x4{ v.x, v.y, 0, 0 };
      lut[k] = v;
    }
  }

  uint32_t lastRLE = 0, i = 0, p = 0, e = 0;
  bool ended = false;

  for (;;)
  {
    bool have;
    if constexpr (S == 1) have = next_run8(d, n, i, p, e);
    else have = next_runS<S, TR::kAligned>(d, n, i, p, e);
    if (!have)
      break;

    const u32x4 sym = load_sym<S>(d + p);
    const uint32_t count = e - p;
    const uint32_t range = p - lastRLE + 2u;

    uint32_t m = K;
#pragma unroll
    for (int k = K - 1; k >= 0; k--)
      if (sym_eq(lut[k], sym)) m = (uint32_t)k;

    const uint32_t c = TR::kAligned ? (count / (uint32_t)S - 3u / (uint32_t)S + 2u) : (count - 3u + 2u);

    // the penalty uses 0xFFFFF where the writer uses 0xFFFF (A.5 q3; rleX_Xsl.h:130 vs :195)
    uint32_t pen = (range <= 0xFFFFFu) ? (range <= MAXR ? 0u : 2u) : 4u;
    pen += (c <= 0xFFFFFu) ? (c <= MAXC ? 0u : 2u) : 4u;
    pen += (m == (uint32_t)K) ? 1u : 0u;

    if (!(count >= (uint32_t)S + 10u || count >= 3u + pen))
      continue;

    // move to front (rleX_Xsl.h:134-188)
    {
      const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
      for (int k = K - 1; k >= 1; k--)
        if ((uint32_t)k <= limit) lut[k] = lut[k - 1];
      lut[0] = sym;
    }

    const uint32_t c7 = (c <= MAXC) ? c : (c <= 0xFFFFu ? 1u : 0u);
    const uint32_t r7 = (range <= MAXR) ? range : (range <= 0xFFFFu ? 1u : 0u);

    s.put16((m << (K == 3 ? 14 : 13)) | (c7 << RB) | r7);
    if (m == (uint32_t)K) s.template put_sym<S>(sym);
    if (c != c7) { if (c <= 0xFFFFu) s.put16(c); else s.put32(c); }
    if (range != r7) { if (range <= 0xFFFFu) s.put16(range); else s.put32(range); }

    s.putn(d + lastRLE, p - lastRLE);
    lastRLE = e;

    if (e >= n)
    {
      s.put16((1u << RB) | 1u);
      s.put16(0);
      s.put16(0);
      ended = true;
    }
  }

  if (!ended)
  {
    const uint32_t k = n - lastRLE;
    s.put16(1u << RB);
    s.put16(0);
    s.put32(k + 2u);
    s.putn(d + lastRLE, k);
  }

  s.patch32(4, s.at);
  return s.at;
}

// ------------------------------------------------------------------------------------------------------------------
// 8 bit Single: literal restatement

struct Cmp16
{
  uint32_t m0, m1, m2, m3; // 0x80 in every byte that EQUALS the symbol
  __device__ __forceinline__ Cmp16(const uint8_t *p, uint32_t bs)
  {
    const u32x4 x = ld128(p);
    m0 = zero_bytes(x.x ^ bs); m1 = zero_bytes(x.y ^ bs); m2 = zero_bytes(x.z ^ bs); m3 = zero_bytes(x.w ^ bs);
  }
  __device__ __forceinline__ bool all() const { return (m0 & m1 & m2 & m3) == 0x80808080u; }
  __device__ __forceinline__ bool any() const { return (m0 | m1 | m2 | m3) != 0; }
  __device__ __forceinline__ uint32_t pop() const { return (uint32_t)(__builtin_popcount(m0) + __builtin_popcount(m1) + __builtin_popcount(m2) + __builtin_popcount(m3)); }
  __device__ __forceinline__ bool lastByte() const { return (m3 >> 31) != 0; }
  // number of leading bytes that equal the symbol (caller guarantees not all)
  __device__ __forceinline__ uint32_t leading() const
  {
    const uint32_t n0 = ~m0 & 0x80808080u, n1 = ~m1 & 0x80808080u, n2 = ~m2 & 0x80808080u, n3 = ~m3 & 0x80808080u;
    if (n0) return first_set_byte(n0);
    if (n1) return 4u + first_set_byte(n1);
    if (n2) return 8u + first_set_byte(n2);
    return 12u + first_set_byte(n3);
  }
  // index of the first byte that equals the symbol (caller guarantees any)
  __device__ __forceinline__ uint32_t first() const
  {
    if (m0) return first_set_byte(m0);
    if (m1) return 4u + first_set_byte(m1);
    if (m2) return 8u + first_set_byte(m2);
    return 12u + first_set_byte(m3);
  }
};

// symbol pick: rle8_extreme_cpu.c:53-153 (the sse2 estimator on every ISA, A.5 q9).  All arithmetic modulo 2^32.
__device__ inline uint32_t single_pick_symbol(const uint8_t *d, uint32_t n32, uint32_t *prob, uint32_t *pc)
{
  for (int k = 0; k < 256; k++) { prob[k] = 0; pc[k] = 0; }

  const int32_t n = (int32_t)n32;
  if (d[0] != 0)
    pc[0] = 0xFFFFFFFFu;

  int32_t i = 0;
  const int32_t end = n - 16;
  uint32_t last = (~(uint32_t)d[0]) & 0xFFu;
  uint32_t count = 0;

  while (i < end)
  {
    const Cmp16 c(d + i, last * 0x01010101u);

    if (c.all())
    {
      count += 15u; // sic
      i += 15;
    }
    else
    {
      if (c.any() || count > 1u)
      {
        const uint32_t z = c.leading();
        count += z;
        i += (int32_t)z;
        prob[last] += count;
        pc[last]++;
      }

      while (i < end)
      {
        // first k in 0..14 with d[i+k] == d[i+k+1]
        const u32x4 x = ld128(d + i);
        const uint32_t y3 = (x.w >> 8) | (~x.w & 0xFF000000u);
        const uint32_t q0 = zero_bytes(x.x ^ alignbyte(x.y, x.x, 1)), q1 = zero_bytes(x.y ^ alignbyte(x.z, x.y, 1));
        const uint32_t q2 = zero_bytes(x.z ^ alignbyte(x.w, x.z, 1)), q3 = zero_bytes(x.w ^ y3);

        if ((q0 | q1 | q2 | q3) == 0)
          i += 15;
        else
        {
          if (q0) i += (int32_t)first_set_byte(q0);
          else if (q1) i += 4 + (int32_t)first_set_byte(q1);
          else if (q2) i += 8 + (int32_t)first_set_byte(q2);
          else i += 12 + (int32_t)first_set_byte(q3);
          break;
        }
      }

      count = 1;
      last = d[i];
    }

    i++;
  }

  prob[last] += count;
  pc[last]++;

  uint32_t best = 0, bestSym = 0;

  for (uint32_t sy = 0; sy < 256u; sy++)
  {
    if (pc[sy] > 0 && prob[sy] / pc[sy] > 2u)
    {
      const uint32_t saved = prob[sy] - pc[sy] * 2u;
      if (saved > best) { best = saved; bestSym = sy; }
    }
  }

  return bestSym;
}

template <bool PACKEDSINGLE>
__device__ inline uint32_t encode_block_single(const uint8_t *d, uint32_t n32, Sink &s)
{
  constexpr int32_t SHORT = PACKEDSINGLE ? 2 : 4;
  constexpr int32_t MEDIUM = 6;
  constexpr int32_t LONG = PACKEDSINGLE ? 10 : 8;

  uint32_t prob[256], pc[256]; // per-lane histograms (private segment)
  const uint32_t sym = single_pick_symbol(d, n32, prob, pc);
  const uint32_t bs = sym * 0x01010101u;

  s.put32(n32);
  s.put32(0);
  s.put8(1); // mode = single
  s.put8(sym);

  const int32_t n = (int32_t)n32;
  const int32_t end = n - 16;
  int32_t i = 0, count = 0, lastRLE = 0, wasted = 0, firstW = 0;

  auto put_count = [&](int32_t cnt) {
    const uint32_t c = (uint32_t)(cnt - SHORT + 1);
    if (c <= 255u) s.put8(c); else { s.put8(0); s.put32(c); }
  };
  auto emit_short = [&](int32_t range) {
    put_count(count);
    s.put8((uint32_t)range);
    s.putn(d + lastRLE, (uint32_t)(i - count - lastRLE));
    lastRLE = i;
  };
  auto emit_long = [&](int32_t range) {
    put_count(count);
    s.put8(0);
    s.put32((uint32_t)range);
    s.putn(d + lastRLE, (uint32_t)(i - count - lastRLE));
    lastRLE = i;
  };

  while (i < end)
  {
    const Cmp16 c(d + i, bs);

    if (c.all())
    {
      count += 16;
      i += 15;
    }
    else
    {
      if (c.any() || count > 1)
      {
        const int32_t z = (int32_t)c.leading();
        count += z;
        i += z;

        const int32_t range = i - lastRLE - count + 1;

        if (count >= SHORT)
        {
          if (range <= 255)
          {
            emit_short(range);
            wasted = 0;
          }
          else if (count >= LONG || (PACKEDSINGLE && (count - SHORT + 1 <= 255 && count >= MEDIUM)))
          {
            emit_long(range);
            wasted = 0;
          }
          else
          {
            wasted++;

            if (wasted == 1 || i - firstW > 255)
            {
              firstW = i - count;
              wasted = 1;
            }
            else if (wasted > 2)
            {
              // back-track to the first skipped run and force a long packet (rle8_extreme_cpu.h:1244-1285)
              i = firstW;
              wasted = 0;
              count = 0;

              while (i < end && d[i] == sym)
              {
                count++;
                i++;
              }

              s.put8((uint32_t)(count - SHORT + 1) & 0xFFu);
              s.put8(0);
              s.put32((uint32_t)(i - lastRLE - count + 1));
              s.putn(d + lastRLE, (uint32_t)(i - count - lastRLE));
              lastRLE = i;
            }
          }
        }
      }

      count = 0;

      while (i < end)
      {
        const Cmp16 b(d + i, bs);

        if (!b.any() || (!b.lastByte() && b.pop() < (uint32_t)SHORT))
          i += 16;
        else
        {
          i += (int32_t)b.first();
          count = 1;
          break;
        }
      }
    }

    i++;
  }

  for (; i < n; i++)
  {
    if (d[i] == sym)
      count++;
    else
    {
      const int32_t range = i - lastRLE - count + 1;
      if (range <= 255 && count >= SHORT) emit_short(range);
      else if (count >= LONG) emit_long(range);
      count = 0;
    }
  }

  {
    const int32_t range = i - lastRLE - count + 1;

    if (range <= 255 && count >= SHORT)
    {
      emit_short(range);
      s.put8(0); s.put32(0); s.put8(0); s.put32(0);
    }
    else if (count >= LONG)
    {
      emit_long(range);
      s.put8(0); s.put32(0); s.put8(0); s.put32(0);
    }
    else
    {
      s.put8(0); s.put32(0); s.put8(0); s.put32((uint32_t)(range + count));
      s.putn(d + lastRLE, (uint32_t)(i - lastRLE));
    }
  }

  s.patch32(4, s.at);
  return s.at;
}

// ------------------------------------------------------------------------------------------------------------------

template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encode_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                      uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes)
{
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x) * 64u + threadIdx.x;   // XCD-aware tile order (hsrle_common.hip.h)
  if (b >= nBlocks)
    return;

  const uint64_t start = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  const uint8_t *d = in + start;
  Sink s{ slots + (uint64_t)b * slotStride, 0u, in + U };
  uint32_t size;

  if constexpr (FAM == SINGLE || FAM == PACKED_SINGLE)
    size = encode_block_single<FAM == PACKED_SINGLE>(d, n, s);
  else if constexpr (FAM == LUT3 || FAM == LUT7)
    size = encode_block_lut<FAM, S, AL>(d, n, s);
  else if constexpr (S == 16)
    size = encode_block_128<FAM, AL>(d, n, s);
  else
    size = encode_block_multi<FAM, S, AL>(d, n, s);

  sizes[b] = size;
}

// One CHUNK of a monolithic 8 bit Single stream (hsrle_mono_encode.hip.h).  Behind ANY stored run the scanner's state is its position alone
// (lastRLE = i, wasted = 0, count = 0, then the search: rle8_extreme_cpu.h:1140-1312), and a run of >= LONG bytes of the symbol is stored whatever
// the state and found from its first byte whatever the scanner's phase -- so the input is cut behind those, and a chunk is the block loop
// started at its first byte (which is not the symbol: the loop's first trip falls through to the search).  `end` is the TRUE end - 16; the
// scalar tail and the terminator belong to the last chunk.  No stream header; a chunk in front of the last ends with its boundary run's packet.
// Long literal stretches are not copied by the chunk's one lane (a Single stream of data its symbol is rare in is 99.8 % literals: 19 KB per
// chunk on the run-distributed buffer, 43 GiB/s): the lane NOTES them -- source, destination, length, appended to a list with one atomic --
// and k_copy_jobs copies them with whole waves afterwards.  (The list is sized for every possible stretch of >= kCopyJobMin bytes; a lane
// that finds it full copies itself.)
constexpr uint32_t kCopyJobMin = 1024u;
struct CopyJobs
{
  uint64_t *list;            // 3 words per job: source offset in the input, destination offset in the staging area, bytes
  uint32_t *count;
  uint32_t cap;
  uint64_t srcBase, dstBase; // offsets of the chunk's first input byte / of its staging slot
};

template <bool PACKEDSINGLE>
__device__ inline uint32_t encode_chunk_single(const uint8_t *d, uint32_t nChunk, uint32_t nTrue32, uint32_t sym, Sink &s, const CopyJobs &jobs)
{
  auto put_literals = [&](int32_t from, uint32_t len) {
    if (len >= kCopyJobMin && jobs.list != nullptr)
    {
      const uint32_t k = atomicAdd(jobs.count, 1u);
      if (k < jobs.cap)
      {
        jobs.list[3ull * k] = jobs.srcBase + (uint64_t)(uint32_t)from;
        jobs.list[3ull * k + 1] = jobs.dstBase + s.at;
        jobs.list[3ull * k + 2] = len;
        s.at += len;
        return;
      }
    }
    s.putn(d + from, len);
  };
  constexpr int32_t SHORT = PACKEDSINGLE ? 2 : 4;
  constexpr int32_t MEDIUM = 6;
  constexpr int32_t LONG = PACKEDSINGLE ? 10 : 8;
  const uint32_t bs = sym * 0x01010101u;
  const int32_t n = (int32_t)nTrue32, stopAt = (int32_t)nChunk;
  const bool finalChunk = nChunk == nTrue32;
  const int32_t end = n - 16;
  int32_t i = 0, count = 0, lastRLE = 0, wasted = 0, firstW = 0;

  auto put_count = [&](int32_t cnt) {
    const uint32_t c = (uint32_t)(cnt - SHORT + 1);
    if (c <= 255u) s.put8(c); else { s.put8(0); s.put32(c); }
  };
  auto emit_short = [&](int32_t range) {
    put_count(count);
    s.put8((uint32_t)range);
    put_literals(lastRLE, (uint32_t)(i - count - lastRLE));
    lastRLE = i;
  };
  auto emit_long = [&](int32_t range) {
    put_count(count);
    s.put8(0);
    s.put32((uint32_t)range);
    put_literals(lastRLE, (uint32_t)(i - count - lastRLE));
    lastRLE = i;
  };

  while (i < end)
  {
    const Cmp16 c(d + i, bs);

    if (c.all())
    {
      count += 16;
      i += 15;
    }
    else
    {
      if (c.any() || count > 1)
      {
        const int32_t z = (int32_t)c.leading();
        count += z;
        i += z;

        const int32_t range = i - lastRLE - count + 1;
        bool stored = false;

        if (count >= SHORT)
        {
          if (range <= 255)
          {
            emit_short(range);
            wasted = 0;
            stored = true;
          }
          else if (count >= LONG || (PACKEDSINGLE && (count - SHORT + 1 <= 255 && count >= MEDIUM)))
          {
            emit_long(range);
            wasted = 0;
            stored = true;
          }
          else
          {
            wasted++;

            if (wasted == 1 || i - firstW > 255)
            {
              firstW = i - count;
              wasted = 1;
            }
            else if (wasted > 2)
            {
              // back-track to the first skipped run and force a long packet (rle8_extreme_cpu.h:1244-1285)
              i = firstW;
              wasted = 0;
              count = 0;

              while (i < end && d[i] == sym)
              {
                count++;
                i++;
              }

              s.put8((uint32_t)(count - SHORT + 1) & 0xFFu);
              s.put8(0);
              s.put32((uint32_t)(i - lastRLE - count + 1));
              put_literals(lastRLE, (uint32_t)(i - count - lastRLE));
              lastRLE = i;
            }
          }
        }
        if (!finalChunk && i >= stopAt)
          return (stored && i == stopAt) ? s.at : 0u;                   // the boundary run's packet ends the chunk (anything else: the cut was wrong)
      }

      count = 0;

      bool streak = false;                                              // the last window was skipped: try four at a time
      while (i < end)
      {
        // four windows per trip while all four are skipped (far from the symbol's runs the loop is one dependent load per window otherwise:
        // the longest chunk's lane sets the kernel's time)
        if (streak && i + 48 < end)
        {
          const Cmp16 b0(d + i, bs), b1(d + i + 16, bs), b2(d + i + 32, bs), b3(d + i + 48, bs);
          const bool s0 = !b0.any() || (!b0.lastByte() && b0.pop() < (uint32_t)SHORT), s1 = !b1.any() || (!b1.lastByte() && b1.pop() < (uint32_t)SHORT);
          const bool s2 = !b2.any() || (!b2.lastByte() && b2.pop() < (uint32_t)SHORT), s3 = !b3.any() || (!b3.lastByte() && b3.pop() < (uint32_t)SHORT);
          if (s0 && s1 && s2 && s3) { i += 64; continue; }
        }
        const Cmp16 b(d + i, bs);

        if (!b.any() || (!b.lastByte() && b.pop() < (uint32_t)SHORT))
        {
          i += 16;
          streak = true;
        }
        else
        {
          i += (int32_t)b.first();
          count = 1;
          break;
        }
      }
    }

    i++;
  }
  if (!finalChunk)
    return 0u;

  for (; i < n; i++)
  {
    if (d[i] == sym)
      count++;
    else
    {
      const int32_t range = i - lastRLE - count + 1;
      if (range <= 255 && count >= SHORT) emit_short(range);
      else if (count >= LONG) emit_long(range);
      count = 0;
    }
  }

  {
    const int32_t range = i - lastRLE - count + 1;

    if (range <= 255 && count >= SHORT)
    {
      emit_short(range);
      s.put8(0); s.put32(0); s.put8(0); s.put32(0);
    }
    else if (count >= LONG)
    {
      emit_long(range);
      s.put8(0); s.put32(0); s.put8(0); s.put32(0);
    }
    else
    {
      s.put8(0); s.put32(0); s.put8(0); s.put32((uint32_t)(range + count));
      put_literals(lastRLE, (uint32_t)(i - lastRLE));
    }
  }
  return s.at;
}

// chunks of ONE monolithic 8 bit Single stream, one lane per chunk (pick[0]: the stream's symbol, k_single_pick_final)
// B != 0 (round 4): chunks of the BLOCKS of a container (split encode, hsrle_capi.hip: compress_split) -- every block is a stream of its own:
// its end is "the end of the input", its first chunk writes the stream header (size patched in at placement: k_split_finish), mode 1 and the
// block's symbol (pick: a byte per block, k_single_pick)
template <bool PACKEDSINGLE>
__global__ __launch_bounds__(64) void k_encode_single_chunks(const uint8_t *__restrict__ in, uint64_t U, uint32_t chunks, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ slotOff,
                                                             uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes, const uint32_t *__restrict__ pick,
                                                             uint64_t *__restrict__ jobList, uint32_t *__restrict__ jobCount, uint32_t jobCap, uint32_t B,
                                                             const uint32_t *__restrict__ chunkCount)
{
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (B != 0u && chunkCount != nullptr) chunks = umin(chunks, chunkCount[0]);   // (split encode: the launch covers the most chunks there can be, the device knows how many there are)
  if (c >= chunks) return;
  const uint64_t start = starts[c];
  Sink s{ slots + slotOff[c], 0u, in + U };
  const CopyJobs jobs{ jobList, jobCount, jobCap, start, slotOff[c] };
  if (B == 0u)
  {
    sizes[c] = encode_chunk_single<PACKEDSINGLE>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(U - start), pick[0] & 0xFFu, s, jobs);
    return;
  }
  const uint64_t blk = start / B, blockEnd = ((blk + 1ull) * B < U) ? (blk + 1ull) * B : U;
  const uint32_t sym = ((const uint8_t *)pick)[blk];
  if (start == blk * B) { s.put32((uint32_t)(blockEnd - start)); s.put32(0); s.put8(1); s.put8(sym); }
  sizes[c] = encode_chunk_single<PACKEDSINGLE>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(blockEnd - start), sym, s, jobs);
}

// the noted literal stretches of the chunk encoders: persistent waves, one job at a time, destination-aligned 16-byte stores
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_copy_jobs(const uint8_t *__restrict__ in, uint8_t *__restrict__ slots, const uint64_t *__restrict__ jobList, const uint32_t *__restrict__ jobCount,
                                                   uint32_t jobCap)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = jobCount[0] < jobCap ? jobCount[0] : jobCap;
  for (uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6); j < n; j += gridDim.x * 4u)
  {
    const uint8_t *src = in + jobList[3ull * j];
    uint8_t *dst = slots + jobList[3ull * j + 1];
    const uint64_t size = jobList[3ull * j + 2];
    uint64_t head = (16u - ((uintptr_t)dst & 15u)) & 15u;
    if (head > size) head = size;
    if (lane < head) dst[lane] = src[lane];
    const uint64_t body = (size - head) & ~15ull;
    for (uint64_t k = (uint64_t)lane * 16u; k < body; k += 4096u)
    {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) if (k + 1024u * u < body) v[u] = ld128(src + head + k + 1024u * u);
#pragma unroll
      for (int u = 0; u < 4; u++) if (k + 1024u * u < body) st128(dst + head + k + 1024u * u, v[u]);
    }
    const uint64_t tail = size - head - body;
    if (lane < tail) dst[head + body + lane] = src[head + body + lane];
  }
}

// chunks of ONE monolithic 128 bit stream, one lane per chunk (hsrle_mono_encode.hip.h; syms[c] = where the boundary run in front of chunk c starts)
// B != 0 (round 4): chunks of the blocks of a container, as k_encode_single_chunks
template <int FAM, int AL>
__global__ __launch_bounds__(64) void k_encode128_chunks(const uint8_t *__restrict__ in, uint64_t U, uint32_t chunks, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ syms,
                                                         const uint64_t *__restrict__ slotOff, uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes, uint32_t B,
                                                         const uint32_t *__restrict__ chunkCount)
{
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (B != 0u && chunkCount != nullptr) chunks = umin(chunks, chunkCount[0]);   // (split encode: see k_encode_single_chunks)
  if (c >= chunks) return;
  const uint64_t start = starts[c];
  Sink s{ slots + slotOff[c], 0u, in + U };
  if (B == 0u)
  {
    sizes[c] = encode_chunk_128<FAM, AL>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(U - start), c == 0u, in + syms[c], s);
    return;
  }
  const uint64_t blk = start / B, blockEnd = ((blk + 1ull) * B < U) ? (blk + 1ull) * B : U;
  const bool opens = start == blk * B;
  if (opens) { s.put32((uint32_t)(blockEnd - start)); s.put32(0); }
  sizes[c] = encode_chunk_128<FAM, AL>(in + start, (uint32_t)(starts[c + 1u] - start), (uint32_t)(blockEnd - start), opens, in + syms[c], s);
}

} // namespace hsrle
