// hsrle_mono_encode.hip.h -- ONE monolithic reference stream encoded by many lanes (the encode side of SURVEY.md 8f-3 / 7 step 7).
//
// The reference's encoders are sequential state machines: whether a run is stored depends on where the last stored run ended
// (`lastRLE`: the range field) and, for the Packed codecs, on its symbol (reference: src/rle8_extreme_cpu.h:974-1001,
// src/rleX_extreme_cpu_encode.h:174-311).  But a run of LONG bytes or more is stored WHATEVER the state is, and
// behind it the state is fully known: lastRLE = its end, lastSymbol = its symbol.  So the input can be cut behind such runs and every
// piece encoded on its own -- by the block kernels in their MONO mode (hsrle_encode8.hip.h) -- into exactly the packets the sequential
// encoder writes for it.  Codecs with a move-to-front list: the list in front of a chunk is guessed, checked and repaired (below).
//
//   k_mono_cuts     one lane per nominal piece of G input bytes: the first maximal run of >= LONG equal symbols that ENDS inside the
//                   piece (the lane looks LONG bytes back, so a run that began in an earlier piece is seen long enough)
//   k_mono_scatter  the pieces that found one, compacted: chunk starts, the symbol in front of every chunk, staging slot offsets
//   k_encode*_blocks<.., MONO>   one lane per chunk
//   k_compact_var_t a wave (or, for the small chunks of a container's blocks, a quarter wave) per chunk: staging slot -> its place behind the 9 / 8 byte stream header
//   k_mono_finish   the stream header
// Data without long runs (random bytes) gives few chunks or one: then one lane walks it all, as the block kernel would a huge block.
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

constexpr uint64_t MONO_NO_CUT = ~0ull;

// 8 bit symbols.  cutPos[c] = end (exclusive) of the first maximal run of >= LONGC equal bytes with c * G < end <= (c + 1) * G and
// end < U, cutSym[c] its byte; MONO_NO_CUT if there is none.  Piece 0 additionally starts the stream: the host adds position 0.
// onlySym != nullptr (the Single codecs): only runs of that byte are cuts.
__global__ __launch_bounds__(64) void k_mono_cuts8(const uint8_t *__restrict__ in, uint64_t U, uint32_t G, uint32_t pieces, uint32_t LONGC, uint64_t *__restrict__ cutPos,
                                                   uint64_t *__restrict__ cutSym, uint32_t *__restrict__ flags, const uint32_t *__restrict__ onlySym = nullptr,
                                                   uint32_t blockB = 0)
{
  // blockB != 0 ("split encode" of a container, G divides blockB): every block is a stream of its own -- the scan does not look in front of
  // the piece's block, and "the end of the input" is the end of that block
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= pieces) return;
  const uint64_t x = (uint64_t)c * G;
  // (split encode: onlySym is a BYTE per block -- every block of a Single container has its own symbol)
  const uint32_t want = onlySym == nullptr ? 0x100u : (blockB ? (uint32_t)((const uint8_t *)onlySym)[x / blockB] : (onlySym[0] & 0xFFu));   // 0x100: any symbol
  if (x >= U) { cutPos[c] = MONO_NO_CUT; cutSym[c] = 0; flags[c] = 0u; return; }   // (split encode: a piece behind the last, short block)
  const uint64_t lo = blockB ? (x / blockB) * blockB : 0ull;
  if (blockB) { const uint64_t be = lo + blockB; if (be < U) U = be; }
  const uint64_t hiEnd = (x + G < U) ? x + G : U;
  uint64_t i = (x > lo + LONGC) ? x - LONGC : lo;
  uint64_t st = i;
  uint32_t sy = in[i];
  uint64_t found = MONO_NO_CUT;
  uint32_t fsym = 0;
  i++;
  // a run [st, i) ends when in[i] differs; it is a cut iff it is long enough and ends inside (x, hiEnd] (and in front of the input's end).
  // Eight positions per trip from two 8-byte loads: w1 byte k = in[i + k], w byte k = its left neighbour
  const uint64_t lastCut = (U > 64u) ? U - 64u : 0u;
  const uint64_t stop = (hiEnd < lastCut) ? hiEnd : lastCut;             // run ends behind `stop` are of no use
  while (i <= stop)
  {
    if (want != 0x100u && sy != want && i + 32u <= U && i + 32u <= stop)
    {
      // Single codecs, 32 bytes at a time (two loads in flight: the loop is one memory latency per trip): no byte of the symbol, no cut
      const u32x4 a = ld128(in + i), b = ld128(in + i + 16u);
      const uint32_t ws = want * 0x01010101u;
      if ((zero_bytes(a.x ^ ws) | zero_bytes(a.y ^ ws) | zero_bytes(a.z ^ ws) | zero_bytes(a.w ^ ws) | zero_bytes(b.x ^ ws) | zero_bytes(b.y ^ ws) | zero_bytes(b.z ^ ws) | zero_bytes(b.w ^ ws)) == 0u)
      {
        st = i + 31u; sy = b.w >> 24;
        i += 32u;
        continue;
      }
    }
    if (i + 9u <= U)
    {
      const uint64_t w = ld64(in + i - 1), w1 = ld64(in + i);
      if (want != 0x100u && sy != want)
      {
        // Single codecs: eight bytes without the symbol (and no run of it open) hold no cut and leave nothing to remember but the last byte --
        // on data its symbol is rare in this is nearly every trip (64 MiB run-distributed: the cut finder 294 -> 209 us; with the 32-byte trips above R4CUTS)
        const uint64_t t = w1 ^ ((uint64_t)want * 0x0101010101010101ull);
        if (((((t & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | t) & 0x8080808080808080ull) == 0x8080808080808080ull)
        {
          st = i + 7u; sy = (uint32_t)(w1 >> 56);
          i += 8u;
          continue;
        }
      }
      const uint64_t d = w ^ w1;                                        // byte k: in[i - 1 + k] ^ in[i + k]
      const uint64_t z = ((d & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | d;
      const uint64_t eq = ~z & 0x8080808080808080ull;                   // 0x80 in byte k: in[i + k] continues the run of its left neighbour
      if (eq == 0x8080808080808080ull) { i += 8u; continue; }           // the open run goes on through all eight
      // positions where a run ends = bytes that do NOT continue: visit them in order (all from registers)
      uint64_t brk = ~eq & 0x8080808080808080ull;
      bool hit = false;
      while (brk != 0ull)
      {
        const uint32_t k = (uint32_t)__builtin_ctzll(brk) >> 3;
        brk &= brk - 1ull;
        const uint64_t at = i + k;
        if (at > stop) break;
        if (at - st >= LONGC && at > x && (want == 0x100u || sy == want)) { found = at; fsym = sy; hit = true; break; }
        st = at; sy = (uint32_t)(w1 >> (8u * k)) & 0xFFu;
      }
      if (hit) break;
      i += 8u;
      continue;
    }
    const uint32_t v = in[i];
    if (v != sy)
    {
      if (i - st >= LONGC && i > x && (want == 0x100u || sy == want)) { found = i; fsym = sy; break; }
      st = i; sy = v;
    }
    i++;
  }
  cutPos[c] = found;
  cutSym[c] = (uint64_t)fsym;
  flags[c] = (found != MONO_NO_CUT) ? 1u : 0u;
}

// Symbols of S = 2, 3, 4, 6, 8 bytes (reference: src/rleX_extreme_cpu_encode.h:315-366 run search, :79-163 extension; restated in
// hsrle_encodeS.hip.h).  With the match bits m[j] = (d[j] == d[j + S]) a run is a maximal stretch [q, q + L) of set bits with L >= S: it
// starts at the first searched position in it and ends at e = q + S + L (byte-aligned) or q + S * floor((L + S) / S) (sym-aligned).  The
// search resumes at the previous run's end, which may lie up to S - 1 bytes INSIDE the next stretch (then that run starts later, with a
// rotated symbol); so a stretch is a cut only if its start is CLEAN: no stretch of >= S set bits ends within the S positions in front
// of it (then the encoder's search position is at or in front of q when it gets there, and the run is [q, e) with symbol d[q, q + S)).
// cutSym[c] = that symbol (low S bytes).  Stretches that begin in front of the lane's look-back are not used.
// bit j: m[j .. j + S) are all set (bits beyond 31 count as clear)
template <int S>
__device__ __forceinline__ uint32_t stretch_of_s(uint32_t m)
{
  const uint32_t a = m & (m >> 1);                                       // 2
  if constexpr (S == 2) return a;
  else if constexpr (S == 3) return a & (m >> 2);
  else
  {
    const uint32_t b = a & (a >> 2);                                     // 4
    if constexpr (S == 4) return b;
    else if constexpr (S == 6) return b & (a >> 4);
    else
    {
      const uint32_t c = b & (b >> 4);                                   // 8
      if constexpr (S == 8) return c;
      else { static_assert(S == 16, "symbols of 2, 3, 4, 6, 8 or 16 bytes"); return c & (c >> 8); }
    }
  }
}

template <int S, int ALIGNED>
__global__ __launch_bounds__(64) void k_mono_cutsS(const uint8_t *__restrict__ in, uint64_t U, uint32_t G, uint32_t pieces, uint32_t LONGC, uint64_t *__restrict__ cutPos,
                                                   uint64_t *__restrict__ cutSym, uint32_t *__restrict__ flags, uint32_t blockB = 0)
{
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= pieces) return;
  constexpr uint64_t SU = (uint64_t)S;
  const uint64_t x = (uint64_t)c * G;
  if (x >= U) { cutPos[c] = MONO_NO_CUT; cutSym[c] = 0; flags[c] = 0u; return; }
  const uint64_t lo = blockB ? (x / blockB) * blockB : 0ull;              // split encode: the piece's block is the whole world (see k_mono_cuts8)
  if (blockB) { const uint64_t be = lo + blockB; if (be < U) U = be; }
  const uint64_t hiEnd = (x + G < U) ? x + G : U;
  const uint64_t lastCut = (U > 64u + 4u * SU) ? U - 64u - 4u * SU : 0u;     // no cut near the end of the input (its rules look at the end)
  const uint64_t back = (uint64_t)LONGC + 4u * SU + 16u;
  uint64_t j = (x > lo + back) ? x - back : lo;
  // state of the scan over the match bits: length of the current stretch of set bits and its start; whether that stretch is usable (its
  // start was seen and is clean); position of the zero bit that ended the last stretch of >= S set bits
  uint64_t ones = (j == lo) ? 0 : SU;           // (a stretch that is open where the scan begins counts as long and is not usable)
  uint64_t q = j;
  bool usable = false;
  uint64_t zLong = (j == lo) ? ~0ull : j;       // ~0: none yet.  Scan start: unknown history -> as if a long stretch had just ended here
  uint64_t found = MONO_NO_CUT, fsym = 0;
  const uint64_t symMask = (S >= 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);
  // 16 or 32 positions per trip: their match bits from 16-byte loads (zero_mask16: the run detectors' movemask), then the stretches of
  // set / clear bits with ctz -- a trip per position (two byte loads each) was 117 us of the 88 MB frame's 420 us
  auto stretch_begins = [&](uint64_t at) {
    q = at;
    usable = (zLong == ~0ull) || (zLong + SU <= q);                     // clean start: the last long stretch ended >= S positions in front
    // (16 byte symbols: the first 16 bytes of the stream count as a match of themselves (A.5 q4), a "run" no stretch stands for --
    //  no cut that close to the start)
    if (S == 16 && q < lo + 64u) usable = false;
  };
  bool done = false;
  while (!done && j + SU < U)
  {
    uint64_t m;                                                           // bit k: d[j + k] == d[j + k + S]; `have` of them are positions in front of U - S
    uint32_t have;
    if (j + 32u + SU <= U)
    {
      // 32 positions per trip, all loads in flight together (the scan is one memory latency per trip: 88 MB frame, 16 byte symbols,
      // 16 positions per trip 108 us)
      const u32x4 a0 = ld128(in + j), a1 = ld128(in + j + 16u);
      const u32x4 b0 = (S == 16) ? a1 : ld128(in + j + SU), b1 = ld128(in + j + 16u + SU);
      m = (uint64_t)(zero_mask16(a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w) | (zero_mask16(a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w) << 16));
      have = 32u;
    }
    else if (j + 16u + SU <= U)
    {
      const u32x4 a = ld128(in + j), b = ld128(in + j + SU);
      m = (uint64_t)zero_mask16(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
      have = 16u;
    }
    else
    {
      m = 0u; have = (uint32_t)(U - SU - j);
      for (uint32_t k = 0; k < have; k++) m |= (in[j + k] == in[j + k + SU]) ? (1ull << k) : 0ull;
    }
    // Most windows of noisy data hold no stretch of S set bits and end none: then all that happens is that the stretch open at the window's end
    // (if any) begins -- straight-line code instead of a divergent trip per stretch (88 MB frame, 16 byte symbols: the finder 93 us, half of it
    // in the stretch loop).  Not taken near the piece's end (the loop below knows when to stop there).
    if (have == 32u && (uint32_t)m != 0xFFFFFFFFu && j + 32u <= hiEnd)
    {
      const uint32_t m32 = (uint32_t)m, lead = (uint32_t)__builtin_ctz(~m32);
      if (ones + lead < SU && stretch_of_s<S>(m32) == 0u)
      {
        const uint32_t trail = (m32 >> 31) ? (uint32_t)__builtin_clz(~m32) : 0u;
        ones = trail;
        if (trail != 0u) stretch_begins(j + 32u - trail);
        j += 32u;
        continue;
      }
    }
    uint32_t pos = 0;
    while (pos < have)
    {
      const uint64_t rest = m >> pos;
      if (rest & 1u)
      {
        const uint32_t t = umin((uint32_t)__builtin_ctzll(~rest), have - pos);    // set bits from pos on
        if (ones == 0) stretch_begins(j + pos);
        ones += t; pos += t;
      }
      else
      {
        // the stretch (if any) ended in front of position j + pos
        if (ones >= SU)
        {
          const uint64_t L = ones;
          const uint64_t e = ALIGNED ? q + SU * ((L + SU) / SU) : q + SU + L;
          if (usable && e - q >= LONGC && e > x && e <= hiEnd && e <= lastCut) { found = e; fsym = (S == 16) ? q : (ld64(in + q) & symMask); done = true; break; }   // (16 bytes: WHERE the symbol is)
          zLong = j + pos;
        }
        ones = 0;
        if (j + pos > hiEnd) { done = true; break; }                     // every later stretch ends behind this piece
        pos += umin(rest ? (uint32_t)__builtin_ctzll(rest) : 64u, have - pos);     // clear bits from pos on
      }
    }
    j += have;
  }
  cutPos[c] = found;
  cutSym[c] = fsym;
  flags[c] = (found != MONO_NO_CUT) ? 1u : 0u;
}

// idx = exclusive scan of flags (idx[pieces] = number of cuts).  Chunk 0 starts at 0 with symbol 0; cut j starts chunk j + 1.
__global__ __launch_bounds__(256) void k_mono_scatter(const uint64_t *__restrict__ cutPos, const uint64_t *__restrict__ cutSym, const uint32_t *__restrict__ flags,
                                                      const uint64_t *__restrict__ idx, uint32_t pieces, uint64_t U, uint64_t *__restrict__ starts, uint64_t *__restrict__ syms,
                                                      uint64_t *__restrict__ slotOff, uint32_t *__restrict__ ctrl)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  auto slot_of = [](uint64_t at, uint64_t k) -> uint64_t { return (at + (at >> 7) + 256ull * k + 15ull) & ~15ull; };
  if (c == 0u)
  {
    const uint64_t cuts = idx[pieces];
    starts[0] = 0; syms[0] = 0; slotOff[0] = 0;
    starts[cuts + 1ull] = U;
    ctrl[0] = (uint32_t)(cuts + 1ull);                                  // chunks
  }
  if (c < pieces && flags[c] != 0u)
  {
    const uint64_t k = idx[c] + 1ull;
    starts[k] = cutPos[c]; syms[k] = cutSym[c]; slotOff[k] = slot_of(cutPos[c], k);
  }
}

// longest chunk (for the encode kernel's step bound): ctrl[1] (zeroed by the host) = max over the chunks
__global__ __launch_bounds__(256) void k_mono_longest(const uint64_t *__restrict__ starts, uint32_t *__restrict__ ctrl)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  uint32_t m = 0;
  if (k < ctrl[0])
  {
    const uint64_t n = starts[k + 1u] - starts[k];
    m = (n > 0xFFFFFFFFull) ? 0xFFFFFFFFu : (uint32_t)n;
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    const uint32_t y = (uint32_t)__shfl_xor((int)m, dd, 64);
    m = y > m ? y : m;
  }
  if ((threadIdx.x & 63u) == 0u && m != 0u) atomicMax(ctrl + 1, m);
}

// ---- codecs with a move-to-front list (LUT3 / LUT7, Short1 / 3 / 7) -------------------------------------------------------------
// A run of >= LONG bytes fixes lastRLE, not the list: the list in front of a chunk is whatever the runs stored before it left.  A
// move-to-front list is an LRU stack, so what a chunk does to it is "these d symbols (most recent first) now lead, the older entries
// follow in their order" -- IF the chunk stores the same runs whatever list it is handed, which is nearly always so (the list only
// enters the decision through a penalty of one symbol's worth of bytes for an absent symbol).  So:
//   1. every chunk is walked DRY (no stores) from the default list: its d leading symbols (k_encode*_blocks<.., MONO>, mtfDepth)
//   2. k_mono_list_tiles x 2 + k_mono_list_guess: the lists in front of all chunks by composing those (64 / 4096 chunk roll-ups)
//   3. every chunk is encoded from its guessed list and reports the list it ends with and its d -- now from (nearly) the right list
//   4. 2 again with those: chunks whose list comes out different are encoded again (3, they only), until no list changes.  A change
//      travels any distance in one round as long as the chunks in between keep their decisions.
//   5. k_mono_list_verify, the proof: is every chunk's list the one its predecessor ended with?  Chunk 0's is (the default list), so
//      if all agree the stream is the sequential encoder's by induction.  (A fixed point of 4 passes; chunks that fail would get the
//      right list and go through 3 again.)
// Lists are 8 words per chunk: entry k (the symbol, S bytes, zero-extended) in word k, word 7 = d (results) / "encode me" (guesses).

// (mono_default_entry, MonoListAcc: hsrle_common.hip.h -- the ring encoders compose lists in their split mode too)

// roll-ups: out[t] = the transformers src[64 t .. 64 t + 63] composed (one lane each)
__global__ __launch_bounds__(64) void k_mono_list_tiles(const uint64_t *__restrict__ src, uint32_t count, uint32_t K, uint64_t *__restrict__ out)
{
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  const uint32_t first = t * 64u;
  if (first >= count) return;
  const uint32_t last = (first + 64u < count) ? first + 64u : count;
  MonoListAcc a; a.n = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) a.e[k] = 0;
  for (uint32_t i = last; i > first && a.n < K; i--) a.add(src + 8ull * (i - 1u), K);
#pragma unroll
  for (int k = 0; k < 7; k++) out[8ull * t + k] = a.e[k];
  out[8ull * t + 7] = a.n;
}

// guess[c] = the list behind chunks 0 .. c - 1 applied to the default list; word 7 = 1 (encode) if that is not the list the chunk was
// last encoded from (or `force`), *todo counts those
__global__ __launch_bounds__(64) void k_mono_list_guess(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, const uint64_t *__restrict__ t2, uint32_t chunks, uint32_t K,
                                                        uint32_t S, uint64_t *__restrict__ guess, uint32_t force, uint32_t *__restrict__ todo)
{
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= chunks) return;
  MonoListAcc a; a.n = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) a.e[k] = 0;
  for (uint32_t i = c; i > (c & ~63u) && a.n < K; i--) a.add(t0 + 8ull * (i - 1u), K);
  const uint32_t t = c >> 6;
  for (uint32_t i = t; i > (t & ~63u) && a.n < K; i--) a.add(t1 + 8ull * (i - 1u), K);
  for (uint32_t i = t >> 6; i > 0u && a.n < K; i--) a.add(t2 + 8ull * (i - 1u), K);
  for (uint32_t k = 0; k < K && a.n < K; k++) a.add_one(mono_default_entry(k, S), K);
  bool changed = force != 0u;
#pragma unroll
  for (int k = 0; k < 7; k++)
    if ((uint32_t)k < K && guess[8ull * c + k] != a.e[k]) { changed = true; guess[8ull * c + k] = a.e[k]; }
  guess[8ull * c + 7] = changed ? 1ull : 0ull;
  if (changed) atomicAdd(todo, 1u);
}

// all chunks start from the default list (the dry pass)
__global__ __launch_bounds__(256) void k_mono_list_default(uint32_t chunks, uint32_t K, uint32_t S, uint64_t *__restrict__ guess)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= chunks) return;
#pragma unroll
  for (int k = 0; k < 7; k++) guess[8ull * c + k] = ((uint32_t)k < K) ? mono_default_entry((uint32_t)k, S) : 0ull;
  guess[8ull * c + 7] = 1ull;
}

// chunk c's guess against the list chunk c - 1 ended with; wrong ones are corrected and marked for another pass; *bad counts them
__global__ __launch_bounds__(256) void k_mono_list_verify(uint64_t *__restrict__ guess, const uint64_t *__restrict__ listOut, uint32_t chunks, uint32_t K, uint32_t *__restrict__ bad)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= chunks) return;
  bool wrong = false;
  if (c > 0u)
  {
#pragma unroll
    for (int k = 0; k < 7; k++)
      if ((uint32_t)k < K)
      {
        const uint64_t want = listOut[8ull * (c - 1u) + k];
        if (guess[8ull * c + k] != want) { wrong = true; guess[8ull * c + k] = want; }
      }
  }
  guess[8ull * c + 7] = wrong ? 1ull : 0ull;
  if (wrong) atomicAdd(bad, 1u);
}

// LANES lanes per chunk: staging slot -> dst + offsets[c].  A wave per chunk for the chunks of a monolithic stream (kilobytes each); a quarter wave for
// the chunks of a small container's blocks (a 1 KiB piece of a video-shaped frame is ~170 bytes of stream: 11 of a wave's 64 lanes had work,
// 88 MB frame 29 us)
template <int LANES>
__global__ __launch_bounds__(256) void k_compact_var_t(const uint8_t *__restrict__ slots, const uint64_t *__restrict__ slotOff, const uint64_t *__restrict__ offsets,
                                                       uint8_t *__restrict__ dst, uint32_t chunks)
{
  static_assert(LANES == 16 || LANES == 64, "a quarter wave or a wave per chunk");
  const uint32_t lane = threadIdx.x % (uint32_t)LANES;
  const uint32_t c = blockIdx.x * (256u / (uint32_t)LANES) + threadIdx.x / (uint32_t)LANES;
  if (c >= chunks) return;
  const uint64_t off = offsets[c];
  const uint64_t size = offsets[c + 1u] - off;
  const uint8_t *src = slots + slotOff[c];
  uint8_t *d = dst + off;
  uint64_t head = (16u - ((uintptr_t)d & 15u)) & 15u;
  if (head > size) head = size;
  if (lane < head) d[lane] = src[lane];
  const uint64_t body = (size - head) & ~15ull;
  for (uint64_t k = (uint64_t)lane * 16u; k < body; k += (uint64_t)LANES * 16u)
    st128(d + head + k, ld128(src + head + k));
  const uint64_t tail = size - head - body;
  if (lane < tail) d[head + body + lane] = src[head + body + lane];
}
inline void launch_compact_var(bool smallChunks, const uint8_t *slots, const uint64_t *slotOff, const uint64_t *offsets, uint8_t *dst, uint32_t chunks, hipStream_t st)
{
  if (smallChunks) hipLaunchKernelGGL((k_compact_var_t<16>), dim3((chunks + 15u) / 16u), dim3(256), 0, st, slots, slotOff, offsets, dst, chunks);
  else hipLaunchKernelGGL((k_compact_var_t<64>), dim3((chunks + 3u) / 4u), dim3(256), 0, st, slots, slotOff, offsets, dst, chunks);
}

// the Single chunk encoders return 0 for a chunk that did not end on a stored run at its boundary (a cut that was none): ctrl[5] = 1, the
// host then falls back to one lane instead of handing out a stream with a hole
__global__ __launch_bounds__(256) void k_mono_zero_sizes(const uint32_t *__restrict__ sizes, uint32_t chunks, uint32_t *__restrict__ flag)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c < chunks && sizes[c] == 0u) *flag = 1u;
}

// stream header: {u32 uncompressed, u32 compressed (header included), [u8 mode = 0]}; ctrl[2..3] = the stream's size
// symbolAt8: rle8_single_short -- an 8-byte header, then the stream's symbol
__global__ void k_mono_finish(uint8_t *__restrict__ out, uint32_t U, uint32_t headerSize, const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ ctrlIn, uint32_t *__restrict__ ctrl,
                              uint32_t symbolAt8 = 0)
{
  if (threadIdx.x == 0)
  {
    const uint64_t total = offsets[ctrlIn[0]] + headerSize;
    st32(out, U);
    st32(out + 4, (uint32_t)total);
    if (headerSize == 9u) out[8] = symbolAt8 ? (uint8_t)ctrlIn[8] : (uint8_t)0;
    if (headerSize == 10u) { out[8] = 1; out[9] = (uint8_t)ctrlIn[8]; }  // 8 bit Single: mode 1 and the symbol (k_single_pick_final left it in ctrl[8])
    ctrl[2] = (uint32_t)total; ctrl[3] = (uint32_t)(total >> 32);
  }
}

// ---- split encode: the chunk table of a CONTAINER (blocks of B bytes, pieces of G = B / ppb bytes).  Chunk starts = the block starts and the
//      cuts; idx = exclusive scan of the pieces' cut flags.  Piece p's cut opens chunk idx[p] + p / ppb + 1 (every block start up to and including
//      p's block comes first), block b opens chunk idx[b * ppb] + b.  ctrl[0] = the number of chunks.
constexpr uint32_t kSplitGuessCutSym = 0x100u;   // listWords flag: the first guess of a cut-born chunk's list leads with the boundary run's symbol
__global__ __launch_bounds__(256) void k_split_scatter(const uint64_t *__restrict__ cutPos, const uint64_t *__restrict__ cutSym, const uint32_t *__restrict__ flags,
                                                       const uint64_t *__restrict__ idx, uint32_t pieces, uint32_t ppb, uint32_t nBlocks, uint64_t U, uint32_t B,
                                                       uint64_t *__restrict__ starts, uint64_t *__restrict__ syms, uint64_t *__restrict__ slotOff, uint32_t *__restrict__ firstChunk,
                                                       uint32_t *__restrict__ ctrl, uint32_t listWords, uint32_t blocksPerWave)
{
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  auto slot_of = [](uint64_t at, uint64_t k) -> uint64_t { return (at + (at >> 7) + 256ull * k + 15ull) & ~15ull; };
  if (p == 0u)
  {
    const uint64_t total = idx[pieces] + nBlocks;
    starts[total] = U;
    firstChunk[nBlocks] = (uint32_t)total;
    ctrl[0] = (uint32_t)total;
    ctrl[1] = blocksPerWave;                                             // != 0: the encoder takes whole blocks per wave and settles the lists itself
    ctrl[2] = nBlocks;
  }
  if (p >= pieces) return;
  const uint32_t b = p / ppb;
  if (p % ppb == 0u)
  {
    const uint64_t k = idx[p] + b, at = (uint64_t)b * B;
    starts[k] = at; slotOff[k] = slot_of(at, k); firstChunk[b] = (uint32_t)k;
    if (listWords == 0u) syms[k] = 0ull;                                 // (codecs with a list: syms is the guess array, filled by k_mono_list_default)
  }
  if (flags[p] != 0u)
  {
    const uint64_t k = idx[p] + b + 1ull;
    starts[k] = cutPos[p]; slotOff[k] = slot_of(cutPos[p], k);
    if (listWords == 0u) syms[k] = cutSym[p];
    else if (listWords == (1u | kSplitGuessCutSym)) syms[8ull * k] = cutSym[p];   // a list of ONE symbol behind a stored run is that run's symbol
  }
}

// split encode, list codecs: after the pass from the default lists every chunk knows what it does to a list (its d leading symbols, listOut);
// the list in front of chunk c = those of the chunks in front of it IN ITS BLOCK composed over the default list (at most ppb of them)
__global__ __launch_bounds__(256) void k_split_list_guess(uint64_t *__restrict__ guess, const uint64_t *__restrict__ listOut, const uint64_t *__restrict__ starts, const uint32_t *__restrict__ ctrl,
                                                          uint32_t B, uint32_t K, uint32_t S, uint32_t *__restrict__ todo)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= ctrl[0]) return;
  MonoListAcc a; a.n = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) a.e[k] = 0;
  for (uint32_t i = c; a.n < K && (starts[i] % B) != 0ull; i--) a.add(listOut + 8ull * (i - 1u), K);      // (chunk i does not open its block: chunk i - 1 is in it)
  for (uint32_t k = 0; k < K && a.n < K; k++) a.add_one(mono_default_entry(k, S), K);
  bool changed = false;
#pragma unroll
  for (int k = 0; k < 7; k++)
    if ((uint32_t)k < K && guess[8ull * c + k] != a.e[k]) { changed = true; guess[8ull * c + k] = a.e[k]; }
  guess[8ull * c + 7] = changed ? 1ull : 0ull;
  if (changed) atomicAdd(todo, 1u);
}

// the proof for the list codecs, block aware: chunk c's list against the list chunk c - 1 ended with -- or the default list if c opens a block
__global__ __launch_bounds__(256) void k_split_list_verify(uint64_t *__restrict__ guess, const uint64_t *__restrict__ listOut, const uint64_t *__restrict__ starts, const uint32_t *__restrict__ ctrl,
                                                           uint32_t B, uint32_t K, uint32_t S, uint32_t *__restrict__ bad)
{
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= ctrl[0]) return;
  const bool opens = (starts[c] % B) == 0ull;
  bool wrong = false;
#pragma unroll
  for (int k = 0; k < 7; k++)
    if ((uint32_t)k < K)
    {
      const uint64_t want = opens ? mono_default_entry((uint32_t)k, S) : listOut[8ull * (c - 1u) + k];
      if (guess[8ull * c + k] != want) { wrong = true; guess[8ull * c + k] = want; }
    }
  guess[8ull * c + 7] = wrong ? 1ull : 0ull;
  if (wrong) atomicAdd(bad, 1u);
}

// The per-lane chunk encoders of the split encode (8 bit Single, rle8_single_short, 128 bit, Greedy) report a chunk that did not end on its boundary
// run with size 0 (the cut heuristics make that impossible as far as anyone has seen; the monolithic path checks the same thing with
// k_mono_zero_sizes and falls back).  A hole in a block's stream must not leave the library as a valid container: a non-final, non-empty chunk of
// size 0 raises the flag, and k_split_verdict then strikes the container's magic -- every decoder refuses it.
__global__ __launch_bounds__(256) void k_split_check(const uint32_t *__restrict__ firstChunk, const uint64_t *__restrict__ starts, const uint32_t *__restrict__ sizes, uint32_t nBlocks,
                                                     uint32_t *__restrict__ flag)
{
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= nBlocks) return;
  const uint32_t c0 = firstChunk[b], c1 = firstChunk[b + 1u];
  for (uint32_t c = c0; c + 1u < c1; c++)
    if (sizes[c] == 0u && starts[c + 1u] > starts[c]) *flag = 1u;
}
__global__ void k_split_verdict(const uint32_t *__restrict__ flag, uint8_t *__restrict__ container)
{
  if (threadIdx.x < 8u && *flag != 0u) container[threadIdx.x] = 0;
}

// after the placement of the chunks: the container's offset table, every block stream's compressedLength field, and (the workgroup of the last block)
// the container's header and tail pad (FINISH: finish_container of hsrle_capi.hip -- one launch less)
template <typename FINISH>
__global__ __launch_bounds__(256) void k_split_finish(const uint32_t *__restrict__ firstChunk, const uint64_t *__restrict__ chunkOff, uint32_t nBlocks, uint64_t *__restrict__ offsets,
                                                      uint8_t *__restrict__ payload, FINISH finish)
{
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x == (nBlocks - 1u) / 256u) finish(chunkOff[firstChunk[nBlocks]]);     // (every thread of this workgroup: finish_container hands out roles by threadIdx)
  if (b >= nBlocks) return;
  const uint64_t off = chunkOff[firstChunk[b]], next = chunkOff[firstChunk[b + 1u]];
  offsets[b] = off;
  st32(payload + off + 4, (uint32_t)(next - off));
  if (b + 1u == nBlocks) offsets[nBlocks] = next;
}

// ---- the symbol pick for ONE monolithic stream (hsrle_mono_encode.hip.h: the drop-in rle8_single_compress / rle8_packed_single_compress by many
//      lanes).  The same closed form as k_single_pick -- a maximal run of L >= 2 equal bytes that ends in front of U - 16 adds L - (L - 1) / 16
//      to prob[s] and 1 to pcount[s] -- over pieces of P bytes, one wave each; a run belongs to the piece it STARTS in, and the one run that
//      is still open at the piece's end is followed through the next pieces by the whole wave (1 KiB per trip; a run of more than 16 MiB
//      makes the call give up: the drop-in function then walks with one lane, as it did for every input before).  All sums are modulo 2^32,
//      like the reference's counters.  table: u32 prob[256], pcount[256], [512] end of the last safe run, [513] gave up, [514..515] the first
//      late run as (position << 32 | L - 1) (zeroed / set to ~0 by the host).  k_single_pick_final does what lane 0 of k_single_pick does.
constexpr uint32_t kPickPiece = 4096u;
__global__ __launch_bounds__(64) void k_single_pick_mono(const uint8_t *__restrict__ in, uint32_t U, uint32_t pieces, uint32_t *__restrict__ table)
{
  __shared__ __attribute__((aligned(16))) uint8_t bytes[kPickPiece + 80];
  __shared__ uint64_t eqw[kPickPiece / 64u + 1u];
  __shared__ uint32_t ltab[512];                                         // this piece's sums: one device atomic per symbol and wave, not per run
  const uint32_t lane = threadIdx.x, P = kPickPiece;
#pragma unroll
  for (int k = 0; k < 8; k++) ltab[lane * 8u + (uint32_t)k] = 0u;
  uint32_t lastSafeEnd = 0u;
  uint64_t firstLate = ~0ull;
  // a wave takes every gridDim.x-th piece and keeps its sums in LDS: one device atomic per symbol and WAVE at the end (one per symbol and piece
  // was 31 M device-scope atomics for 1 GiB of run-distributed bytes: 12 ms)
  for (uint32_t piece = blockIdx.x; piece < pieces; piece += gridDim.x)
  {
  __syncthreads();
  const uint32_t at = piece * P;
  const uint32_t n = (U - at) < P ? (U - at) : P;
  const uint32_t words = (n + 63u) / 64u;

  // the piece + the byte behind it
  for (uint32_t p = lane * 16u; p < P + 16u; p += 1024u)
  {
    u32x4 v = u32x4{ 0, 0, 0, 0 };
    if ((uint64_t)at + p + 16u <= U) v = ld128(in + at + p);
    else if ((uint64_t)at + p < U)
    {
      uint32_t t[4] = { 0, 0, 0, 0 };
      for (uint32_t k = 0; (uint64_t)at + p + k < U && k < 16u; k++) t[k >> 2] |= (uint32_t)in[at + p + k] << (8u * (k & 3u));
      v = u32x4{ t[0], t[1], t[2], t[3] };
    }
    lds_st128(bytes + p, v);
  }
  __syncthreads();
  // bit i = (d[i] == d[i + 1]) and at + i + 1 < U, for the positions of the piece
  for (uint32_t w = lane; w < words; w += 64u)
  {
    uint64_t e64 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const u32x4 x = lds_ld128(bytes + w * 64u + j * 16u);
      const uint32_t x4 = lds_ld32(bytes + w * 64u + j * 16u + 16u);
      e64 |= (uint64_t)zero_mask16(x.x ^ alignbyte(x.y, x.x, 1), x.y ^ alignbyte(x.z, x.y, 1), x.z ^ alignbyte(x.w, x.z, 1), x.w ^ alignbyte(x4, x.w, 1)) << (16u * j);
    }
    const uint32_t base = w * 64u;
    uint32_t valid = (n > base) ? n - base : 0u;                                       // positions of the piece
    if ((uint64_t)at + n >= U && valid != 0u) valid = (n - 1u > base) ? n - 1u - base : 0u;   // the input's last byte has no successor
    if (valid < 64u) e64 &= (1ull << valid) - 1ull;
    eqw[w] = e64;
  }
  if (lane == 0u) eqw[words] = 0ull;
  __syncthreads();

  const int64_t end = (int64_t)U - 16;
  const uint64_t prevRun = (at > 0u && in[at - 1u] == in[at]) ? 1ull : 0ull;
  uint32_t openJ = 0xFFFFFFFFu;                                                        // the run that is open at the piece's end starts here
  auto account = [&](uint32_t gj, uint32_t L, uint32_t sy) {
    if ((int64_t)gj + (int64_t)L < end)
    {
      atomicAdd(ltab + sy, L - (L - 1u) / 16u);
      atomicAdd(ltab + 256u + sy, 1u);
      lastSafeEnd = (gj + L > lastSafeEnd) ? gj + L : lastSafeEnd;
    }
    else
    {
      const uint64_t key = ((uint64_t)gj << 32) | (uint64_t)(L - 1u);
      firstLate = key < firstLate ? key : firstLate;
    }
  };
  for (uint32_t w = lane; w < words; w += 64u)
  {
    const uint64_t m = eqw[w];
    const uint64_t prevBit = (w > 0u) ? eqw[w - 1u] >> 63 : prevRun;
    uint64_t starts = m & ~((m << 1) | prevBit);
    while (starts != 0ull)
    {
      const uint32_t p = (uint32_t)__builtin_ctzll(starts);
      starts &= starts - 1ull;
      const uint32_t j = w * 64u + p;
      uint32_t ones;
      const uint64_t t = ~(m >> p);
      if (p == 0u) ones = (m == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~m);
      else ones = (uint32_t)__builtin_ctzll(t | (1ull << 63));
      if (ones == 64u - p)
      {
        uint32_t w2 = w + 1u;
        for (;;)
        {
          const uint64_t mm = eqw[w2];                                   // eqw[words] = 0 ends every run
          if (mm == ~0ull) { ones += 64u; w2++; continue; }
          ones += (uint32_t)__builtin_ctzll(~mm);
          break;
        }
      }
      if (j + ones >= n && (uint64_t)at + n < U) openJ = j;              // its last byte is the first of the next piece: followed below
      else account(at + j, ones + 1u, bytes[j]);
    }
  }
  // the open run (at most one per piece), by the whole wave
  const uint64_t who = __ballot(openJ != 0xFFFFFFFFu);
  if (who != 0ull)
  {
    const uint32_t j = (uint32_t)__shfl((int)openJ, (int)__builtin_ctzll(who), 64);
    const uint32_t sy = bytes[j];
    const uint32_t sym4 = sy * 0x01010101u;
    uint64_t pos = (uint64_t)at + n + 1u;                                // d[at + n] belongs to the run
    uint32_t ext = 0;
    bool gaveUp = false;
    for (uint32_t trip = 0;; trip++)
    {
      if (trip >= 16384u) { gaveUp = true; break; }
      const uint64_t q = pos + (uint64_t)lane * 16u;
      uint32_t good = 0;                                                 // leading bytes of my 16 that continue the run
      if (q + 16u <= U)
      {
        const u32x4 x = ld128(in + q);
        const uint32_t m16 = zero_mask16(x.x ^ sym4, x.y ^ sym4, x.z ^ sym4, x.w ^ sym4);
        good = (m16 == 0xFFFFu) ? 16u : (uint32_t)__builtin_ctz(~m16);
      }
      else
        for (uint32_t k = 0; q + k < U && in[q + k] == sy; k++) good++;
      const uint64_t broke = __ballot(good < 16u);
      if (broke != 0ull)
      {
        const uint32_t f = (uint32_t)__builtin_ctzll(broke);
        ext += f * 16u + (uint32_t)__shfl((int)good, (int)f, 64);
        break;
      }
      ext += 1024u;
      pos += 1024u;
    }
    if (gaveUp) { if (lane == 0u) table[513] = 1u; }
    else if (lane == 0u) account(at + j, (n - j) + 1u + ext, sy);
  }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    const uint32_t a = (uint32_t)__shfl_xor((int)lastSafeEnd, dd, 64);
    lastSafeEnd = a > lastSafeEnd ? a : lastSafeEnd;
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)firstLate, dd, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(firstLate >> 32), dd, 64);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    firstLate = o < firstLate ? o : firstLate;
  }
  if (lane == 0u)
  {
    if (lastSafeEnd != 0u) atomicMax(table + 512u, lastSafeEnd);
    if (firstLate != ~0ull) atomicMin(reinterpret_cast<unsigned long long *>(table + 514u), (unsigned long long)firstLate);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    const uint32_t v = ltab[lane * 8u + (uint32_t)k];
    if (v != 0u) atomicAdd(table + lane * 8u + (uint32_t)k, v);
  }
}

// one wave: the scanner's first window, the run that reaches U - 16, the final registration (rle8_extreme_cpu.c:66-139, as k_single_pick's lane 0),
// then the argmax.  out[0] = the symbol, out[1] = 1 if the pick gave up
__global__ __launch_bounds__(64) void k_single_pick_final(const uint8_t *__restrict__ in, uint32_t U, uint32_t *__restrict__ table, uint32_t *__restrict__ out)
{
  const uint32_t lane = threadIdx.x;
  const int64_t end = (int64_t)U - 16;
  if (lane == 0u)
  {
    const uint32_t d0 = in[0];
    const uint32_t inv = (~d0) & 0xFFu;
    uint32_t finSym, finCount;
    if (end <= 0) { finSym = inv; finCount = 0u; }
    else
    {
      bool any = false;
      for (uint32_t k = 0; k < 16u; k++) any = any || in[k] == inv;
      if (any) table[256u + inv] += 1u;                                  // registered with count 0
      const uint32_t i0 = table[512];                                    // the search behind the last safe run starts here (< end)
      const uint64_t late = *reinterpret_cast<const uint64_t *>(table + 514u);
      bool have = false;
      finSym = 0; finCount = 1u;
      if (late != ~0ull)
      {
        const uint32_t j = (uint32_t)(late >> 32), L = (uint32_t)late + 1u;
        const uint32_t q = i0 + 15u * ((j - i0) / 15u);                 // the search trip that would find it
        if ((int64_t)q < end)
        {
          have = true;
          uint32_t i = j + 1u, count = 1u;
          bool registered = false;
          while ((int64_t)i < end)
          {
            const uint32_t rem = L - (i - j);
            if (rem >= 16u) { count += 15u; i += 16u; }
            else
            {
              count += rem; i += rem;
              table[in[j]] += count; table[256u + in[j]] += 1u;
              registered = true;
              break;
            }
          }
          if (registered) { finSym = in[i]; finCount = 1u; }
          else { finSym = in[j]; finCount = count; }
        }
      }
      if (!have)
      {
        const uint32_t i = ((int64_t)i0 < end) ? i0 + 15u * (((uint32_t)end - i0 + 14u) / 15u) : i0;   // the search runs off the end
        finSym = in[i]; finCount = 1u;
      }
    }
    table[finSym] += finCount; table[256u + finSym] += 1u;
  }
  __syncthreads();
  const bool zeroStartsFull = in[0] != 0u;                              // pcount[0] starts as 0xFFFFFFFF unless d[0] == 0 (:61-62)
  uint64_t bestKey = 0ull;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t s = lane * 4u + (uint32_t)k;
    const uint32_t prob = table[s];
    uint32_t pc = table[256u + s];
    if (s == 0u && zeroStartsFull) pc -= 1u;                            // modulo 2^32, as the reference's counter
    if (pc > 0u && prob / pc > 2u)
    {
      const uint32_t saved = prob - pc * 2u;
      const uint64_t key = ((uint64_t)saved << 8) | (uint64_t)(255u - s);
      bestKey = key > bestKey ? key : bestKey;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bestKey, dd, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(bestKey >> 32), dd, 64);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    bestKey = o > bestKey ? o : bestKey;
  }
  if (lane == 0u)
  {
    out[0] = (bestKey >> 8) != 0ull ? (255u - (uint32_t)(bestKey & 0xFFull)) : 0u;
    out[1] = table[513];
  }
}


} // namespace hsrle
