// hsrle_encode8s.hip.h -- the 8 bit Single encoders (rle8_single, rle8_packed_single) on the ring data path.
//
// Replaces: src/rle8_extreme_cpu.c:53-153 (symbol pick), src/rle8_extreme_cpu.h:346-700 (wrapper, scalar tail, final block),
//           :1103-1321 (the SSE2 body with its `wastedChances` back-tracking).  SURVEY.md A.7; restated in oracle/hsrle_oracle.c.
//
// Two kernels per container:
//   k_single_pick     one WAVE per block.  The reference's estimator walks the block with a scanner whose bookkeeping is quirky (a full
//                     16-byte window of the run counts 15, the first window is compared with ~d[0], what the last registered run is
//                     depends on the scanner's phase at n - 16) but which, away from the end, registers exactly the maximal runs of
//                     >= 2 equal bytes: a run of L bytes adds L - (L - 1) / 16 to prob[s] and 1 to pcount[s].  (Closed form checked
//                     against the oracle on 90 000 inputs: tools/pick_model.py.)  So: equality bits of the whole block in
//                     LDS, every lane takes the run starts of 64 positions, LDS atomics into a 256-entry table, the one run that
//                     reaches n - 16 and the final registration by lane 0, argmax by shuffles.  The symbol goes to byte 9 (rle8_single_short:
//                     byte 8) of the block's staging slot -- where the stream keeps it -- and the encoder picks it up there.
//   k_encode8_single_blocks<MODE>     (rle8_single, rle8_packed_single and the Short family's rle8_single_short) one lane per block like the other ring encoders (hsrle_encode8.hip.h: same top-up, same
//                     output accumulator).  The scanner of the reference is run as it is -- 16-byte windows at its own, data
//                     dependent phase, the skip rule of the search, the back-track to the first wasted run -- on match bits taken
//                     from the ring (from global memory for positions that have left it).
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low_m, wave_sync

namespace hsrle {

#ifndef HSRLE_SINGLE_RING
#define HSRLE_SINGLE_RING 256   // (128: 12 instead of 9 waves per CU, run data 14.2 -> 13.1 ms, video-shaped 13.8 -> 18.9: lanes that lag get fewer chunks per step)
#endif
constexpr uint32_t kSinglePickMaxBlock = 32768u;   // larger blocks use the first-generation kernel (hsrle_encode.hip.h)

// dynamic LDS: [0, 1024) table (prob | pcount << 16), then n bytes of the block (padded to 64), then the equality bits
// cutPos != nullptr (round 4, split encode of a small container): the wave also finds its block's CUTS -- for each of the block's pieces of
// cutG bytes the first run of the picked symbol of >= cutLong bytes that ends inside the piece (and 64 bytes in front of the block's end), as
// k_mono_cuts8 does with one lane per piece; here the block is in LDS already and the 64 lanes look at 64 positions each.
__global__ __launch_bounds__(64) void k_single_pick(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *__restrict__ slots, uint32_t slotStride,
                                                    uint32_t symAt, uint64_t *__restrict__ cutPos = nullptr, uint64_t *__restrict__ cutSym = nullptr, uint32_t *__restrict__ cutFlags = nullptr,
                                                    uint32_t cutG = 0, uint32_t cutLong = 0)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t pickLds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
  if (b >= nBlocks) return;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint32_t padded = (B + 63u) & ~63u;
  uint32_t *const table = (uint32_t *)pickLds;
  uint8_t *const bytes = pickLds + 1024;
  uint64_t *const eqw = (uint64_t *)(pickLds + 1024 + padded + 64u);    // (+64: the byte reads of the last window stay inside)
  const uint32_t words = (n + 63u) / 64u;

#pragma unroll
  for (int k = 0; k < 4; k++) table[lane * 4u + k] = 0u;

  // ---- phase 1: the block to LDS, 16 bytes per lane and load ----
  for (uint32_t p = lane * 16u; p < padded; p += 1024u)
  {
    u32x4 v = u32x4{ 0, 0, 0, 0 };
    if (p + 16u <= n) v = ld128(in + at + p);
    else if (p < n)
    {
      uint32_t t[4] = { 0, 0, 0, 0 };
      for (uint32_t k = 0; p + k < n; k++) t[k >> 2] |= (uint32_t)in[at + p + k] << (8u * (k & 3u));
      v = u32x4{ t[0], t[1], t[2], t[3] };
    }
    lds_st128(bytes + p, v);
  }
  __syncthreads();

  // ---- phase 2: equality bits, 64 positions per lane and trip: bit i = (d[i] == d[i + 1]) and i + 1 < n ----
  for (uint32_t w = lane; w < words; w += 64u)
  {
    uint64_t e64 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const u32x4 x = lds_ld128(bytes + w * 64u + j * 16u);
      const uint32_t x4 = lds_ld32(bytes + w * 64u + j * 16u + 16u);
      const uint32_t zm16 = zero_mask16(x.x ^ alignbyte(x.y, x.x, 1), x.y ^ alignbyte(x.z, x.y, 1), x.z ^ alignbyte(x.w, x.z, 1), x.w ^ alignbyte(x4, x.w, 1));
      e64 |= (uint64_t)zm16 << (16u * j);
    }
    const uint32_t base = w * 64u;
    const uint32_t valid = (n - 1u > base) ? n - 1u - base : 0u;        // positions base + i with base + i + 1 < n
    if (valid < 64u) e64 &= (1ull << valid) - 1ull;
    eqw[w] = e64;
  }
  if (lane == 0u) eqw[words] = 0ull;
  __syncthreads();

  // ---- phase 3: the runs that start in my 64 positions.  Safe runs (they end in front of n - 16) go to the table; of the others the
  //      first one is kept for lane 0, and the end of the last safe run ----
  const int32_t end = (int32_t)n - 16;
  uint32_t lastSafeEnd = 0u;                 // max over the safe runs of j + L
  uint32_t firstLate = 0xFFFFFFFFu;          // min over the late runs of (j << 16) | (L - 1)   (j < 32768, L <= 32768)
  for (uint32_t w = lane; w < words; w += 64u)
  {
    const uint64_t m = eqw[w];
    const uint64_t prevBit = (w > 0u) ? eqw[w - 1u] >> 63 : 0ull;
    uint64_t starts = m & ~((m << 1) | prevBit);
    while (starts != 0ull)
    {
      const uint32_t p = (uint32_t)__builtin_ctzll(starts);
      starts &= starts - 1ull;
      const uint32_t j = w * 64u + p;
      // ones from position j on
      uint32_t ones;
      const uint64_t t = ~(m >> p);
      const uint32_t inWord = (uint32_t)__builtin_ctzll(t | (1ull << 63));      // (bit 63 of m >> p is 0 for p > 0; for p = 0 see below)
      if (p == 0u) ones = (m == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~m);
      else ones = inWord;
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
      const uint32_t L = ones + 1u;
      if ((int32_t)(j + L) < end)
      {
        const uint32_t sy = bytes[j];
        atomicAdd(table + sy, (1u << 16) | (L - (L - 1u) / 16u));
        lastSafeEnd = (j + L > lastSafeEnd) ? j + L : lastSafeEnd;
      }
      else
      {
        const uint32_t key = (j << 16) | (L - 1u);
        firstLate = key < firstLate ? key : firstLate;
      }
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    const uint32_t a = (uint32_t)__shfl_xor((int)lastSafeEnd, dd, 64), c = (uint32_t)__shfl_xor((int)firstLate, dd, 64);
    lastSafeEnd = a > lastSafeEnd ? a : lastSafeEnd;
    firstLate = c < firstLate ? c : firstLate;
  }
  __syncthreads();

  // ---- the scanner's first window, the run that reaches n - 16 and the final registration (rle8_extreme_cpu.c:66-139) ----
  if (lane == 0u)
  {
    const uint32_t d0 = bytes[0];
    const uint32_t inv = (~d0) & 0xFFu;
    uint32_t finSym, finCount;
    if (end <= 0) { finSym = inv; finCount = 0u; }
    else
    {
      bool any = false;
      for (uint32_t k = 0; k < 16u; k++) any = any || bytes[k] == inv;
      if (any) table[inv] += 1u << 16;                                  // registered with count 0
      const uint32_t i0 = lastSafeEnd;                                  // the search behind the last safe run starts here (< end)
      bool have = false;
      finSym = 0; finCount = 1u;
      if (firstLate != 0xFFFFFFFFu)
      {
        const uint32_t j = firstLate >> 16, L = (firstLate & 0xFFFFu) + 1u;
        const uint32_t q = i0 + 15u * ((j - i0) / 15u);                 // the search trip that would find it
        if ((int32_t)q < end)
        {
          have = true;
          uint32_t i = j + 1u, count = 1u;
          bool registered = false;
          while ((int32_t)i < end)
          {
            const uint32_t rem = L - (i - j);
            if (rem >= 16u) { count += 15u; i += 16u; }
            else
            {
              count += rem; i += rem;
              table[bytes[j]] += (1u << 16) | count;
              registered = true;
              break;
            }
          }
          if (registered) { finSym = bytes[i]; finCount = 1u; }
          else { finSym = bytes[j]; finCount = count; }
        }
      }
      if (!have)
      {
        const uint32_t i = ((int32_t)i0 < end) ? i0 + 15u * (((uint32_t)end - i0 + 14u) / 15u) : i0;   // the search runs off the end
        finSym = bytes[i]; finCount = 1u;
      }
    }
    table[finSym] += (1u << 16) | finCount;
  }
  __syncthreads();

  // ---- argmax of prob - 2 pcount over the symbols with pcount > 0 and prob / pcount > 2; the first maximum wins ----
  const bool zeroStartsFull = bytes[0] != 0u;                           // pcount[0] starts as 0xFFFFFFFF unless d[0] == 0 (:61-62)
  uint32_t bestKey = 0u;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t s = lane * 4u + (uint32_t)k;
    const uint32_t v = table[s];
    const uint32_t prob = v & 0xFFFFu;
    uint32_t pc = v >> 16;
    if (s == 0u && zeroStartsFull) pc -= 1u;                            // modulo 2^32, as the reference's counter
    if (pc > 0u && prob / pc > 2u)
    {
      const uint32_t saved = prob - pc * 2u;
      const uint32_t key = (saved << 8) | (255u - s);
      bestKey = key > bestKey ? key : bestKey;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    const uint32_t a = (uint32_t)__shfl_xor((int)bestKey, dd, 64);
    bestKey = a > bestKey ? a : bestKey;
  }
  const uint32_t picked = (bestKey >> 8) != 0u ? (255u - (bestKey & 0xFFu)) : 0u;
  if (lane == 0u)
    slots[(uint64_t)b * slotStride + symAt] = (uint8_t)picked;

  if (cutPos != nullptr)
  {
    // ---- the block's cuts: runs of `picked` of >= cutLong bytes, the first that ends in each piece ----
    __syncthreads();
    uint32_t *const firstCut = table;                                   // (the estimator's table is done with) one word per piece
    const uint32_t ppb = B / cutG;
    if (lane < 64u) for (uint32_t k = lane; k < ppb; k += 64u) firstCut[k] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lastCut = n > 64u ? n - 64u : 0u;
    const uint32_t pv = picked * 0x01010101u;
    uint32_t carry = 0;                                                 // bytes of the symbol that reach the start of this row of 64 words
    for (uint32_t w0 = 0; w0 < words; w0 += 64u)
    {
      const uint32_t w = w0 + lane;
      uint64_t m = 0;
      if (w < words)
      {
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++)
        {
          const u32x4 x = lds_ld128(bytes + w * 64u + j * 16u);
          m |= (uint64_t)zero_mask16(x.x ^ pv, x.y ^ pv, x.z ^ pv, x.w ^ pv) << (16u * j);
        }
        const uint32_t left = n - w * 64u;                              // positions of the word inside the block
        if (left < 64u) m &= (1ull << left) - 1ull;
      }
      // symbol bytes at the END of each word, and through whole words: inclusive scan of (length, "the word is all symbol")
      uint32_t tl = (m == ~0ull) ? 64u : (uint32_t)__builtin_clzll(~m);
      uint32_t full = (m == ~0ull) ? 1u : 0u;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1)
      {
        const uint32_t pl = (uint32_t)__shfl_up((int)tl, d, 64), pf = (uint32_t)__shfl_up((int)full, d, 64);
        if (lane >= d) { tl += full ? pl : 0u; full &= pf; }
      }
      uint32_t cin = (uint32_t)__shfl_up((int)tl, 1, 64);
      const uint32_t cfull = (uint32_t)__shfl_up((int)full, 1, 64);
      if (lane == 0u) cin = carry; else if (cfull) cin += carry;       // (all words in front of this one in the row are symbol: the row's carry comes on top)
      const uint32_t rowTail = (uint32_t)__builtin_amdgcn_readlane((int)tl, 63), rowFull = (uint32_t)__builtin_amdgcn_readlane((int)full, 63);
      // run ends in this word: a clear bit whose left neighbour is set
      uint64_t ends = ~m & ((m << 1) | (cin != 0u ? 1ull : 0ull));
      if (w >= words) ends = 0ull;
      while (ends != 0ull)
      {
        const uint32_t k = (uint32_t)__builtin_ctzll(ends);
        ends &= ends - 1ull;
        const uint64_t zerosBelow = ~m & ((1ull << k) - 1ull);
        const uint32_t len = zerosBelow == 0ull ? k + cin : k - 64u + (uint32_t)__builtin_clzll(zerosBelow);
        const uint32_t e = w * 64u + k;
        if (len >= cutLong && e <= lastCut && e != 0u)
          atomicMin(firstCut + (e - 1u) / cutG, e);
      }
      carry = rowFull ? carry + rowTail : rowTail;
    }
    __syncthreads();
    for (uint32_t k = lane; k < ppb; k += 64u)
    {
      const uint32_t e = firstCut[k];
      const uint64_t pieceAt = at + (uint64_t)k * cutG;
      const bool have = e != 0xFFFFFFFFu && pieceAt < U;
      cutPos[(uint64_t)b * ppb + k] = have ? at + e : ~0ull;             // (MONO_NO_CUT)
      cutSym[(uint64_t)b * ppb + k] = have ? (uint64_t)picked : 0ull;
      cutFlags[(uint64_t)b * ppb + k] = have ? 1u : 0u;
    }
  }
}

// MODE 0: rle8_single, 1: rle8_packed_single, 2: rle8_single_short (rleX_Xsl_short.h with SINGLE: wrapper :380-523, body :1058-1120 -- the
// same estimator and the same kind of scanner: windows with fewer than two occurrences are skipped unless their last byte is one, every
// run end goes through process_symbol (:152-372, one-byte or three-byte header, no symbol in the packet), no wasted-chances logic)
template <int MODE>
__global__ __launch_bounds__(64) void k_encode8_single_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *__restrict__ slots, uint32_t slotStride,
                                                              uint32_t *__restrict__ sizes)
{
  // (every lambda below is always_inline: one that stays a function -- match16 did, with its five call sites -- keeps all it captures by
  //  reference in scratch memory: 2x the kernel time)
  constexpr int Q = 64;                      // input bytes per lane and step
  constexpr int H = HSRLE_SINGLE_RING;       // history ring per lane
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr bool PACKED = MODE == 1, SSHORT = MODE == 2;
  using TRS = Traits<SHORT_SINGLE, 1, 0>;                                // header parameters of the Short family's Single codec
  constexpr int32_t SHORT = SSHORT ? 2 : (PACKED ? 2 : 4), MEDIUM = 6, LONG = PACKED ? 10 : 8;
  constexpr uint32_t SYM_AT = SSHORT ? 8u : 9u;                           // where the stream keeps its symbol (k_single_pick leaves it there)

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
#define HS_SMERGE1(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))

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
  const uint32_t sym = active ? (uint32_t)slot[SYM_AT] : 0u;            // k_single_pick's choice
  const uint32_t sym4 = sym * 0x01010101u;

  uint32_t avail = 0;
  bool finished = !active;

  // ---- output: 16-byte accumulator + stream position (as k_encode8_blocks) ----
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  u32x4 oacc = zero4;
  uint32_t opos = 0;
  // The bytes below accLow of the accumulator's chunk are not the accumulator's: they are the end of a literal stretch that the wave copies
  // later (emit_literals).  Such a chunk is stored byte by byte from accLow on; in all other chunks accLow is 0.
  uint32_t accLow = 0;
  auto store_bytes = [&](uint8_t *p, u32x4 w, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
    const uint64_t w0 = (uint64_t)w.x | ((uint64_t)w.y << 32), w1 = (uint64_t)w.z | ((uint64_t)w.w << 32);
    for (uint32_t k = lo; k < hi; k++)
      p[k] = (uint8_t)((k < 8u ? w0 >> (8u * k) : w1 >> (8u * (k - 8u))) & 0xFFull);
  };
  auto store_chunk = [&](uint8_t *p, u32x4 w) __attribute__((always_inline)) {
    if (__builtin_expect(accLow == 0u, 1)) st128(p, w);
    else { store_bytes(p, w, accLow, 16u); accLow = 0u; }
  };
  auto append = [&](u32x4 hv, uint32_t nb) __attribute__((always_inline)) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);
    const u32x4 w = HS_SMERGE1(oacc, lowp, c);
    if (c + nb >= 16u)
    {
      store_chunk(slot + (opos & ~15u), w);
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
    append(u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), hhi, 0u }, hn);
    hlo = 0; hhi = 0; hn = 0;
  };
  auto ring_win = [&](uint32_t p) __attribute__((always_inline)) -> u32x4 {
    const uint32_t a0 = p & ~15u;
    return funnel16(lds_ld128(hist + (hbase ^ (a0 & HM))), lds_ld128(hist + (hbase ^ ((a0 + 16u) & HM))), p & 15u);
  };
  // is the 16-byte window at block position p still in the ring?
  auto in_ring = [&](uint32_t p) __attribute__((always_inline)) -> bool { return p + (uint32_t)H >= avail + 16u; };
  // Literal bytes [from, from + len) of the block.  The Single codecs store few runs, so literal stretches are long and have mostly left the
  // ring when their packet is written.  Fetching them from global memory inside the lane's own code pays a memory latency per 16 bytes with
  // ONE lane of the wave active (a 4 KiB block without a stored run: 256 of them -- 4x the time of everything else; and even a single such
  // load per packet was 2/3 of the kernel), so such a stretch is only NOTED here (two note slots per lane) and copied by the whole wave at
  // a point where the wave is converged (coop_flush): 1 KiB per load / store pair, any alignment.  Every trip of the scanner and every
  // round of the tail stores at most one run, and a flush follows as soon as a lane has both slots in use.
  uint32_t pendSrc = 0, pendDst = 0, pendBytes = 0, pend2Src = 0, pend2Dst = 0, pend2Bytes = 0;
  auto emit_literals = [&](uint32_t from, uint32_t len) __attribute__((always_inline)) {
    if (len == 0u) return;
    if (__builtin_expect(in_ring(from), 1))
    {
      const uint32_t c = opos & 15u, total = c + len;
      const uint32_t srcp = from - c;
      uint8_t *const dst = slot + (opos & ~15u);
      u32x4 w = HS_SMERGE1(oacc, ring_win(srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        store_chunk(dst + k, w);
        k += 16u;
        if (k < total) w = ring_win(srcp + k);
      }
      oacc = w;
      opos += len;
    }
    else
    {
      // what the accumulator holds goes out now; the stretch is noted; the accumulator starts again behind it
      const uint32_t c = opos & 15u;
      if (c > accLow) store_bytes(slot + (opos & ~15u), oacc, accLow, c);
      if (pendBytes == 0u) { pendSrc = from; pendDst = opos; pendBytes = len; }
      else { pend2Src = from; pend2Dst = opos; pend2Bytes = len; }
      opos += len;
      accLow = opos & 15u;
      oacc = zero4;
    }
  };
  // wave-converged: copy the noted stretches, 16 bytes per lane and load (source at any alignment), two stretches in flight.  Source and
  // destination are private to the lane's block and stay as they are, so the copies can wait: coop_flush(false) only acts when some lane
  // has both of its note slots in use -- by then most lanes have a note or two, and a flush pays one memory latency for all of them.
  auto coop_copy2 = [&](uint32_t l0, uint32_t s0, uint32_t d0, uint32_t n0, uint32_t l1, uint32_t s1, uint32_t d1, uint32_t n1) __attribute__((always_inline)) {
    const uint8_t *const sp0 = in + (uint64_t)(wgFirst + l0) * B + s0, *const sp1 = in + (uint64_t)(wgFirst + l1) * B + s1;
    uint8_t *const dp0 = slots + (uint64_t)(wgFirst + l0) * slotStride + d0, *const dp1 = slots + (uint64_t)(wgFirst + l1) * slotStride + d1;
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
    // the last n & 15 bytes: lanes 0..15 for the first stretch, 16..31 for the second
    const uint32_t r = lane & 15u;
    if (lane < 16u) { if (r < (n0 & 15u)) dp0[f0 + r] = sp0[f0 + r]; }
    else if (lane < 32u) { if (r < (n1 & 15u)) dp1[f1 + r] = sp1[f1 + r]; }
  };
  auto coop_flush = [&](bool all) __attribute__((always_inline)) {
    if (!all && __ballot(pendBytes != 0u && pend2Bytes != 0u) == 0ull) return;
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
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l0), d0 = (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l0);
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l0);
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l1), d1 = (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l1);
        const uint32_t n1 = (l1 != l0) ? (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l1) : 0u;
        coop_copy2(l0, s0, d0, n0, l1, s1, d1, n1);
      }
    }
    pendBytes = 0u; pend2Bytes = 0u;
  };
  // match bits of the 16 bytes at p: bit k = (d[p + k] == sym)
  auto match16 = [&](uint32_t p) __attribute__((always_inline)) -> uint32_t {
    u32x4 x;
    if (__builtin_expect(in_ring(p), 1)) x = ring_win(p);
    else
    {
      lds_st128(accScratch + lane * 16u, global_window16(in, blockAt, U, p));
      x = lds_ld128(accScratch + lane * 16u);
    }
    const uint32_t zm16 = zero_mask16(x.x ^ sym4, x.y ^ sym4, x.z ^ sym4, x.w ^ sym4);
    return zm16;
  };

  // ---- stream header: sizes, mode = single, the symbol ----
  if (active)
  {
    h32(n); h32(0);
    if constexpr (!SSHORT) hb(1);                                         // mode = single
    hb(sym);
    hflush();
  }

  // ---- input top-up (as k_encode8_blocks) ----
  auto publish = [&](uint32_t v) __attribute__((always_inline)) { rinfo[(lane % (uint32_t)RPL) * (uint32_t)LPR + lane / (uint32_t)RPL] = v; };
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0;
  uint32_t landedAt = 0, landedChunks = 0;
  uint64_t M0 = 0, M1 = 0;
  uint32_t mEnd = 0;

  // ---- the scanner (rle8_extreme_cpu.h:1140-1312) ----
  const int32_t end = (int32_t)n - 16;
  int32_t i = 0, count = 0, lastRLE = 0, wasted = 0, firstW = 0;
  bool searching = false;                    // in the "find the next candidate block" loop
  bool bodyDone = !(0 < end);

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
    // the ring keeps the chunk the scanner stands in and the one before it; a scanner that was sent back behind the ring (it reads global
    // memory there) gets no new bytes until it has caught up
    const uint32_t pos = (uint32_t)i;
    const uint32_t keep = ((pos & ~15u) >= 16u) ? (pos & ~15u) - 16u : 0u;
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
  // Match bits of the last 128 landed positions, [mEnd - 128, mEnd), in two registers: the scanner's windows are bit fields of them (16
  // bytes of ring + funnel + compare per window was most of the kernel).  Called behind every land(), once the row's chunks are visible.
  auto extend_masks = [&]() __attribute__((always_inline)) {
    uint64_t fresh = 0;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)LPR; j++)
    {
      const uint32_t at = landedAt + 16u * j;
      const u32x4 x = lds_ld128(hist + (hbase ^ (at & HM)));
      const uint32_t zm16 = zero_mask16(x.x ^ sym4, x.y ^ sym4, x.z ^ sym4, x.w ^ sym4);
      uint32_t bits = zm16;
      const uint32_t left = (n > at) ? n - at : 0u;                     // bytes at or beyond n never match
      if (left < 16u) bits &= (1u << left) - 1u;
      if (j >= landedChunks) bits = 0u;
      fresh |= (uint64_t)bits << (16u * j);
    }
    const uint32_t sh = 16u * landedChunks;                             // 0, 16, 32, 48 or 64
    if (sh == 64u) { M0 = M1; M1 = fresh; }
    else if (sh != 0u)
    {
      M0 = (M0 >> sh) | (M1 << (64u - sh));
      M1 = (M1 >> sh) | (fresh << (64u - sh));
    }
    mEnd += sh;
  };
  // match bits of the 16 bytes at pos: from the registers, or -- a scanner that was sent back -- from the ring / global memory
  auto win16 = [&](int32_t pos) __attribute__((always_inline)) -> uint32_t {
    const int32_t sft = pos - ((int32_t)mEnd - 128);
    if (__builtin_expect(sft >= 0, 1))
    {
      const uint32_t u = (uint32_t)sft;
      const uint64_t v = (u < 64u) ? ((M0 >> u) | ((u != 0u) ? (M1 << (64u - u)) : 0ull)) : (M1 >> (u - 64u));
      return (uint32_t)v & 0xFFFFu;
    }
    return match16((uint32_t)pos);
  };

  // 64 match bits from pos on (bits behind the landed positions read 0); wmax: how many 16-bit windows the value is good for
  auto win64 = [&](int32_t pos, uint32_t &wmax) __attribute__((always_inline)) -> uint64_t {
    const int32_t sft = pos - ((int32_t)mEnd - 128);
    if (__builtin_expect(sft >= 0, 1))
    {
      const uint32_t u = (uint32_t)sft;
      wmax = 4u;
      return (u < 64u) ? ((M0 >> u) | ((u != 0u) ? (M1 << (64u - u)) : 0ull)) : (M1 >> (u - 64u));
    }
    wmax = 1u;
    return (uint64_t)match16((uint32_t)pos);
  };

  // kind 0: short form (range in one byte), 1: long form, 2: the forced packet of the back-track (count byte without the 32 bit escape)
  // rle8_single_short: is the run [at - cnt, at) stored (process_symbol's penalty rule, rleX_Xsl_short.h:152-197)?
  auto short_stored = [&](int32_t cnt, int32_t gap) __attribute__((always_inline)) -> bool {
    const uint32_t range = (uint32_t)gap + 2u;
    const int32_t sc = cnt - (int32_t)TRS::SMINS + 2;
    const bool pack1 = (uint32_t)gap <= TRS::SMAXPR && (uint32_t)(sc - 2) <= TRS::SMAXPC;
    uint32_t pen = 0u;
    if (!pack1)
    {
      pen = 2u;
      if (!(sc <= (int32_t)TRS::SMAXTC && range <= TRS::SMAXTR))
        pen += ((range <= 0xFFFFFu) ? (range <= TRS::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TRS::SMAXTC ? 0u : 2u) : 4u);
    }
    return cnt >= (int32_t)TRS::SMINL || cnt >= (int32_t)(TRS::SMINS + pen);
  };
  auto emit = [&](int kind, int32_t at, int32_t cnt, int32_t from) __attribute__((always_inline)) {
    if constexpr (SSHORT)
    {
      // one-byte header [count | range] or the three-byte form with its 16 / 32 bit extensions (:199-357); no symbol
      const uint32_t gap = (uint32_t)(at - from - cnt), range = gap + 2u;
      const int32_t sc = cnt - (int32_t)TRS::SMINS + 2;
      if (gap <= TRS::SMAXPR && (uint32_t)(sc - 2) <= TRS::SMAXPC)
        hb(((uint32_t)(sc - 2) << TRS::SRBP) | gap);
      else
      {
        const uint32_t scu = (uint32_t)sc;
        const uint32_t scx = (scu <= TRS::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
        const uint32_t rx = (range <= TRS::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
        hb((TRS::SCINV << TRS::SRBP) | ((scx << (TRS::SRB - 8u)) >> 8));
        hb((scx << (TRS::SRB - 8u)) | (rx >> 8));
        hb(rx);
        if (scx != scu) { if (scu <= 0xFFFFu) { hb(scu); hb(scu >> 8); } else h32(scu); }
        if (rx != range) { if (range <= 0xFFFFu) { hb(range); hb(range >> 8); } else h32(range); }
      }
      hflush();
      emit_literals((uint32_t)from, (uint32_t)(at - cnt - from));
      return;
    }
    const uint32_t range = (uint32_t)(at - from - cnt + 1);
    const uint32_t c = (uint32_t)(cnt - SHORT + 1);
    if (kind == 2) hb(c);
    else if (c <= 255u) hb(c);
    else { hb(0); h32(c); }
    if (kind == 0) hb(range);
    else { hb(0); h32(range); }
    hflush();
    emit_literals((uint32_t)from, (uint32_t)(at - cnt - from));
  };
  // Writing a packet is ~250 instructions, and where few runs are stored every lane comes to it in another trip: one lane active, all
  // others waiting (measured: 29 k of the 58 k vector instructions of a wave at 1/64 utilisation).  The DECISION only needs lastRLE, so
  // the scanner notes the packet (queue of one + the one it stands at) and goes on; the packets are written when a lane comes to its
  // second one, when half of the lanes have one, or at the end -- by all lanes that have one, together.
  int32_t qKind = -1, qAt = 0, qCnt = 0, qFrom = 0;      // noted packet
  int32_t bKind = -1, bAt = 0, bCnt = 0, bFrom = 0;      // the packet the lane is blocked at (queue full)

  issue();
  land();
  wave_sync();
  extend_masks();

  uint32_t stepsLeft = 2u * (B / (uint32_t)Q) + 64u;
  uint32_t tripsLeft = 8u * B + 1024u;       // the back-track re-scans at most 255 bytes per three runs: bounded, and bounded here again

  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    issue();

    // One trip of the wave = per lane: the search up to the next candidate, the run through its full windows, ONE run end (the only part
    // that may write a packet).  (No lane is masked off around this loop: the copies of coop_flush are done by all 64 lanes; finished and
    // unused lanes have bodyDone set.)
    for (;;)
    {
      const bool can = !bodyDone && bKind < 0 && (uint32_t)i + 16u <= avail && tripsLeft != 0u;
      if (__ballot(can) == 0ull)
      {
        // nobody can scan on.  Write the noted packets?  (always at the end of the input: the tail needs an empty queue)
        const uint64_t noted = __ballot(qKind >= 0);
        if (noted == 0ull) break;
        if (__ballot(bKind >= 0 || (qKind >= 0 && (bodyDone || avail >= n))) == 0ull && __builtin_popcountll(noted) < 32) break;
        while (__ballot(qKind >= 0) != 0ull)
        {
          if (qKind >= 0)
          {
            emit(qKind, qAt, qCnt, qFrom);
            qKind = bKind; qAt = bAt; qCnt = bCnt; qFrom = bFrom;
            bKind = -1;
          }
          coop_flush(false);
        }
        continue;
      }
      if (can)
      {
        // the windows at i, i + 16, ... that may be looked at now: their 16 bytes have landed and they start in front of `end`
        auto windows_at = [&](int32_t pos, uint32_t wmax) __attribute__((always_inline)) -> uint32_t {
          const uint32_t wa = (avail - (uint32_t)pos) >> 4, we = ((uint32_t)(end - pos) + 15u) >> 4;
          return umin(wmax, umin(wa, we));
        };
        if (searching)
        {
          // find the next candidate block (:1291-1312): windows without the symbol, or with fewer than SHORT of it and not at their end,
          // are skipped -- up to four windows per pass, from one 64-bit field of the match masks
          if (!(i < end)) { searching = false; i += 1; }
          else
          {
            uint32_t wmax;
            const uint64_t v = win64(i, wmax);
            const uint32_t w = windows_at(i, wmax);
            tripsLeft--;
            uint32_t hit = 4u, mh = 0u;                                   // the first window that is not skipped
#pragma unroll
            for (int k = 3; k >= 0; k--)
            {
              const uint32_t m = (uint32_t)(v >> (16 * k)) & 0xFFFFu;
              const bool skip = m == 0u || ((m & 0x8000u) == 0u && (uint32_t)__builtin_popcount(m) < (uint32_t)SHORT);
              if (!skip && (uint32_t)k < w) { hit = (uint32_t)k; mh = m; }
            }
            if (hit < 4u)
            {
              i += (int32_t)(16u * hit + (uint32_t)__builtin_ctz(mh) + 1u);
              count = 1;
              searching = false;
            }
            else
            {
              i += (int32_t)(16u * w);
              if (!(i < end)) { searching = false; i += 1; }
            }
          }
        }
        if (!searching && i < end && (uint32_t)i + 16u <= avail && tripsLeft != 0u)
        {
          // the run through its full windows (count += 16 each: i += 15 and the loop's i++), then the window where it ends
          uint32_t wmax;
          const uint64_t v = win64(i, wmax);
          const uint32_t w = windows_at(i, wmax);
          tripsLeft--;
          const uint32_t t = (wmax == 1u) ? (uint32_t)__builtin_ctz(~(uint32_t)v | 0x10000u) : (v == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~v));
          uint32_t m = 0xFFFFu;
          if (t >= 16u * w) { count += (int32_t)(16u * w); i += (int32_t)(16u * w); }
          else
          {
            const uint32_t full = t & ~15u;
            m = (uint32_t)(v >> full) & 0xFFFFu;
            count += (int32_t)full; i += (int32_t)full;
          }
          if (m != 0xFFFFu)
          {
            tripsLeft--;
            if (m != 0u || count > 1)
            {
              const int32_t z = (int32_t)__builtin_ctz(~m);
              count += z; i += z;
              const int32_t range = i - lastRLE - count + 1;
              int ek = -1;                                               // the packet to write: one emit site (the code is inlined)
              if constexpr (SSHORT)
              {
                if (short_stored(count, i - lastRLE - count)) ek = 3;
              }
              else if (count >= SHORT)
              {
                if (range <= 255) { ek = 0; wasted = 0; }
                else if (count >= LONG || (PACKED && count - SHORT + 1 <= 255 && count >= MEDIUM)) { ek = 1; wasted = 0; }
                else
                {
                  wasted++;
                  if (wasted == 1 || i - firstW > 255) { firstW = i - count; wasted = 1; }
                  else if (wasted > 2)
                  {
                    // back to the first skipped run: it is stored with a long range whatever its length (:1244-1285)
                    i = firstW; wasted = 0; count = 0;
                    for (;;)
                    {
                      if (!(i < end)) break;
                      const uint32_t mm = win16(i);
                      int32_t zz = (int32_t)__builtin_ctz(~mm | 0x10000u);
                      if (zz > end - i) zz = end - i;
                      count += zz; i += zz;
                      if (zz < 16) break;
                    }
                    ek = 2;
                  }
                }
              }
              if (ek >= 0)
              {
                if (qKind < 0) { qKind = ek; qAt = i; qCnt = count; qFrom = lastRLE; }
                else { bKind = ek; bAt = i; bCnt = count; bFrom = lastRLE; }
                lastRLE = i;
              }
            }
            count = 0;
            searching = true;                                            // the search starts at this i (if i < end)
            if (!(i < end)) { searching = false; i += 1; }
          }
        }
        if (!searching && !(i < end)) bodyDone = true;
      }
    }

    // scalar tail (:392-470: no wasted-chances logic, no MEDIUM clause) and final block (:472-560), in rounds of at most one stored run
    // per lane so that the wave can copy the noted literal stretches in between
    {
      bool inTail = !finished && bodyDone && avail >= n && qKind < 0;
      while (__ballot(inTail) != 0ull)
      {
        if (inTail)
        {
          int ek = -1;
          int32_t at = 0, cnt = 0;
          while (ek < 0 && i < (int32_t)n)
          {
            const uint32_t v = hist[hbase ^ ((uint32_t)i & HM)];
            if (v == sym) count++;
            else
            {
              const int32_t range = i - lastRLE - count + 1;
              if constexpr (SSHORT) { if (short_stored(count, i - lastRLE - count)) ek = 3; }
              else if (range <= 255 && count >= SHORT) ek = 0;
              else if (count >= LONG) ek = 1;
              at = i; cnt = count;
              count = 0;
            }
            i++;
          }
          const bool fin = ek < 0;                                       // the input is through: the pending run, then the terminator
          int32_t frange = 0;
          if (fin)
          {
            frange = i - lastRLE - count + 1;
            if constexpr (SSHORT) { if (short_stored(count, i - lastRLE - count)) ek = 3; }
            else if (frange <= 255 && count >= SHORT) ek = 0;
            else if (count >= LONG) ek = 1;
            at = i; cnt = count;
          }
          if (ek >= 0) { emit(ek, at, cnt, lastRLE); lastRLE = at; }
          if (fin)
          {
            if constexpr (SSHORT)
            {
              // terminators of the Short family (rleX_Xsl_short.h:470-523)
              if (ek >= 0) { hb(TRS::SCINV << TRS::SRBP); hb(TRS::STB); hb(1); hb(0); hb(0); hb(0); hb(0); hflush(); }
              else
              {
                const uint32_t kLit = (uint32_t)(i - lastRLE);
                hb(TRS::SCINV << TRS::SRBP); hb(TRS::STB); hb(0); hb(0); hb(0); h32(kLit + 2u);
                hflush();
                emit_literals((uint32_t)lastRLE, kLit);
              }
            }
            else if (ek >= 0) { hb(0); h32(0); hb(0); h32(0); hflush(); }
            else
            {
              hb(0); h32(0); hb(0); h32((uint32_t)(frange + count));
              hflush();
              emit_literals((uint32_t)lastRLE, (uint32_t)(i - lastRLE));
            }
            if ((opos & 15u) > accLow)
            {
              if (accLow == 0u) st128(slot + (opos & ~15u), oacc);
              else store_bytes(slot + (opos & ~15u), oacc, accLow, opos & 15u);
            }
            st32(slot + 4, opos);
            sizes[b] = opos;
            finished = true;
            inTail = false;
          }
        }
        coop_flush(false);
      }
    }

    wave_sync();
    land();
    wave_sync();
    extend_masks();
  }
  coop_flush(true);
#undef HS_SMERGE1
}

} // namespace hsrle
