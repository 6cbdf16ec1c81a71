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
//                     against the oracle on 90 000 inputs: tools/scratch/pick_model.py.)  So: equality bits of the whole block in
//                     LDS, every lane takes the run starts of 64 positions, LDS atomics into a 256-entry table, the one run that
//                     reaches n - 16 and the final registration by lane 0, argmax by shuffles.  The symbol goes to byte 9 of the
//                     block's staging slot -- where the stream keeps it -- and the encoder picks it up there.
//   k_encode8_single_blocks<PACKED>   one lane per block like the other ring encoders (hsrle_encode8.hip.h: same top-up, same
//                     output accumulator).  The scanner of the reference is run as it is -- 16-byte windows at its own, data
//                     dependent phase, the skip rule of the search, the back-track to the first wasted run -- on match bits taken
//                     from the ring (from global memory for positions that have left it).
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low_m, wave_sync

namespace hsrle {

constexpr uint32_t kSinglePickMaxBlock = 32768u;   // larger blocks use the first-generation kernel (hsrle_encode.hip.h)

// dynamic LDS: [0, 1024) table (prob | pcount << 16), then n bytes of the block (padded to 64), then the equality bits
__global__ __launch_bounds__(64) void k_single_pick(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *__restrict__ slots, uint32_t slotStride)
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
      const uint32_t z0 = zero_bytes(x.x ^ alignbyte(x.y, x.x, 1)), z1 = zero_bytes(x.y ^ alignbyte(x.z, x.y, 1));
      const uint32_t z2 = zero_bytes(x.z ^ alignbyte(x.w, x.z, 1)), z3 = zero_bytes(x.w ^ alignbyte(x4, x.w, 1));
      const uint32_t b0 = (((z0 >> 7) * 0x00204081u) >> 21) & 0xFu, b1 = (((z1 >> 7) * 0x00204081u) >> 21) & 0xFu;
      const uint32_t b2 = (((z2 >> 7) * 0x00204081u) >> 21) & 0xFu, b3 = (((z3 >> 7) * 0x00204081u) >> 21) & 0xFu;
      e64 |= (uint64_t)(b0 | (b1 << 4) | (b2 << 8) | (b3 << 12)) << (16u * j);
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
  if (lane == 0u)
    slots[(uint64_t)b * slotStride + 9u] = (bestKey >> 8) != 0u ? (uint8_t)(255u - (bestKey & 0xFFu)) : (uint8_t)0;
}

template <bool PACKED>
__global__ __launch_bounds__(64) void k_encode8_single_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *__restrict__ slots, uint32_t slotStride,
                                                              uint32_t *__restrict__ sizes)
{
  constexpr int Q = 64;                      // input bytes per lane and step
  constexpr int H = 256;                     // history ring per lane
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr int32_t SHORT = PACKED ? 2 : 4, MEDIUM = 6, LONG = PACKED ? 10 : 8;

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
  auto hsw_of = [](uint32_t r) -> uint32_t { return (r & 7u) << 4; };
  const uint32_t hbase = (lane * (uint32_t)H) ^ hsw_of(lane);

  uint32_t n = 0;
  const uint64_t blockAt = (uint64_t)b * B;
  if (active) n = (uint32_t)((U - blockAt) < (uint64_t)B ? (U - blockAt) : (uint64_t)B);
  uint8_t *const slot = slots + (uint64_t)b * slotStride;
  const uint32_t sym = active ? (uint32_t)slot[9] : 0u;                 // k_single_pick's choice
  const uint32_t sym4 = sym * 0x01010101u;

  uint32_t avail = 0;
  bool finished = !active;

  // ---- output: 16-byte accumulator + stream position (as k_encode8_blocks) ----
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  u32x4 oacc = zero4;
  uint32_t opos = 0;
  auto append = [&](u32x4 hv, uint32_t nb) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);
    const u32x4 w = HS_SMERGE1(oacc, lowp, c);
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
  auto hpush = [&](uint32_t v, uint32_t k) {
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
  auto hb = [&](uint32_t v) { hpush(v & 0xFFu, 1u); };
  auto h32 = [&](uint32_t v) { hpush(v, 4u); };
  auto hflush = [&]() {
    append(u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), hhi, 0u }, hn);
    hlo = 0; hhi = 0; hn = 0;
  };
  auto ring_win = [&](uint32_t p) -> u32x4 {
    const uint32_t a0 = p & ~15u;
    return funnel16(lds_ld128(hist + (hbase ^ (a0 & HM))), lds_ld128(hist + (hbase ^ ((a0 + 16u) & HM))), p & 15u);
  };
  // is the 16-byte window at block position p still in the ring?
  auto in_ring = [&](uint32_t p) -> bool { return p + (uint32_t)H >= avail + 16u; };
  // Literal bytes [from, from + len) of the block.  The Single codecs store few runs, so literal stretches are long and have mostly left the
  // ring when their packet is written.  Fetching them 16 bytes at a time per lane pays one memory latency per chunk (a 4 KiB block without a
  // stored run: 256 of them -- 4x the time of everything else), so the 16-byte aligned middle of such a stretch is only NOTED here
  // (pendSrc / pendDst / pendBytes) and copied by the whole wave at the next point where the wave is converged (coop_flush): 1 KiB per
  // load / store pair.  At most one stretch is pending per lane: every trip of the scanner and every round of the tail stores at most one run.
  uint32_t pendSrc = 0, pendDst = 0, pendBytes = 0;
  auto emit_literals = [&](uint32_t from, uint32_t len) {
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
        st128(dst + k, w);
        k += 16u;
        if (k < total) w = ring_win(srcp + k);
      }
      oacc = w;
      opos += len;
    }
    else
    {
      // up to the next 16-byte boundary of the stream | whole chunks, noted for the wave | the rest
      const uint32_t head = umin(len, (16u - (opos & 15u)) & 15u);
      const uint32_t mid = ((len - head) & ~15u) >= 64u ? ((len - head) & ~15u) : 0u;
#pragma unroll 1
      for (uint32_t piece = 0; piece < 2u; piece++)
      {
        uint32_t f = from, l = (mid != 0u) ? head : len;
        if (piece == 1u)
        {
          if (mid == 0u) break;
          pendSrc = from + head; pendDst = opos; pendBytes = mid;
          opos += mid;                                                  // (a multiple of 16: the accumulator is empty here and stays so)
          f = from + head + mid; l = len - head - mid;
        }
        if (l == 0u) continue;
        const uint32_t c = opos & 15u, total = c + l;
        const uint32_t srcp = f - c;
        uint8_t *const dst = slot + (opos & ~15u);
        u32x4 w = HS_SMERGE1(oacc, global_window16(in, blockAt, U, srcp), c);
        uint32_t k = 0;
        while (k + 16u <= total)
        {
          st128(dst + k, w);
          k += 16u;
          if (k < total) w = global_window16(in, blockAt, U, srcp + k);
        }
        lds_st128(accScratch + lane * 16u, w);                          // (see k_encode8_blocks: keeps vmcnt waits out of the common path)
        oacc = lds_ld128(accScratch + lane * 16u);
        opos += l;
      }
    }
  };
  // wave-converged: copy the noted stretches, one after the other, 16 bytes per lane and trip (source at any alignment)
  auto coop_flush = [&]() {
    uint64_t todo = __ballot(pendBytes != 0u);
    while (todo != 0ull)
    {
      const uint32_t l = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)pendSrc, (int)l), dstOff = (uint32_t)__builtin_amdgcn_readlane((int)pendDst, (int)l);
      const uint32_t bytes = (uint32_t)__builtin_amdgcn_readlane((int)pendBytes, (int)l);
      const uint8_t *const sp = in + (uint64_t)(wgFirst + l) * B + src;
      uint8_t *const dp = slots + (uint64_t)(wgFirst + l) * slotStride + dstOff;
      for (uint32_t k = lane * 16u; k < bytes; k += 4096u)
      {
        // four loads in flight (bytes is a multiple of 16)
        u32x4 v[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++)
          if (k + q * 1024u < bytes) v[q] = ld128(sp + k + q * 1024u);
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++)
          if (k + q * 1024u < bytes) st128(dp + k + q * 1024u, v[q]);
      }
    }
    pendBytes = 0u;
  };
  // match bits of the 16 bytes at p: bit k = (d[p + k] == sym)
  auto match16 = [&](uint32_t p) -> uint32_t {
    u32x4 x;
    if (__builtin_expect(in_ring(p), 1)) x = ring_win(p);
    else
    {
      lds_st128(accScratch + lane * 16u, global_window16(in, blockAt, U, p));
      x = lds_ld128(accScratch + lane * 16u);
    }
    const uint32_t z0 = zero_bytes(x.x ^ sym4), z1 = zero_bytes(x.y ^ sym4), z2 = zero_bytes(x.z ^ sym4), z3 = zero_bytes(x.w ^ sym4);
    const uint32_t b0 = (((z0 >> 7) * 0x00204081u) >> 21) & 0xFu, b1 = (((z1 >> 7) * 0x00204081u) >> 21) & 0xFu;
    const uint32_t b2 = (((z2 >> 7) * 0x00204081u) >> 21) & 0xFu, b3 = (((z3 >> 7) * 0x00204081u) >> 21) & 0xFu;
    return b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
  };

  // ---- stream header: sizes, mode = single, the symbol ----
  if (active)
  {
    h32(n); h32(0); hb(1); hb(sym);
    hflush();
  }

  // ---- input top-up (as k_encode8_blocks) ----
  auto publish = [&](uint32_t v) { rinfo[(lane % (uint32_t)RPL) * (uint32_t)LPR + lane / (uint32_t)RPL] = v; };
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0;

  // ---- the scanner (rle8_extreme_cpu.h:1140-1312) ----
  const int32_t end = (int32_t)n - 16;
  int32_t i = 0, count = 0, lastRLE = 0, wasted = 0, firstW = 0;
  bool searching = false;                    // in the "find the next candidate block" loop
  bool bodyDone = !(0 < end);

  auto issue = [&]() {
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
        {
          uint32_t t[4] = { 0, 0, 0, 0 };
          for (uint32_t k = 0; k < 16u && g + k < U; k++)
            t[k >> 2] |= (uint32_t)in[g + k] << (8u * (k & 3u));
          v = u32x4{ t[0], t[1], t[2], t[3] };
        }
      }
      pf[q] = v;
      pfAt[q] = (r * (uint32_t)H) ^ hsw_of(r) ^ ((e + c * 16u) & HM);
    }
  };
  auto land = [&]() {
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
    avail = umin(avail + (take << 4), n);
  };

  // kind 0: short form (range in one byte), 1: long form, 2: the forced packet of the back-track (count byte without the 32 bit escape)
  auto emit = [&](int kind, int32_t at, int32_t cnt) {
    const uint32_t range = (uint32_t)(at - lastRLE - cnt + 1);
    const uint32_t c = (uint32_t)(cnt - SHORT + 1);
    if (kind == 2) hb(c);
    else if (c <= 255u) hb(c);
    else { hb(0); h32(c); }
    if (kind == 0) hb(range);
    else { hb(0); h32(range); }
    hflush();
    emit_literals((uint32_t)lastRLE, (uint32_t)(at - cnt - lastRLE));
    lastRLE = at;
  };

  issue();
  land();
  wave_sync();

  uint32_t stepsLeft = 2u * (B / (uint32_t)Q) + 64u;
  uint32_t tripsLeft = 8u * B + 1024u;       // the back-track re-scans at most 255 bytes per three runs: bounded, and bounded here again

  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    issue();

    // (no lane is masked off around this loop: the copies of coop_flush are done by all 64 lanes; finished and unused lanes have bodyDone set)
    {
      for (;;)
      {
        const bool can = !bodyDone && (uint32_t)i + 16u <= avail && tripsLeft != 0u;
        if (__ballot(can) == 0ull) break;
        if (can)
        {
          tripsLeft--;
          const uint32_t m = match16((uint32_t)i);
          if (!searching)
          {
            if (m == 0xFFFFu) { count += 16; i += 16; }                  // (i += 15 and the loop's i++)
            else
            {
              if (m != 0u || count > 1)
              {
                const int32_t z = (int32_t)__builtin_ctz(~m);
                count += z; i += z;
                const int32_t range = i - lastRLE - count + 1;
                int ek = -1;                                             // the packet to write: one emit site (the code is inlined)
                if (count >= SHORT)
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
                        const uint32_t mm = match16((uint32_t)i);
                        int32_t zz = (int32_t)__builtin_ctz(~mm | 0x10000u);
                        if (zz > end - i) zz = end - i;
                        count += zz; i += zz;
                        if (zz < 16) break;
                      }
                      ek = 2;
                    }
                  }
                }
                if (ek >= 0) emit(ek, i, count);
              }
              count = 0;
              searching = true;                                          // the search starts at this i, in the next trip (if i < end)
              if (!(i < end)) { searching = false; i += 1; }
            }
          }
          else
          {
            const uint32_t pop = (uint32_t)__builtin_popcount(m);
            if (m == 0u || ((m & 0x8000u) == 0u && pop < (uint32_t)SHORT))
            {
              i += 16;
              if (!(i < end)) { searching = false; i += 1; }
            }
            else
            {
              i += (int32_t)__builtin_ctz(m);
              count = 1;
              searching = false;
              i += 1;
            }
          }
          if (!searching && !(i < end)) bodyDone = true;
        }
        coop_flush();
      }

    }

    // scalar tail (:392-470: no wasted-chances logic, no MEDIUM clause) and final block (:472-560), in rounds of at most one stored run
    // per lane so that the wave can copy the noted literal stretches in between
    {
      bool inTail = !finished && bodyDone && avail >= n;
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
              if (range <= 255 && count >= SHORT) ek = 0;
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
            if (frange <= 255 && count >= SHORT) ek = 0;
            else if (count >= LONG) ek = 1;
            at = i; cnt = count;
          }
          if (ek >= 0) emit(ek, at, cnt);
          if (fin)
          {
            if (ek >= 0) { hb(0); h32(0); hb(0); h32(0); hflush(); }
            else
            {
              hb(0); h32(0); hb(0); h32((uint32_t)(frange + count));
              hflush();
              emit_literals((uint32_t)lastRLE, (uint32_t)(i - lastRLE));
            }
            if ((opos & 15u) != 0u)
              st128(slot + (opos & ~15u), oacc);
            st32(slot + 4, opos);
            sizes[b] = opos;
            finished = true;
            inTail = false;
          }
        }
        coop_flush();
      }
    }

    wave_sync();
    land();
    wave_sync();
  }
#undef HS_SMERGE1
}

} // namespace hsrle
