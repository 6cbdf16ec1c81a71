// experiments/: NOT PART OF THE SHIPPED LIBRARY (-DHSRLE_EXPERIMENTS builds only: bit-exact, measured slower than the lane-per-block encoder)
// hsrle_encode8w.hip.h -- ONE WAVE PER BLOCK encoder for the two list-free 8 bit multi-symbol codecs (rle8_multi, rle8_packed_multi),
// blocks of at most 4096 bytes: the position-parallel form of the reference's scan (src/rle8_extreme_cpu.h:936-1099: cmpeq + movemask
// + ctz over the input, emit rule :974-1001, scalar tail :111-199, final block :203-338).
//
// Replaces the lane-per-block kernel (hsrle_encode8.hip.h) for these two codecs, and with it the second pass over the compressed
// bytes: that kernel wrote 16-byte pieces into per-block staging slots (WRITE_SIZE 1.71 x the payload: partial lines) and k_compact
// moved them to their place -- 26 .. 31 GB of HBM traffic for 13.3 GB algorithmic (profiles/r02_traffic.json).  Here
//   * a lane holds 64 consecutive input bytes of the block in registers; the equality of neighbouring bytes is 64 bits per lane (SWAR),
//     run starts / run ends are bit masks, runs that cross lanes are stitched with one carry bit and a prefix maximum of start positions;
//   * the emit decisions are a chain through (lastRLE, lastSymbol), but a run of >= 11 (Packed) / 6 (plain) bytes is stored whatever
//     the state: a lane whose bytes hold the end of such a run knows its OUTGOING state without its incoming one, a lane
//     without a storable run passes the state through, and only the few lanes in between wait for their left neighbour -- a wave scan over
//     (kind, state), repeated while such lanes are left (0 .. 2 times on the synthetic buffers);
//   * with its incoming state a lane sizes, then writes its packets: headers from registers, literals from the LDS copy of the input,
//     both through a 16-byte accumulator into an LDS image of the block's stream (whole chunks are stored, shared edge chunks are OR-ed
//     into the zeroed image; literal stretches of 64 bytes and more are copied by the whole wave);
//   * the stream's place in the payload comes from a decoupled look-back over the block sizes (one 64 bit word per block: aggregate /
//     inclusive prefix), so the image leaves LDS exactly once, as whole 16-byte stores to its final position.
// Blocks are taken in ticket order by resident waves (a block's predecessors are always finished or in flight: no deadlock).
#pragma once

#include "../hsrle_common.hip.h"
#include "../hsrle_decode.hip.h" // funnel16, lds_read16, wave_sync

namespace hsrle {

constexpr uint32_t kWaveEncodeMaxBlock = 4096u;
constexpr uint32_t kTicketCounters = 64u, kTicketStride = 1024u;   // counters 4 KiB apart (dwords): [0] wave start order, [1 + c] blocks of counter c
constexpr uint64_t kTicketBytes = 4ull * kTicketStride * (kTicketCounters + 1u);
constexpr unsigned long long TILE_AGGREGATE = 1ull << 62, TILE_PREFIX = 2ull << 62, TILE_VALUE = (1ull << 62) - 1ull;

__device__ __forceinline__ u32x4 or4(u32x4 a, u32x4 b) { return u32x4{ a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w }; }

// the low m (0..16) bytes of v, the rest zero
__device__ __forceinline__ u32x4 keep_low(u32x4 v, uint32_t m)
{
  auto dm = [&](uint32_t j) -> uint32_t { const int32_t k = (int32_t)m - (int32_t)(4u * j); return k >= 4 ? ~0u : (k <= 0 ? 0u : ((1u << (8 * k)) - 1u)); };
  return u32x4{ v.x & dm(0), v.y & dm(1), v.z & dm(2), v.w & dm(3) };
}
// bytes [lo, hi) of v (0 <= lo <= hi <= 16), the rest zero
__device__ __forceinline__ u32x4 keep_range(u32x4 v, uint32_t lo, uint32_t hi)
{
  const u32x4 a = keep_low(v, hi), b = keep_low(v, lo);
  return u32x4{ a.x & ~b.x, a.y & ~b.y, a.z & ~b.z, a.w & ~b.w };
}

template <int FAM>
__global__ __launch_bounds__(64) void k_encode8_wave(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint64_t *__restrict__ offsets,
                                                     uint8_t *__restrict__ payload, unsigned long long *__restrict__ tiles, uint32_t *__restrict__ ticket)
{
  using TR = Traits<FAM, 1, 0>;
  static_assert(FAM == PLAIN || FAM == PACKED, "the two list-free 8 bit multi-symbol codecs");
  constexpr bool PK = TR::kPacked;
  constexpr uint32_t LONGC = PK ? 11u : 6u;          // runs this long are stored whatever the state
  constexpr uint32_t MINC = PK ? 3u : 6u;            // shorter runs are never stored
  constexpr uint32_t TERM = PK ? 9u : 11u;           // bytes of either terminator's fixed part
  constexpr uint32_t IN_PAD = 16u;                   // the input image starts 16 bytes into its buffer (a cooperative copy may look below its source)

  __shared__ __attribute__((aligned(16))) uint8_t inraw[IN_PAD + kWaveEncodeMaxBlock + 48];
  __shared__ __attribute__((aligned(16))) uint8_t outbuf[kWaveEncodeMaxBlock + 320];
  __shared__ uint32_t jobs[3 * 72];
  __shared__ uint32_t runs[kWaveEncodeMaxBlock / 3 + 8];   // at most one storable run per 3 input bytes
  __shared__ uint32_t jobCount;
  uint8_t *const inbuf = inraw + IN_PAD;
  const uint32_t lane = threadIdx.x;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };

  auto or_store = [&](uint32_t at, u32x4 v) {         // at: multiple of 16
    uint32_t *const p = reinterpret_cast<uint32_t *>(outbuf + at);
    if (v.x) atomicOr(p + 0, v.x);
    if (v.y) atomicOr(p + 1, v.y);
    if (v.z) atomicOr(p + 2, v.z);
    if (v.w) atomicOr(p + 3, v.w);
  };

  // cooperative copy inbuf[src, src + len) -> outbuf[dst, dst + len): whole destination chunks are stored, the two edge chunks OR-ed
  auto coop_copy = [&](uint32_t src, uint32_t dst, uint32_t len) {
    const uint32_t d0 = dst & ~15u, dEnd = dst + len;
    for (uint32_t D = d0 + 16u * lane; D < dEnd; D += 16u * 64u)
    {
      const u32x4 v = lds_read16(inraw, IN_PAD + src + D - dst);        // (D - dst may be "negative": the pad in front of the image absorbs it)
      const uint32_t lo = (D < dst) ? dst - D : 0u, hi = (dEnd - D < 16u) ? dEnd - D : 16u;
      if (lo == 0u && hi == 16u) lds_st128(outbuf + D, v);
      else or_store(D, keep_range(v, lo, hi));
    }
  };

  // Blocks are taken in TICKET order, from kTicketCounters interleaved counters (counter c hands out the blocks c, c + 64, c + 128, ...):
  // one counter for all blocks serialises 2 M device-scope atomics on one address -- measured 11.5 ns each = the whole 24 ms of the first
  // version of this kernel, whatever the waves did in between.  A wave picks its counter by its START order (the first 64 waves to start
  // cover all counters, and they are resident by definition), so the smallest unfinished block always belongs to a counter that a running
  // wave serves: the look-back below never waits for a wave that has not started.  (A static assignment blockIdx + k * gridDim deadlocks
  // as soon as the grid is larger than what is really resident -- measured: the occupancy query over-states it.)
  uint32_t myCounter = 0;
  if (lane == 0) myCounter = atomicAdd(ticket, 1u) % kTicketCounters;
  myCounter = (uint32_t)__builtin_amdgcn_readfirstlane((int)myCounter);
  uint32_t *const myTicket = ticket + kTicketStride * (1u + myCounter);
  for (;;)
  {
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(myTicket, 1u);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    if (b > (0xFFFFFFFFu - myCounter) / kTicketCounters) break;
    b = b * kTicketCounters + myCounter;
    if (b >= nBlocks) break;

    const uint64_t at = (uint64_t)b * B;
    const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
    const uint8_t *const d = in + at;
    const uint32_t base = lane * 64u;

    // ---- the lane's 64 input bytes: registers + LDS image ----
    u32x4 x[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const uint32_t pos = base + 16u * j;
      u32x4 v = zero4;
      if (pos + 16u <= n) v = ld128(d + pos);
      else if (pos < n)
      {
        uint32_t t[4] = { 0, 0, 0, 0 };
        for (uint32_t k = 0; pos + k < n; k++) t[k >> 2] |= (uint32_t)d[pos + k] << (8u * (k & 3u));
        v = u32x4{ t[0], t[1], t[2], t[3] };
      }
      x[j] = v;
      lds_st128(inbuf + pos, v);
    }
    if (lane == 0) jobCount = 0;

    // ---- equality of neighbouring bytes, run starts / ends ----
    const uint32_t nextFirst = (uint32_t)__shfl_down((int)x[0].x, 1, 64);
    uint64_t e64 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const u32x4 a = x[j];
      const uint32_t s = (j < 3u) ? x[j < 3u ? j + 1u : 3u].x : nextFirst;
      const uint32_t z0 = zero_bytes(a.x ^ alignbyte(a.y, a.x, 1)), z1 = zero_bytes(a.y ^ alignbyte(a.z, a.y, 1));
      const uint32_t z2 = zero_bytes(a.z ^ alignbyte(a.w, a.z, 1)), z3 = zero_bytes(a.w ^ alignbyte(s, a.w, 1));
      const uint32_t b0 = (((z0 >> 7) * 0x00204081u) >> 21) & 0xFu, b1 = (((z1 >> 7) * 0x00204081u) >> 21) & 0xFu;
      const uint32_t b2 = (((z2 >> 7) * 0x00204081u) >> 21) & 0xFu, b3 = (((z3 >> 7) * 0x00204081u) >> 21) & 0xFu;
      e64 |= (uint64_t)(b0 | (b1 << 4) | (b2 << 8) | (b3 << 12)) << (16u * j);
    }
    // position i matches only if its successor exists (bytes at or beyond n never match)
    const uint32_t validBits = (n > base + 1u) ? ((n - 1u - base) < 64u ? (n - 1u - base) : 64u) : 0u;
    e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint32_t hiPrev = (uint32_t)__shfl_up((int)(uint32_t)(e64 >> 32), 1, 64);
    const uint64_t carry = (lane > 0u) ? (uint64_t)(hiPrev >> 31) : 0ull;
    const uint64_t prev = (e64 << 1) | carry;
    const uint64_t starts = e64 & ~prev;
    const uint64_t ends = ~e64 & prev;                                   // bit i: a run's last byte is base + i

    // start of the run that is open when this lane begins: the last start in the lanes in front (prefix maximum)
    int32_t lastStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
    {
      int32_t v = lastStart;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1)
      {
        const int32_t y = __shfl_up(v, dd, 64);
        if ((int)lane >= dd) v = (y > v) ? y : v;
      }
      lastStart = __shfl_up(v, 1, 64);
      if (lane == 0u) lastStart = -1;
    }
    const uint32_t carryStart = (uint32_t)lastStart;
    wave_sync();

#ifdef HSRLE_W8_ONLYLOAD   // timing experiment only: load + masks, nothing else
    if (ends == 0x123456789ull && starts == 77ull) offsets[b] = carryStart;
    continue;
#endif
    // ---- the runs that can be stored (count >= MINC), compacted in block order: runs[r] = start | end << 16 ----
    uint32_t myRuns = 0;
    {
      uint64_t m = ends;
      while (m != 0ull)
      {
        const uint32_t i = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint64_t sBelow = starts & ((i >= 63u) ? ~0ull : ((2ull << i) - 1ull));
        const uint32_t p = (sBelow != 0ull) ? base + 63u - (uint32_t)__builtin_clzll(sBelow) : carryStart;
        if (base + i + 1u - p >= MINC) myRuns++;
      }
    }
    uint32_t runAt = myRuns;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1)
    {
      const uint32_t y = (uint32_t)__shfl_up((int)runAt, dd, 64);
      if ((int)lane >= dd) runAt += y;
    }
    const uint32_t R = (uint32_t)__shfl((int)runAt, 63, 64);
    runAt -= myRuns;
    {
      uint64_t m = ends;
      while (m != 0ull)
      {
        const uint32_t i = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint64_t sBelow = starts & ((i >= 63u) ? ~0ull : ((2ull << i) - 1ull));
        const uint32_t p = (sBelow != 0ull) ? base + 63u - (uint32_t)__builtin_clzll(sBelow) : carryStart;
        const uint32_t e = base + i + 1u;
        if (e - p >= MINC) runs[runAt++] = p | (e << 16);
      }
    }
    // the image: zeroed, so that shared edge chunks can be OR-ed in
    for (uint32_t c = lane * 16u; c < sizeof(outbuf); c += 64u * 16u) lds_st128(outbuf + c, zero4);
    wave_sync();

    // a header / terminator of nb <= 12 bytes at image position dst (always OR-ed: it shares its chunks)
    auto put_small = [&](uint32_t dst, u32x4 hv, uint32_t nb) {
      const uint32_t c = dst & 15u;
      or_store(dst & ~15u, (c == 0u) ? hv : funnel16(zero4, hv, 16u - c));
      if (c + nb > 16u) or_store((dst & ~15u) + 16u, funnel16(hv, zero4, 16u - c));
    };
    // literals inbuf[src, src + len) -> image[dst, ..): by this lane (short) or, as a job, by the whole wave
    auto put_literals = [&](uint32_t src, uint32_t dst, uint32_t len) {
      if (len >= 64u)
      {
        const uint32_t slot = atomicAdd(&jobCount, 1u);
        jobs[3u * slot] = src; jobs[3u * slot + 1u] = dst; jobs[3u * slot + 2u] = len;
        return;
      }
      const uint32_t dEnd = dst + len;
      for (uint32_t D = dst & ~15u; D < dEnd; D += 16u)
      {
        const u32x4 v = lds_read16(inraw, IN_PAD + src + D - dst);
        const uint32_t lo = (D < dst) ? dst - D : 0u, hi = (dEnd - D < 16u) ? dEnd - D : 16u;
        if (lo == 0u && hi == 16u) lds_st128(outbuf + D, v);
        else or_store(D, keep_range(v, lo, hi));
      }
    };

#ifdef HSRLE_W8_ONLYRUNS   // timing experiment only: + run compaction and image zeroing
    if (R == 0x12345u) offsets[b] = runs[5];
    continue;
#endif
    // ---- one run per lane, 64 runs per round.  The state (lastRLE, lastSymbol) behind a run that is stored for sure is known at once;
    //      a run that MAY be stored needs the state behind its predecessor: those resolve in (few) passes of neighbour exchanges ----
    uint32_t carRLE = 0, carSym = 0;                  // state in front of the round's first run
    uint32_t imagePos = 9u;                           // where the round's first packet goes
    bool ended = false;
    for (uint32_t r0 = 0; r0 < R; r0 += 64u)
    {
      const bool have = r0 + lane < R;
      const uint32_t pe = have ? runs[r0 + lane] : 0u;
      const uint32_t p = pe & 0xFFFFu, e = pe >> 16, count = e - p;
      const uint32_t sym = have ? inbuf[p] : 0u;
      bool body = true;
      if constexpr (PK)
      {
        const int32_t kk = (int32_t)(count - 1u) / 32;                 // body / tail split of the canonical AVX2 encoder
        body = (e < n) && ((int32_t)p + 1 + 32 * kk < (int32_t)n - 32);
      }
      const bool sure = have && count >= LONGC;                        // stored whatever the state is
      // decision for a given incoming state
      auto decide = [&](uint32_t inR, uint32_t inS, bool &same) -> int {
        const uint32_t rng = p - inR + 1u;
        same = false;
        if constexpr (PK)
        {
          if (!body) return (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
          same = sym == inS;
          const bool emit = count >= 11u || (rng <= 127u && ((same && count >= 3u) || count >= 4u));
          return emit ? (rng <= 127u ? 1 : 2) : 0;
        }
        else
          return (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;
      };
      // outgoing state: known (1) for a sure run and for a lane without a run (= incoming, resolved below), else pending
      uint32_t outR = e, outS = sym;
      bool outKnown = sure || !have;                                   // (lanes behind the last run take no part)
      bool inKnown = !have;
      uint32_t inR = 0, inS = 0;
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        uint32_t lk = (uint32_t)__shfl_up((int)(outKnown ? 1u : 0u), 1, 64), lr = (uint32_t)__shfl_up((int)outR, 1, 64), ls = (uint32_t)__shfl_up((int)outS, 1, 64);
        if (lane == 0u) { lk = 1u; lr = carRLE; ls = carSym; }
        if (!inKnown && lk != 0u) { inKnown = true; inR = lr; inS = ls; }
        if (inKnown && !outKnown)
        {
          bool same;
          const int k = decide(inR, inS, same);
          if (k == 0) { outR = inR; outS = inS; }
          else { outR = e; outS = (PK && !body) ? inS : sym; }
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      // (a sure run in the tail region keeps lastSymbol: nothing reads it any more -- every later run is a tail run as well)

      // ---- sizes and places ----
      bool same = false;
      const int k = have ? decide(inR, inS, same) : 0;
      const uint32_t gap = p - inR, rng = gap + 1u;
      uint32_t hl = 0;
      if (k)
      {
        if constexpr (PK) hl = ((count - 2u <= 127u) ? 1u : 5u) + (same ? 0u : 1u) + (k == 1 ? 1u : 4u);
        else hl = 1u + ((count - 5u <= 255u) ? 1u : 5u) + (k == 1 ? 1u : 5u);
      }
      const bool endsBlock = k != 0 && e >= n;
      const uint32_t myBytes = k ? hl + gap + (endsBlock ? TERM : 0u) : 0u;
      uint32_t incl = myBytes;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1)
      {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, dd, 64);
        if ((int)lane >= dd) incl += y;
      }
      const uint32_t dst = imagePos + incl - myBytes;
      imagePos += (uint32_t)__shfl((int)incl, 63, 64);
      const int lastRun = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
      carRLE = (uint32_t)__shfl((int)outR, lastRun, 64);
      carSym = (uint32_t)__shfl((int)outS, lastRun, 64);
      if (__ballot(endsBlock) != 0ull) ended = true;

      // ---- the packet ----
#ifndef HSRLE_W8_NOEMIT     // timing experiment only (no packets written)
      if (k)
      {
        uint64_t hlo = 0; uint32_t hhi = 0, hn = 0;
        auto hpush = [&](uint32_t v, uint32_t kb) {
          const uint32_t sh = hn * 8u;
          if (hn < 8u) { hlo |= (uint64_t)v << sh; if (hn + kb > 8u) hhi |= v >> (64u - sh); }
          else hhi |= v << (sh - 64u);
          hn += kb;
        };
        if constexpr (PK)
        {
          const uint32_t c = count - 2u, sm = same ? 0x80u : 0u;
          if (c <= 127u) hpush(c | sm, 1u); else { hpush(sm, 1u); hpush(c, 4u); }
          if (!same) hpush(sym, 1u);
          if (k == 1) hpush((rng << 1) & 0xFFu, 1u); else hpush((rng << 1) | 1u, 4u);
        }
        else
        {
          const uint32_t c = count - 5u;
          hpush(sym, 1u);
          if (c <= 255u) hpush(c, 1u); else { hpush(0u, 1u); hpush(c, 4u); }
          if (k == 1) hpush(rng, 1u); else { hpush(0u, 1u); hpush(rng, 4u); }
        }
        put_small(dst, u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), hhi, 0u }, hn);
        if (gap != 0u) put_literals(inR, dst + hn, gap);
        if (endsBlock)
        {
          if constexpr (PK) put_small(dst + hn + gap, u32x4{ 0x00000080u, 0x00000100u, 0u, 0u }, 9u);     // 80 | 00 00 00 00 | 01 00 00 00
          // (plain: eleven zero bytes -- the image is zero already)
        }
      }
#endif
    }
    const uint32_t packetBytes = imagePos - 9u;
    const uint32_t finalRLE = carRLE;
    const uint32_t kLit = n - finalRLE;
    const uint32_t streamSize = 9u + packetBytes + (ended ? 0u : TERM + kLit);
    if (lane == 0u)
    {
      __hip_atomic_store(tiles + b, (b == 0u ? TILE_PREFIX : TILE_AGGREGATE) | (unsigned long long)streamSize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // stream header; literal terminator's fixed part
      put_small(0u, u32x4{ n, streamSize, 0u, 0u }, 9u);
      if (!ended)
      {
        if constexpr (PK) put_small(9u + packetBytes, u32x4{ 0x00000080u, (((kLit + 1u) << 1) | 1u) << 8, (((kLit + 1u) << 1) | 1u) >> 24, 0u }, 9u);
        else put_small(9u + packetBytes, u32x4{ 0u, (kLit + 1u) << 24, (kLit + 1u) >> 8, 0u }, 11u);
      }
    }
    wave_sync();
    {
      const uint32_t nj = jobCount;
      for (uint32_t j = 0; j < nj; j++) coop_copy(jobs[3u * j], jobs[3u * j + 1u], jobs[3u * j + 2u]);
      if (!ended && kLit != 0u) coop_copy(finalRLE, 9u + packetBytes + TERM, kLit);
    }
    wave_sync();

    // ---- where does the stream go?  decoupled look-back over the blocks in front ----
    unsigned long long excl = 0;
#ifdef HSRLE_W8_NOLOOKBACK   // timing experiment only (wrong offsets)
    excl = (unsigned long long)b * 4304ull;
    if (false)
#else
    if (b != 0u)
#endif
    {
      int64_t j = (int64_t)b - 1;
      for (;;)
      {
        const int64_t mine = j - (int64_t)lane;
        unsigned long long v = 0;
        if (mine >= 0) v = __hip_atomic_load(tiles + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long flag = (mine >= 0) ? (v >> 62) : 2ull;     // in front of block 0: an inclusive prefix of 0
        const unsigned long long missing = __ballot(flag == 0ull);
        const unsigned long long prefixes = __ballot(flag == 2ull);
        // usable lanes: those in front of the first missing one; stop at the first inclusive prefix among them
        const uint32_t firstMissing = missing ? (uint32_t)__builtin_ctzll(missing) : 64u;
        const uint32_t firstPrefix = prefixes ? (uint32_t)__builtin_ctzll(prefixes) : 64u;
        const uint32_t take = (firstPrefix < firstMissing) ? firstPrefix + 1u : firstMissing;
        unsigned long long part = (lane < take) ? (v & TILE_VALUE) : 0ull;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) part += __shfl_xor(part, dd, 64);
        excl += part;
        if (firstPrefix < firstMissing) break;
        j -= (int64_t)take;                                                // (take == 0: the nearest block is not there yet: look again)
        if (take == 0u) __builtin_amdgcn_s_sleep(2);
      }
      if (lane == 0u) __hip_atomic_store(tiles + b, TILE_PREFIX | (excl + (unsigned long long)streamSize), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0u)
    {
      offsets[b] = excl;
      if (b + 1u == nBlocks) offsets[nBlocks] = excl + streamSize;
    }

    // ---- the image leaves LDS once: destination-aligned 16-byte stores ----
    {
      uint8_t *const dst = payload + excl;
      const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
      const uint32_t h = head < streamSize ? head : streamSize;
      if (lane < h) dst[lane] = outbuf[lane];
      const uint32_t bodyBytes = (streamSize - h) & ~15u;
      for (uint32_t k2 = lane * 16u; k2 < bodyBytes; k2 += 64u * 16u)
        __builtin_nontemporal_store(lds_read16(outbuf, h + k2), (u32x4 *)(dst + h + k2));
      const uint32_t tail = streamSize - h - bodyBytes;
      if (lane < tail) dst[h + bodyBytes + lane] = outbuf[h + bodyBytes + lane];
    }
    wave_sync();
  }
}

} // namespace hsrle
