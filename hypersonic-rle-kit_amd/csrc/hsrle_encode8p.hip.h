// hsrle_encode8p.hip.h -- POSITION-PARALLEL encoder for the two list-free 8 bit multi-symbol codecs (rle8_multi, rle8_packed_multi), blocks of at
// most 4 KiB: one wave per block, every phase spread over the 64 lanes, the payload written ONCE to its final place (round 5).
//
// Replaces: src/rle8_extreme_cpu.h:936-1099 (canonical AVX2 body: cmpeq + movemask + ctz scan, emit rule :974-1001, packet writer :1000-1058),
//           :111-199 (scalar tail), :203-338 (final block / terminators) -- and, in this library, the lane-per-block ring encoder + staging slots +
//           k_compact for big containers of these two codecs (hsrle_encode8.hip.h: 5.1 ms + 1.84 ms per 8 GiB, 22 - 31 GB of traffic).
//
// Why another division of labour: the lane-per-block kernel is a chain of ~350 instructions per run end at 37 % lane use and writes 16-byte pieces
// into 64 different slots per instruction.  Priced in round 5 (experiments/r05/call1.sh): with header assembly and literal emission compiled out it
// falls from 5.4 to 2.0 ms -- the decisions are cheap, the per-lane byte emission is what costs.  Here nothing is emitted by a lane for "its" block:
//   1. a lane holds 64 consecutive input bytes; equality of neighbouring bytes = 64 bits per lane (SWAR + v_dot4 movemask), run starts / ends are
//      bit masks, runs that cross lanes are stitched with a carry bit and a DPP prefix maximum of start positions;
//   2. candidate runs (>= 3 bytes) are compacted into an LDS list by a DPP prefix sum of popcounts: candidate k is judged by lane k;
//   3. the emit decisions are a chain through (lastRLE, lastSymbol) -- but a run of >= 11 (Packed) / 6 (plain) bytes is stored whatever the state
//      (SURVEY.md A.4), so the state behind it is known at once and only the candidates between two such runs wait for their left neighbour:
//      a DPP shift of (known, lastRLE, lastSymbol) per pass, 1 - 3 passes on the synthetic buffers (the worst case, a block without a long
//      run, takes one pass per candidate: as slow as a scalar walk, still correct);
//   4. a DPP prefix sum over header + literal bytes gives every packet its place in the stream; the packet lanes write their header bytes into an
//      LDS image of the stream and one list entry (stream position | literal source | header length);
//   5. the stream leaves as whole 16-byte chunks, one lane per OUTPUT chunk (the shape of k_expand_packets): binary search in the packet list,
//      header bytes from the image, literal bytes by ONE unaligned global load per stretch placed so that no shift is needed (the block's
//      input was read by this wave a moment ago: L1 / L2), merged under byte masks, stored to the chunk's FINAL address.  The chunks at the two
//      ends of a stream are shared with the neighbouring blocks' streams and go out as byte / word pieces.
// MODE 3 (what ships): ONE launch.  A block's place in the payload is the sum of the sizes of the blocks in front of it; every wave publishes its
// size as soon as its decisions are made and looks back for that sum (decoupled look-back on two levels: the 64 blocks of its group, then one word
// per group -- one level is unstable here: 700 blocks per microsecond finish against a 1 - 2 us round trip of an agent-scope load), then copies
// its image out: the input is read once, the payload written once, nothing is staged (U + C of traffic).
// MODE 0 / 1 (kept for A/B runs and as the fallback of an experiment build): two launches around the size scan -- sizes only, then the same
// detection and decisions again + emission at the scanned offsets.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // merge_low_m, wave_sync

namespace hsrle {

#ifdef HSRLE_PP_STAMPS   // diagnostic build (never shipped): cycle sums per phase of k_encode8_pp<.., MODE 1>, added up over all waves
__device__ unsigned long long g_pp_stamps[16];
#define HS_PSTAMP(slot) { if constexpr (MODE == 1 || MODE == 3) { __builtin_amdgcn_sched_barrier(0); if (HSRLE_PP_STAMPS >= 2) __builtin_amdgcn_s_waitcnt(0); const unsigned long long tq_ = __builtin_amdgcn_s_memtime(); pst[slot] += tq_ - pt0; pt0 = tq_; __builtin_amdgcn_sched_barrier(0); } }
#else
#define HS_PSTAMP(slot)
#endif

// (kPpMaxBlock, kPpCtrlWords: hsrle_common.hip.h -- the host side needs them too)
constexpr uint32_t kPpGroup = 64u;                              // blocks per look-back group

// ---- wave-level primitives on DPP (no LDS round trip: a ds_bpermute based __shfl_up costs ~100 cycles of latency per step) ----
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false); }   // lane i <- lane i - 1
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false); }   // lane i <- lane i + 1
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)   // inclusive
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}
__device__ __forceinline__ int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t wave_scan_max(int32_t v)     // inclusive; values >= -1
{
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false));
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xF, 0xF, false));
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xF, 0xF, false));
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xF, 0xF, false));
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xA, 0xF, false));
  v = imax(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xC, 0xF, false));
  return v;
}
__device__ __forceinline__ uint32_t wave_lane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// LDS of one wave's block.  lst: candidate run ends (last byte position); img: the stream's image (EMIT); jobs: literal stretches the whole wave copies
constexpr uint32_t kPpCoopMin = 80u;                            // literal stretches longer than this are copied by the whole wave, shorter ones by their packet's lane
constexpr uint32_t kPpJobs = kPpMaxBlock / (kPpCoopMin + 4u) + 2u;
constexpr uint32_t kPpInPad = 16u;                              // the input image starts 16 bytes into its buffer (a literal window may begin up to 15 bytes in front of its stretch)
template <bool EMIT>
struct PpShared
{
  uint8_t img[EMIT ? (kPpMaxBlock + 193u + 15u + 16u + 15u) / 16u * 16u : 16u] __attribute__((aligned(16)));
  uint8_t inb[EMIT ? kPpInPad + kPpMaxBlock + 32u : 16u] __attribute__((aligned(16)));   // the block's input (EMIT): literals and run symbols come from here, not from L2 (0.8 us per gather under load)
  uint8_t mlut[EMIT ? 17u * 16u : 16u] __attribute__((aligned(16)));   // entry c: the low c bytes, c = 0 .. 16
  uint64_t starts[64];
  uint64_t jobs[EMIT ? kPpJobs : 1u];
  uint16_t lst[64];                                                     // the round's candidates (last byte positions), handed from the lanes that found them to the lanes that judge them
  uint16_t carryStart[64];
  uint32_t jobCount;
};

// what the single-pass launch needs beside the two-launch arguments: the look-back words (zeroed by the caller)
struct PpLookBack
{
  uint32_t *sizeW;               // [nBlocks]  FLAG | stream size of the block
  unsigned long long *grpAcc;    // [groups]   blocks of the group that have published << 48 | sum of their sizes
  unsigned long long *grpPref;   // [groups]   FLAG | payload bytes in front of the group
  uint32_t *status;              // [0] != 0: a look-back gave up (never seen; the container then carries a payload size of 0 and no decoder accepts it)
};
constexpr uint32_t kPpSizeFlag = 0x80000000u;
constexpr unsigned long long kPpPrefFlag = 1ull << 63;
constexpr uint32_t kPpSpinLimit = 1u << 22;                     // polls before a look-back gives up (seconds: a hang is a lost GPU box)

// lane l's 64 bytes of block b (zeros behind the end of the input)
__device__ __forceinline__ void pp_load(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t b, u32x4 (&x)[4])
{
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint8_t *const d = in + at;
  const uint32_t base = threadIdx.x * 64u;
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++)
  {
    const uint32_t pos = base + 16u * j;
    u32x4 v = u32x4{ 0, 0, 0, 0 };
    if (pos + 16u <= n) v = ld128(d + pos);
    else if (pos < n) v = load16_edge(d, (int64_t)pos, (uint64_t)n);
    x[j] = v;
  }
}

// Where does block b's stream go?  Decoupled look-back on two levels (all 64 lanes call it; the answer is wave uniform).  Level 1: the sizes of the
// blocks of b's group in front of b (one coalesced read of the group's size words; they were published within a fraction of a microsecond of each
// other).  Level 2: the payload bytes in front of the group -- found by the group's FIRST block, which reads 64 groups at a time backwards (sum and
// count of each, published prefixes) until it meets a published prefix, and published for the other 63, which only poll that one word.
// Blocks are taken in workgroup index order: a workgroup's predecessors have all been dispatched (each XCD hands out its share of the grid in
// order), so what a look-back waits for is always running or done; every wait is bounded all the same (PpLookBack::status).
__device__ __forceinline__ uint64_t pp_look_back(const PpLookBack &lb, uint32_t nBlocks, uint32_t b, uint32_t mySize, uint64_t *__restrict__ offsets, unsigned long long *diag = nullptr)
{
#ifdef HSRLE_PP_STAMPS
  const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t lane = threadIdx.x, grp = b / kPpGroup, gi = b % kPpGroup;
  auto fail = [&]() __attribute__((always_inline)) -> uint64_t {
    if (lane == 0u) { __hip_atomic_store(lb.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); offsets[nBlocks] = 0ull; }
    return ~0ull;
  };
  // first requests of both levels together
  uint32_t v = kPpSizeFlag;
  if (lane < gi) v = __hip_atomic_load(lb.sizeW + (uint64_t)grp * kPpGroup + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int64_t j = (int64_t)grp - 1;                      // level 2 window: lane l looks at group j - l
  unsigned long long pf = kPpPrefFlag, ac = (unsigned long long)kPpGroup << 48;     // (in front of group 0: a complete group of size 0 with prefix 0)
  if (grp != 0u)
  {
    if (gi == 0u)
    {
      if (j - (int64_t)lane >= 0) { pf = __hip_atomic_load(lb.grpPref + (j - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ac = __hip_atomic_load(lb.grpAcc + (j - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    else
      pf = __hip_atomic_load(lb.grpPref + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  uint32_t spins = 0;
  // level 1
  while (__ballot((v & kPpSizeFlag) == 0u) != 0ull)
  {
    if (++spins > kPpSpinLimit) return fail();
    __builtin_amdgcn_s_sleep(8);
    if (lane < gi) v = __hip_atomic_load(lb.sizeW + (uint64_t)grp * kPpGroup + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint32_t inGroup = wave_lane(wave_scan_add(lane < gi ? (v & ~kPpSizeFlag) : 0u), 63);
#ifdef HSRLE_PP_STAMPS
  const unsigned long long lt1 = __builtin_amdgcn_s_memtime();
  const uint32_t spins1 = spins;
#endif
  // level 2
  uint64_t pref = 0;
  if (grp != 0u)
  {
    if (gi == 0u)
    {
      for (;;)
      {
        const unsigned long long havePref = __ballot((pf & kPpPrefFlag) != 0ull), complete = __ballot((ac >> 48) == (unsigned long long)kPpGroup);
        const uint32_t firstPref = havePref ? (uint32_t)__builtin_ctzll(havePref) : 64u, firstOpen = ~complete ? (uint32_t)__builtin_ctzll(~complete) : 64u;
        bool again = false;                           // wave uniform
        if (firstPref < 64u && firstOpen > firstPref)
        {
          // the groups of lanes 0 .. firstPref are complete and the last of them knows what lies in front of it
          const uint32_t part = wave_lane(wave_scan_add(lane <= firstPref ? (uint32_t)ac : 0u), 63);
          const uint32_t plo = wave_lane((uint32_t)pf, (int)firstPref), phi = wave_lane((uint32_t)(pf >> 32), (int)firstPref);
          pref += (uint64_t)part + ((((uint64_t)phi << 32) | plo) & ~kPpPrefFlag);
          break;
        }
        else if (firstPref == 64u && firstOpen == 64u)
        {
          pref += (uint64_t)wave_lane(wave_scan_add((uint32_t)ac), 63);   // 64 complete groups without a published prefix: on to the 64 in front of them
          j -= 64;
        }
        else
        {
          if (++spins > kPpSpinLimit) return fail();
          __builtin_amdgcn_s_sleep(8);
          again = true;
        }
        (void)again;
        pf = kPpPrefFlag; ac = (unsigned long long)kPpGroup << 48;
        if (j - (int64_t)lane >= 0) { pf = __hip_atomic_load(lb.grpPref + (j - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ac = __hip_atomic_load(lb.grpAcc + (j - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      if (lane == 0u) __hip_atomic_store(lb.grpPref + grp, kPpPrefFlag | pref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
      while ((pf & kPpPrefFlag) == 0ull)
      {
        if (++spins > kPpSpinLimit) return fail();
        __builtin_amdgcn_s_sleep(8);
        pf = __hip_atomic_load(lb.grpPref + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      pref = pf & ~kPpPrefFlag;
    }
  }
#ifdef HSRLE_PP_STAMPS
  if (diag) { diag[14] = lt1 - lt0; diag[15] = __builtin_amdgcn_s_memtime() - lt1; diag[1] = ((unsigned long long)spins1 << 32) | (spins - spins1); }
#endif
  const uint64_t place = pref + inGroup;
  if (lane == 0u) { offsets[b] = place; if (b + 1u == nBlocks) offsets[nBlocks] = place + mySize; }
  return place;
}

// one block by one wave.  MODE 0: sizes[b]; MODE 1: the stream, written to payload + offsets[b]; MODE 3: the stream, its place found by look-back
// (offsets[b] is written, and offsets[nBlocks] by the last block)
template <int FAM, int MODE>
__device__ __forceinline__ void pp_block(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t b, uint32_t *__restrict__ sizes, uint64_t *__restrict__ offsets,
                                         uint8_t *__restrict__ payload, const PpLookBack &lb, PpShared<MODE != 0> &sh, const u32x4 (&x)[4])
{
  static_assert(FAM == PLAIN || FAM == PACKED, "the two list-free 8 bit multi-symbol codecs");
  constexpr bool PK = FAM == PACKED;
  constexpr uint32_t LONGC = PK ? 11u : 6u;          // runs this long are stored whatever the state
  constexpr uint32_t TERM = PK ? 9u : 11u;           // bytes of either terminator's fixed part
  const uint32_t lane = threadIdx.x;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint8_t *const d = in + at;
  const uint32_t base = lane * 64u;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };

#ifdef HSRLE_PP_STAMPS
  unsigned long long pst[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, pt0 = __builtin_amdgcn_s_memtime();
  pst[13] = pt0;
#endif
  // ---- 1. the lane's 64 input bytes (loaded by the caller: pp_load) -> 64 equality bits ----
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);
  }
  HS_PSTAMP(0)
  const uint32_t nextFirst = wave_shl1(x[0].x, 0u);
  HS_PSTAMP(1)
  uint64_t e64 = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++)
  {
    const u32x4 a = x[j];
    const uint32_t s = (j < 3u) ? x[j < 3u ? j + 1u : 3u].x : nextFirst;
    e64 |= (uint64_t)zero_mask16(a.x ^ alignbyte(a.y, a.x, 1), a.y ^ alignbyte(a.z, a.y, 1), a.z ^ alignbyte(a.w, a.z, 1), a.w ^ alignbyte(s, a.w, 1)) << (16u * j);
  }
  // position i matches only if its successor exists (bytes at or beyond n never match)
  const uint32_t validBits = (n > base + 1u) ? ((n - 1u - base) < 64u ? (n - 1u - base) : 64u) : 0u;
  e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
  const uint64_t carry = (uint64_t)wave_shr1((uint32_t)(e64 >> 63), 0u);
  const uint64_t prev = (e64 << 1) | carry;
  const uint64_t starts = e64 & ~prev;
  const uint64_t ends = ~e64 & prev;                                     // bit i: a run's last byte is base + i
  // start of the run that is open when this lane begins: the last start in the lanes in front
  const int32_t ownStart = (starts != 0ull) ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(starts)) : -1;
  const uint32_t carryStart = wave_shr1((uint32_t)wave_scan_max(ownStart), 0xFFFFFFFFu);
  // candidates: every run of at least 3 bytes (a run of exactly 2 ends one position behind its start)
  const uint64_t startPrev63 = (uint64_t)wave_shr1((uint32_t)(starts >> 63), 0u);
  const uint64_t cands = ends & ~((starts << 1) | startPrev63);
  const uint32_t cnt = (uint32_t)__builtin_popcountll(cands);
  const uint32_t inclCnt = wave_scan_add(cnt);
  const uint32_t R = wave_lane(inclCnt, 63);
  sh.starts[lane] = starts;
  sh.carryStart[lane] = (uint16_t)carryStart;
  uint64_t candLeft = cands;                          // this lane's candidates that no round has taken yet; the first of them is candidate number candAt of the block
  uint32_t candAt = inclCnt - cnt;
  // MODE != 0: stream byte s lives at img[s].  The image starts as zeros and
  // every header / literal piece is OR-ed into it (LDS atomics: pieces of different lanes share dwords), so no piece needs to know its neighbours.
  constexpr uint32_t aShift = 0u;                                        // (the image is laid out at STREAM alignment: the copy-out's 16-byte stores need no alignment on gfx950)
  if constexpr (MODE != 0)
  {
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
  }
  wave_sync();

  // literal bytes [src, src + len) of the block -> image bytes [ds, ds + len), the destination chunks t0, t0 + tStep, ... below tEnd of the stretch
  // by this lane: a 16-byte window of the input image placed so that every byte lands where it belongs (two aligned LDS reads + a byte funnel; from
  // L2 a gather took 0.8 us under load and the short-literal loop was half of a wave's life), cut to the stretch under two byte masks, OR-ed into the image
  auto put_chunks = [&](uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd) __attribute__((always_inline)) {
    const uint32_t de = ds + len, D0 = ds & ~15u;
    for (uint32_t t = t0; t < tEnd; t += tStep)
    {
      const uint32_t D = D0 + 16u * t;
      const u32x4 v = lds_read16(sh.inb, kPpInPad + src + D - ds);          // (D - ds may be "negative": the pad in front of the image absorbs it)
      const uint32_t lo = D < ds ? ds - D : 0u, hi = de - D < 16u ? de - D : 16u;      // chunk bytes [lo, hi)
      const u32x4 mh = lds_ld128(sh.mlut + (hi << 4)), ml = lds_ld128(sh.mlut + (lo << 4));
      unsigned long long *const ip = (unsigned long long *)(sh.img + D);
      const uint64_t w0 = (uint64_t)(v.x & mh.x & ~ml.x) | ((uint64_t)(v.y & mh.y & ~ml.y) << 32), w1 = (uint64_t)(v.z & mh.z & ~ml.z) | ((uint64_t)(v.w & mh.w & ~ml.w) << 32);
      atomicOr(ip, w0);
      atomicOr(ip + 1, w1);
    }
  };

 HS_PSTAMP(2)
  // ---- 2. one candidate per lane, 64 per round ----
  uint32_t carL = 0, carY = 0;                       // state in front of the round's first candidate
  uint32_t pos = 9u;                                 // stream position of the round's first packet
  bool ended = false;
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    // candidates r0 .. r0 + 63 go from the lanes that found them to the lanes that judge them (64 list entries whatever the block holds)
    while (candLeft != 0ull && candAt < r0 + 64u)
    {
      sh.lst[candAt - r0] = (uint16_t)(base + (uint32_t)__builtin_ctzll(candLeft));
      candAt++;
      candLeft &= candLeft - 1ull;
    }
    wave_sync();
    const bool have = r0 + lane < R;
    const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;
    wave_sync();
    const uint32_t iq = q >> 6, bit = q & 63u;
    const uint64_t st = sh.starts[iq];
    const uint32_t cs = (uint32_t)sh.carryStart[iq];
    const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
    const uint32_t p = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;
    const uint32_t e = q + 1u, count = e - p;
    uint32_t sym = 0u;
    if constexpr (MODE != 0) sym = (uint32_t)sh.inb[kPpInPad + (have ? p : 0u)]; else sym = have ? (uint32_t)d[p] : 0u;
    bool body = true;
    if constexpr (PK)
    {
      const int32_t kk = (int32_t)(count - 1u) / 32;                   // body / tail split of the canonical AVX2 encoder
      body = (e < n) && ((int32_t)p + 1 + 32 * kk < (int32_t)n - 32);
    }
    HS_PSTAMP(8)
    const bool sure = have && count >= LONGC;
    auto decide = [&](uint32_t inL, uint32_t inY, bool &same) __attribute__((always_inline)) -> int {
      const uint32_t rng = p - inL + 1u;
      same = false;
      if constexpr (PK)
      {
        if (!body) return (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
        same = sym == inY;
        const bool emit = count >= 11u || (rng <= 127u && ((same && count >= 3u) || count >= 4u));     // rle8_extreme_cpu.h:978
        return emit ? (rng <= 127u ? 1 : 2) : 0;
      }
      else
        return (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;                                               // :974
    };
    uint32_t outL = e, outY = sym;
    bool outKnown = sure || !have, inKnown = !have;
    uint32_t inL = 0, inY = 0;
    for (uint32_t pass = 0; pass < 66u; pass++)
    {
      const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), lr = wave_shr1(outL, carL), ls = wave_shr1(outY, carY);
      if (!inKnown && lk != 0u) { inKnown = true; inL = lr; inY = ls; }
      if (inKnown && !outKnown)
      {
        bool same;
        if (decide(inL, inY, same) == 0) { outL = inL; outY = inY; }
        else if (PK && !body) outY = inY;                                // (only the body rule tracks lastSymbol)
        outKnown = true;
      }
      if (__ballot(!inKnown) == 0ull) break;
    }
    HS_PSTAMP(9)
    bool same = false;
    const int k = have ? decide(inL, inY, same) : 0;
    const uint32_t gap = p - inL, rng = gap + 1u;
    // the packet header (rle8_extreme_cpu.h:1000-1058), little endian in hlo : hhi.  Nearly every packet has a one-byte count and a one-byte
    // range: 2 or 3 bytes (Packed) / 3 bytes (plain) assembled in one dword; the wide forms (at most 10 / 11 bytes) take the general path
    uint64_t hlo = 0; uint32_t hhi = 0, hl = 0;
    const uint32_t cfield = count - (PK ? 2u : 5u);
    const bool small = k == 1 && cfield <= (PK ? 127u : 255u);
    if (k)
    {
      if (small)
      {
        if constexpr (PK) { hlo = same ? (cfield | 0x80u | (rng << 9)) : (cfield | (sym << 8) | (rng << 17)); hl = same ? 2u : 3u; }
        else { hlo = sym | (cfield << 8) | (rng << 16); hl = 3u; }
      }
      else
      {
        uint32_t rv, rn;                                                   // range field: value, bytes
        if constexpr (PK)
        {
          const uint32_t sm = same ? 0x80u : 0u;
          const bool wide = cfield > 127u;
          hlo = wide ? ((uint64_t)cfield << 8) | sm : (uint64_t)(cfield | sm);
          hl = wide ? 5u : 1u;
          if (!same) { hlo |= (uint64_t)sym << (8u * hl); hl++; }
          rv = (k == 1) ? (rng << 1) & 0xFFu : ((rng << 1) | 1u);
          rn = (k == 1) ? 1u : 4u;
        }
        else
        {
          const bool wide = cfield > 255u;
          hlo = wide ? (uint64_t)sym | ((uint64_t)cfield << 16) : (uint64_t)(sym | (cfield << 8));
          hl = wide ? 6u : 2u;
          if (k != 1) hl++;                                                // (a zero byte in front of the 32 bit range)
          rv = rng;
          rn = (k == 1) ? 1u : 4u;
        }
        hlo |= (uint64_t)rv << (8u * hl);
        hhi = (hl > 4u) ? (uint32_t)(((uint64_t)rv << 32) >> (96u - 8u * hl)) : 0u;     // the part of the range field beyond byte 8 (hl <= 7)
        hl += rn;
      }
    }
    const uint32_t myBytes = k ? hl + gap : 0u;
    const uint32_t incl = wave_scan_add(myBytes);
    const uint32_t tot = wave_lane(incl, 63);
    if constexpr (MODE != 0)
    {
      const uint32_t at0 = pos + incl - myBytes;                         // the packet's place in the stream
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        // header: shifted to its byte phase, OR-ed into the image (two dwords for the short forms, at most four for the wide ones)
        const uint32_t ib = aShift + at0;
        uint32_t *const wp = (uint32_t *)(sh.img + (ib & ~3u));
        if (small)
        {
          const uint64_t hv = (uint64_t)(uint32_t)hlo << (8u * (ib & 3u));
          atomicOr(wp, (uint32_t)hv);
          if ((uint32_t)(hv >> 32) != 0u) atomicOr(wp + 1, (uint32_t)(hv >> 32));
        }
        else
        {
          const uint32_t sft = 32u - 8u * (ib & 3u);
          const uint32_t d0 = (uint32_t)hlo, d1 = (uint32_t)(hlo >> 32);
          atomicOr(wp, (uint32_t)(((uint64_t)d0 << 32) >> sft));
          atomicOr(wp + 1, (uint32_t)((((uint64_t)d1 << 32) | d0) >> sft));
          atomicOr(wp + 2, (uint32_t)((((uint64_t)hhi << 32) | d1) >> sft));
          if (hl > 7u) atomicOr(wp + 3, (uint32_t)((uint64_t)hhi >> sft));
        }
        // literals: short stretches by this lane, long ones noted for the whole wave
        ds = ib + hl;
        if (gap > kPpCoopMin) { const uint32_t slot = atomicAdd(&sh.jobCount, 1u); sh.jobs[slot] = (uint64_t)inL | ((uint64_t)ds << 13) | ((uint64_t)gap << 26); }
        else if (gap != 0u) nch = ((ds + gap - 1u) >> 4) - (ds >> 4) + 1u;
      }
      for (uint32_t t = 0; __ballot(t < nch) != 0ull; t += 2u)            // (the lanes together, two chunks per trip in flight)
      {
        if (t < nch) put_chunks(inL, ds, gap, t, 1u, t + 1u);
        if (t + 1u < nch) put_chunks(inL, ds, gap, t + 1u, 1u, t + 2u);
      }
    }
    HS_PSTAMP(11)
    const int lastRun = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);
    carL = (uint32_t)__builtin_amdgcn_readlane((int)outL, lastRun);     // (lastRun is wave uniform)
    carY = (uint32_t)__builtin_amdgcn_readlane((int)outY, lastRun);
    pos += tot;
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  HS_PSTAMP(3)
  // ---- 3. terminator, stream size ----
  const uint32_t kLit = ended ? 0u : n - carL;
  const uint32_t streamSize = pos + TERM + kLit;
  if constexpr (MODE == 0)
  {
    if (lane == 0u) sizes[b] = streamSize;
    return;
  }
  else
  {
    [[maybe_unused]] const uint32_t grp = b / kPpGroup, gi = b % kPpGroup;
    if constexpr (MODE == 3)
    {
      // publish: the block's size for its group's later blocks, and into the group's sum (the emission below runs while the words travel)
      if (lane == 0u)
      {
        __hip_atomic_store(lb.sizeW + b, kPpSizeFlag | streamSize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(lb.grpAcc + grp, (1ull << 48) + (unsigned long long)streamSize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane < 8u)
    {
      // stream header (rle8_extreme_cpu.c:5-15): u32 uncompressed, u32 compressed, u8 mode = 0
      const uint64_t h = (uint64_t)n | ((uint64_t)streamSize << 32);
      sh.img[aShift + lane] = (uint8_t)(h >> (8u * lane));
    }
    if (lane >= 16u && lane < 16u + TERM)
    {
      // end terminator / literal terminator (:203-338)
      const uint32_t t = lane - 16u;
      uint32_t v = 0;
      if constexpr (PK)
      {
        const uint32_t w = ended ? 1u : (((kLit + 1u) << 1) | 1u);       // 80 | 00 00 00 00 | u32
        v = (t == 0u) ? 0x80u : (t >= 5u ? (w >> (8u * (t - 5u))) & 0xFFu : 0u);
      }
      else
      {
        const uint32_t w = ended ? 0u : kLit + 1u;                       // 00 00 | 00 00 00 00 | 00 | u32
        v = (t >= 7u) ? (w >> (8u * (t - 7u))) & 0xFFu : 0u;
      }
      if (v != 0u) sh.img[aShift + pos + t] = (uint8_t)v;
    }
    wave_sync();
    // the long stretches and the literals behind the last stored run: every lane a chunk
    {
      const uint32_t nj = sh.jobCount;
      for (uint32_t j = 0; j <= nj; j++)
      {
        uint32_t src, ds, len;
        if (j < nj) { const uint64_t jb = sh.jobs[j]; src = (uint32_t)jb & 0x1FFFu; ds = (uint32_t)(jb >> 13) & 0x1FFFu; len = (uint32_t)(jb >> 26); }
        else { src = carL; ds = aShift + pos + TERM; len = kLit; }
        if (len != 0u) put_chunks(src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
      }
    }
    wave_sync();

    HS_PSTAMP(4)
    // ---- 4. the image leaves LDS once: whole 16-byte chunks at their final (unaligned) addresses, the last bytes one per lane ----
    {
      uint64_t place = 0;
      #ifdef HSRLE_PP_STAMPS
      if constexpr (MODE == 3) place = pp_look_back(lb, nBlocks, b, streamSize, offsets, pst); else place = offsets[b];
#else
      if constexpr (MODE == 3) place = pp_look_back(lb, nBlocks, b, streamSize, offsets); else place = offsets[b];
#endif
      if (place == ~0ull) return;                                          // (a look-back that gave up: see PpLookBack::status)
      uint8_t *const dst = payload + place;
      const uint32_t nFull = streamSize >> 4, tail = streamSize & 15u;
      for (uint32_t c = lane; c < nFull; c += 64u)
#if defined(HSRLE_PP_ABLATE) && HSRLE_PP_ABLATE == 1    // timing only: no payload store at all
        if (U == 0x7FFFFFFFFFFFFFF1ull)
#endif
        st128(dst + 16u * c, lds_ld128(sh.img + 16u * c));
      if (lane < tail) dst[16u * nFull + lane] = sh.img[16u * nFull + lane];
    }
    HS_PSTAMP(5)
#ifdef HSRLE_PP_STAMPS   // (a slot per block in the unused slot area: atomics on ONE address from 2 M waves distorted every memory wait)
    if constexpr (MODE == 1 || MODE == 3) { if (lane < 16u) { pst[6] = 1ull; pst[7] = R; pst[12] = pt0; unsigned long long v_ = 0; for (int q_ = 0; q_ < 16; q_++) if ((int)lane == q_) v_ = pst[q_]; ((unsigned long long *)((uint8_t *)lb.status + (64u << 20)))[16ull * b + lane] = v_; } }
#endif
  }
}

// One block per workgroup.  MODE 0 / 1: XCD-aware order (neighbouring streams share payload lines: they meet in one L2); MODE 3: index order (see pp_look_back).
template <int FAM, int MODE>
__global__ __launch_bounds__(64) void k_encode8_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                   uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpLookBack lb)
{
  __shared__ PpShared<MODE != 0> sh;
  if constexpr (MODE != 0)
  {
    if (threadIdx.x < 17u)
    {
      // merge masks: entry c = the low c bytes (as in k_decode_blocks; entry 16: all of them)
      const uint32_t c = threadIdx.x;
      const uint64_t part = ~(~0ull << (8u * (c & 7u)));
      const bool hiHalf = c >= 8u;
      const uint32_t p0 = (c == 16u) ? ~0u : (uint32_t)part, p1 = (c == 16u) ? ~0u : (uint32_t)(part >> 32);
      lds_st128(sh.mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
    }
    wave_sync();
  }
  // MODE 3: workgroup i runs on XCD i % 8 (hsrle_common.hip.h: xcd_tile); the 64 blocks of a look-back group go to ONE XCD, as 64 consecutive
  // workgroups of its share of the grid -- they start within a fraction of a microsecond of each other, and group g runs on XCD g % 8.  (With
  // block = workgroup index a group is spread over all XCDs, which drift apart by their whole resident capacity: every wave then waited ~6 polls
  // for its group's earlier blocks, and 3 500 waiting waves polling 512 bytes each were 1.9 TB/s of traffic -- 11.5 ms per 8 GiB.)
  uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
  if constexpr (MODE == 3)
  {
    const uint32_t xcd = blockIdx.x % 8u, li = blockIdx.x / 8u;
    b = ((li / kPpGroup) * 8u + xcd) * kPpGroup + li % kPpGroup;
  }
  if (b < nBlocks)
  {
    u32x4 x[4];
    pp_load(in, U, B, b, x);
    pp_block<FAM, MODE>(in, U, B, nBlocks, b, sizes, offsets, payload, lb, sh, x);
  }
}

} // namespace hsrle
