// hsrle_encode8p.hip.h -- POSITION-PARALLEL encoder for the two list-free 8 bit multi-symbol codecs (rle8_multi, rle8_packed_multi), blocks of at
// most 4 KiB: one wave per block, every phase spread over the 64 lanes, the payload written ONCE to its final place (round 5).
//
// Replaces: src/rle8_extreme_cpu.h:936-1099 (canonical AVX2 body: cmpeq + movemask + ctz scan, emit rule :974-1001, packet writer :1000-1058),
//           :111-199 (scalar tail), :203-338 (final block / terminators) -- and, in this library, the lane-per-block ring encoder + staging slots +
//           k_compact for containers of these two codecs (hsrle_encode8.hip.h: 5.1 ms + 1.84 ms per 8 GiB, 22 - 31 GB of traffic).
//
// Why another division of labour: the lane-per-block kernel is a chain of ~350 instructions per run end at 37 % lane use and writes 16-byte pieces
// into 64 different slots per instruction.  Priced in round 5 (LAB_NOTEBOOK.md): with header assembly and literal emission compiled out it
// falls from 5.4 to 2.0 ms -- the decisions are cheap, the per-lane byte emission is what costs.  Here nothing is emitted by a lane for "its" block:
//   1. a lane holds 64 consecutive input bytes; equality of neighbouring bytes = 64 bits per lane (SWAR + v_dot4 movemask), run starts / ends are
//      bit masks, runs that cross lanes are stitched with a carry bit and a DPP prefix maximum of start positions;
//   2. candidate runs (>= 3 bytes) are numbered by a DPP prefix sum of popcounts and handed, 64 per round, to the lanes that judge them;
//   3. the emit decisions are a chain through (lastRLE, lastSymbol) -- but a run of >= 11 (Packed) / 6 (plain) bytes is stored whatever the state
//      (SURVEY.md A.4), so the state behind it is known at once and only the candidates between two such runs wait for their left neighbour:
//      a DPP shift of (known, lastRLE, lastSymbol) per pass, 1 - 3 passes on the synthetic buffers (the worst case, a block without a long
//      run, takes one pass per candidate: as slow as a scalar walk, still correct);
//   4. a DPP prefix sum over header + literal bytes gives every packet its place in the stream;
//   5. the packet lanes OR their header bytes and their literal bytes (16-byte windows of an LDS copy of the input, cut to the stretch under two
//      byte masks) into a zeroed LDS image of the stream; literal stretches of more than kPpCoopMin bytes are copied by the whole wave;
//   6. the image leaves LDS once, as 16-byte stores to the stream's final (unaligned) address.
// Two launches around the size scan.  MODE 0: steps 1 - 4, leaves sizes[b] and ONE RECORD PER STORED RUN (start, end, same-symbol flag, range form).
// MODE 1: the block's input + its records -> steps 4 - 6 (a block whose records did not fit repeats steps 1 - 3).  Both kernels are bound by
// vector-instruction issue (PMC: ~75 % VALU busy at 4 cycles per wave instruction), not by memory: what is saved is instructions.
// A single-launch form (sizes published, places by a two-level decoupled look-back) is bit-exact and 2.3 x slower -- an agent-scope poll costs ~3 us
// under load against a wave's life of 6: experiments/r05/encode8p_single_pass_lookback.hip.h.txt, LAB_NOTEBOOK.md.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // lds_read16, wave_sync

namespace hsrle {

#ifdef HSRLE_PP_STAMPS   // diagnostic build (never shipped): cycles per phase of k_encode8_pp<.., 1>, one slot per block behind the scratch words (tools/probe_pp_stamps.py)
#define HS_PSTAMP(slot) { if constexpr (MODE == 1) { __builtin_amdgcn_sched_barrier(0); if (HSRLE_PP_STAMPS >= 2) __builtin_amdgcn_s_waitcnt(0); const unsigned long long tq_ = __builtin_amdgcn_s_memtime(); pst[slot] += tq_ - pt0; pt0 = tq_; __builtin_amdgcn_sched_barrier(0); } }
#else
#define HS_PSTAMP(slot)
#endif

// (kPpMaxBlock, kPpRecords: hsrle_common.hip.h -- the host side needs them too)
constexpr uint32_t kPpNoRecords = 0xFFFFFFFFu;                  // recCount[b]: the block stored more runs than its records hold

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

#ifndef HSRLE_PP8_INPUT_LDS
#define HSRLE_PP8_INPUT_LDS 1   // (round 6: 8 GiB rle8_packed_multi encode 1 727 -> 1 841 GiB/s; 0 = the global gather, A/B builds)
#endif
constexpr bool kPp8InputLds = HSRLE_PP8_INPUT_LDS != 0;        // MODE 0: a candidate's symbol from an LDS copy of the block instead of a global gather (A/B builds)
#ifndef HSRLE_PP8_BPW
#define HSRLE_PP8_BPW 1
#endif
constexpr uint32_t kPp8Bpw = HSRLE_PP8_BPW;                      // blocks per wave of k_encode8_pp, the next block's input requested ahead (A/B builds; round 6, 8 GiB: 2 -> 1 745, 4 -> 1 775 GiB/s against 1 836 with one block per wave)
constexpr uint32_t kPpCoopMin = 80u;                            // literal stretches longer than this are copied by the whole wave, shorter ones by their packet's lane
constexpr uint32_t kPpJobs = kPpMaxBlock / (kPpCoopMin + 4u) + 2u;
constexpr uint32_t kPpInPad = 16u;                              // the input image starts 16 bytes into its buffer (a literal window may begin up to 15 bytes in front of its stretch)
// LDS of one wave's block
template <bool EMIT, bool INPUT = EMIT, bool MTF = false>
struct PpShared
{
  uint8_t img[EMIT ? (kPpMaxBlock + 193u + 15u + 16u + 15u) / 16u * 16u : 16u] __attribute__((aligned(16)));   // the stream under construction
  uint8_t inb[INPUT ? kPpInPad + kPpMaxBlock + 32u : 16u] __attribute__((aligned(16)));   // the block's input: literals and run symbols come from here, not from L2 (0.8 us per gather under load)
  uint8_t mlut[EMIT ? 17u * 16u : 16u] __attribute__((aligned(16)));   // entry c: the low c bytes, c = 0 .. 16
  uint64_t starts[64];                                                  // run-start bits of every lane's 64 positions
  uint64_t mtfScratch[MTF ? 152u : 1u];                                 // LUT codecs: the round's run symbols and list heads
  uint64_t jobs[EMIT ? kPpJobs : 1u];                                   // literal stretches for the whole wave
  uint16_t lst[64];                                                     // the round's candidates (last byte positions), handed from the lanes that found them to the lanes that judge them
  uint16_t carryStart[64];                                              // start of the run that is open where a lane's positions begin
  uint32_t jobCount;
};

// scratch between the two launches
struct PpScratch
{
  uint32_t *recs;       // [nBlocks][recStride]   start | (end - 1) << 12 | same << 24 | (32 bit range field) << 25, one per stored run
  uint32_t *recCount;   // [nBlocks]              stored runs, kPpNoRecords where they did not fit
  uint32_t recStride;   // pp_record_stride(B)
  uint8_t *stamps;      // diagnostic builds
};

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

// one block by one wave.  MODE 0: sizes[b] and the block's records; MODE 1: the stream, written to payload + offsets[b].  rec0: the block's first 64
// records (MODE 1; requested together with the input)
template <int FAM, int MODE>
__device__ __forceinline__ void pp_block(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t b, uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offsets,
                                         uint8_t *__restrict__ payload, const PpScratch &sc, PpShared<MODE != 0, kPp8InputLds || MODE != 0> &sh, const u32x4 (&x)[4], uint32_t rec0)
{
  static_assert(FAM == PLAIN || FAM == PACKED, "the two list-free 8 bit multi-symbol codecs");
  constexpr bool PK = FAM == PACKED;
  constexpr uint32_t LONGC = PK ? 11u : 6u;          // runs this long are stored whatever the state
  constexpr uint32_t TERM = PK ? 9u : 11u;           // bytes of either terminator's fixed part
  const uint32_t lane = threadIdx.x;
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  [[maybe_unused]] const uint8_t *const d = in + at;
  const uint32_t base = lane * 64u;
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  const uint32_t *const myRecs = sc.recs + (uint64_t)b * sc.recStride;
  // (the stream's place: asked for at the start, needed at the very end -- not a dependent load in front of the copy-out)
  [[maybe_unused]] uint64_t myOffset = 0;
  if constexpr (MODE == 1) myOffset = offsets[b];
  // (and the block's second 64 records, where its record area holds that many: a block of more than 64 stored runs -- the rule on video-shaped data -- does not wait for them in its second round)
  [[maybe_unused]] uint32_t rec1 = 0;                                    // (the third and fourth 64 too: measured no better, four loads for every block)
  if constexpr (MODE == 1) { if (sc.recStride >= 128u) rec1 = myRecs[64u + threadIdx.x]; }

#ifdef HSRLE_PP_STAMPS
  unsigned long long pst[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, pt0 = __builtin_amdgcn_s_memtime();
  pst[13] = pt0;
#endif
  // MODE 1 with records: nothing has to be found or decided again
  uint32_t recN = kPpNoRecords;
  if constexpr (MODE == 1) recN = sc.recCount[b];
  const bool fromRecs = MODE == 1 && recN != kPpNoRecords;

  if constexpr (MODE != 0)
  {
    // the input image, and the zeroed stream image (stream byte s lives at img[s]: every header / literal piece is OR-ed in -- LDS atomics, pieces of
    // different lanes share dwords -- so no piece needs to know its neighbours; the copy-out's 16-byte stores need no alignment on gfx950)
#pragma unroll
    for (uint32_t c = 0; c < (sizeof(sh.img) / 16u + 63u) / 64u; c++)
      if (lane + 64u * c < sizeof(sh.img) / 16u) lds_st128(sh.img + 16u * (lane + 64u * c), zero4);
    if (lane == 0u) sh.jobCount = 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);   // (the first use of the input: the zeroing above ran while it was on its way)
  }
  if constexpr (MODE == 0 && kPp8InputLds)
  {
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) lds_st128(sh.inb + kPpInPad + base + 16u * j, x[j]);
  }
  HS_PSTAMP(0)

  // ---- 1. 64 equality bits per lane, run starts / ends, candidates ----
  uint32_t R = recN;                                  // rounds run over R candidates (or R records)
  uint64_t candLeft = 0;                              // this lane's candidates that no round has taken yet; the first of them is candidate number candAt of the block
  uint32_t candAt = 0;
  if (!fromRecs)
  {
    const uint32_t nextFirst = wave_shl1(x[0].x, 0u);
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
    R = wave_lane(inclCnt, 63);
    sh.starts[lane] = starts;
    sh.carryStart[lane] = (uint16_t)carryStart;
    candLeft = cands;
    candAt = inclCnt - cnt;
  }
  wave_sync();

  // literal bytes [src, src + len) of the block -> image bytes [ds, ds + len), the destination chunks t0, t0 + tStep, ... below tEnd of the stretch
  // by this lane: a 16-byte window of the input image placed so that every byte lands where it belongs (two aligned LDS reads + a byte funnel),
  // cut to the stretch under two byte masks, OR-ed into the image
  [[maybe_unused]] auto put_chunks = [&](uint32_t src, uint32_t ds, uint32_t len, uint32_t t0, uint32_t tStep, uint32_t tEnd) __attribute__((always_inline)) {
    const uint32_t de = ds + len, D0 = ds & ~15u;
    for (uint32_t t = t0; t < tEnd; t += tStep)
    {
      const uint32_t D = D0 + 16u * t;
      // (D - ds may be "negative": the pad in front of the image absorbs it.)  Five dwords from the dword below the window + four v_alignbyte: dword
      // reads need no 16-byte alignment, and the 16-byte form (two aligned reads + a select funnel) was 15 vector instructions of the chunk's 35
      const uint32_t wa = kPpInPad + src + D - ds;
      const uint32_t *const wq = (const uint32_t *)(sh.inb + (wa & ~3u));
      const uint32_t q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3], q4 = wq[4], sb = wa & 3u;
      const u32x4 v = u32x4{ alignbyte(q1, q0, sb), alignbyte(q2, q1, sb), alignbyte(q3, q2, sb), alignbyte(q4, q3, sb) };
      const uint32_t lo = D < ds ? ds - D : 0u, hi = de - D < 16u ? de - D : 16u;      // chunk bytes [lo, hi)
      const u32x4 mh = lds_ld128(sh.mlut + (hi << 4)), ml = lds_ld128(sh.mlut + (lo << 4));
      unsigned long long *const ip = (unsigned long long *)(sh.img + D);
      const uint64_t w0 = (uint64_t)(v.x & mh.x & ~ml.x) | ((uint64_t)(v.y & mh.y & ~ml.y) << 32), w1 = (uint64_t)(v.z & mh.z & ~ml.z) | ((uint64_t)(v.w & mh.w & ~ml.w) << 32);
      atomicOr(ip, w0);
      atomicOr(ip + 1, w1);
    }
  };

  HS_PSTAMP(2)
  // ---- 2. one candidate (or record) per lane, 64 per round ----
  uint32_t carL = 0, carY = 0;                       // state in front of the round's first candidate
  uint32_t pos = 9u;                                 // stream position of the round's first packet
  uint32_t K = 0;                                    // stored runs so far
  bool ended = false;
  for (uint32_t r0 = 0; r0 < R; r0 += 64u)
  {
    const bool have = r0 + lane < R;
    const int lastLane = (int)((R - r0 < 64u) ? R - r0 - 1u : 63u);      // (wave uniform)
    uint32_t p = 0, e = 0, sym = 0, inL = 0, outL = 0;
    bool same = false;
    int k = 0;
    if (fromRecs)
    {
      const uint32_t rec = (r0 == 0u) ? rec0 : ((r0 == 64u && sc.recStride >= 128u) ? rec1 : (have ? myRecs[r0 + lane] : 0u));
      p = rec & 0xFFFu; e = ((rec >> 12) & 0xFFFu) + 1u;
      same = ((rec >> 24) & 1u) != 0u;
      k = have ? 1 + (int)((rec >> 25) & 1u) : 0;
      outL = e;                                                          // (every record is a stored run: lastRLE behind it is its end)
      inL = wave_shr1(outL, carL);
      if constexpr (MODE != 0) sym = (uint32_t)sh.inb[kPpInPad + (have ? p : 0u)];
    }
    else
    {
      // candidates r0 .. r0 + 63 go from the lanes that found them to the lanes that judge them (64 list entries whatever the block holds)
      while (candLeft != 0ull && candAt < r0 + 64u)
      {
        sh.lst[candAt - r0] = (uint16_t)(base + (uint32_t)__builtin_ctzll(candLeft));
        candAt++;
        candLeft &= candLeft - 1ull;
      }
      wave_sync();
      const uint32_t q = have ? (uint32_t)sh.lst[lane] : 0u;
      wave_sync();
      const uint32_t iq = q >> 6, bit = q & 63u;
      const uint64_t st = sh.starts[iq];
      const uint32_t cs = (uint32_t)sh.carryStart[iq];
      const uint64_t sBelow = st & ((bit >= 63u) ? ~0ull : ((2ull << bit) - 1ull));
      p = (sBelow != 0ull) ? (iq << 6) + 63u - (uint32_t)__builtin_clzll(sBelow) : cs;
      e = q + 1u;
      const uint32_t count = e - p;
      if constexpr (MODE != 0 || kPp8InputLds) sym = (uint32_t)sh.inb[kPpInPad + (have ? p : 0u)]; else sym = have ? (uint32_t)d[p] : 0u;
      bool body = true;
      if constexpr (PK)
      {
        const int32_t kk = (int32_t)(count - 1u) / 32;                   // body / tail split of the canonical AVX2 encoder
        body = (e < n) && ((int32_t)p + 1 + 32 * kk < (int32_t)n - 32);
      }
      HS_PSTAMP(8)
      const bool sure = have && count >= LONGC;
      auto decide = [&](uint32_t iL, uint32_t iY, bool &sm) __attribute__((always_inline)) -> int {
        const uint32_t rng = p - iL + 1u;
        sm = false;
        if constexpr (PK)
        {
          if (!body) return (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
          sm = sym == iY;
          const bool emit = count >= 11u || (rng <= 127u && ((sm && count >= 3u) || count >= 4u));     // rle8_extreme_cpu.h:978
          return emit ? (rng <= 127u ? 1 : 2) : 0;
        }
        else
          return (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;                                               // :974
      };
      uint32_t outY = sym, inY = 0;
      outL = e;
      bool outKnown = sure || !have, inKnown = !have;
      for (uint32_t pass = 0; pass < 66u; pass++)
      {
        const uint32_t lk = wave_shr1(outKnown ? 1u : 0u, 1u), lr = wave_shr1(outL, carL), ls = wave_shr1(outY, carY);
        if (!inKnown && lk != 0u) { inKnown = true; inL = lr; inY = ls; }
        if (inKnown && !outKnown)
        {
          bool sm;
          if (decide(inL, inY, sm) == 0) { outL = inL; outY = inY; }
          else if (PK && !body) outY = inY;                                // (only the body rule tracks lastSymbol)
          outKnown = true;
        }
        if (__ballot(!inKnown) == 0ull) break;
      }
      HS_PSTAMP(9)
      k = have ? decide(inL, inY, same) : 0;
      carY = wave_lane(outY, lastLane);
    }
    const uint32_t count = e - p, gap = p - inL, rng = gap + 1u;

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
    const uint32_t incl = wave_scan_add(myBytes | (k ? 0x10000u : 0u));    // bytes below bit 16 (< 4 096 + 64 x 11), stored runs above
    const uint32_t tot = wave_lane(incl, 63);
    if constexpr (MODE == 0)
    {
      // the record of a stored run: what MODE 1 needs to write its packet without finding or judging anything again
      const uint32_t idx = K + (incl >> 16) - 1u;
      if (k && idx < sc.recStride) sc.recs[(uint64_t)b * sc.recStride + idx] = p | ((e - 1u) << 12) | (same ? 1u << 24 : 0u) | (k == 2 ? 1u << 25 : 0u);
    }
    else
    {
      const uint32_t at0 = pos + (incl & 0xFFFFu) - myBytes;             // the packet's place in the stream
      uint32_t nch = 0, ds = 0;
      if (k)
      {
        // header: shifted to its byte phase, OR-ed into the image (two dwords for the short forms, at most four for the wide ones)
        uint32_t *const wp = (uint32_t *)(sh.img + (at0 & ~3u));
        if (small)
        {
          const uint64_t hv = (uint64_t)(uint32_t)hlo << (8u * (at0 & 3u));
          atomicOr(wp, (uint32_t)hv);
          if ((uint32_t)(hv >> 32) != 0u) atomicOr(wp + 1, (uint32_t)(hv >> 32));
        }
        else
        {
          const uint32_t sft = 32u - 8u * (at0 & 3u);
          const uint32_t d0 = (uint32_t)hlo, d1 = (uint32_t)(hlo >> 32);
          atomicOr(wp, (uint32_t)(((uint64_t)d0 << 32) >> sft));
          atomicOr(wp + 1, (uint32_t)((((uint64_t)d1 << 32) | d0) >> sft));
          atomicOr(wp + 2, (uint32_t)((((uint64_t)hhi << 32) | d1) >> sft));
          if (hl > 7u) atomicOr(wp + 3, (uint32_t)((uint64_t)hhi >> sft));
        }
        // literals: short stretches by this lane, long ones noted for the whole wave
        ds = at0 + hl;
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
    carL = wave_lane(outL, lastLane);
    pos += tot & 0xFFFFu;
    K += tot >> 16;
    if (__ballot(k != 0 && e >= n) != 0ull) ended = true;
  }

  HS_PSTAMP(3)
  // ---- 3. terminator, stream size ----
  const uint32_t kLit = ended ? 0u : n - carL;
  const uint32_t streamSize = pos + TERM + kLit;
  if constexpr (MODE == 0)
  {
    if (lane == 0u) { sizes[b] = streamSize; sc.recCount[b] = (K <= sc.recStride) ? K : kPpNoRecords; }
    return;
  }
  else
  {
    if (lane < 8u)
    {
      // stream header (rle8_extreme_cpu.c:5-15): u32 uncompressed, u32 compressed, u8 mode = 0
      const uint64_t h = (uint64_t)n | ((uint64_t)streamSize << 32);
      sh.img[lane] = (uint8_t)(h >> (8u * lane));
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
      if (v != 0u) sh.img[pos + t] = (uint8_t)v;
    }
    wave_sync();
    // the long stretches and the literals behind the last stored run: every lane a chunk
    {
      const uint32_t nj = sh.jobCount;
      for (uint32_t j = 0; j <= nj; j++)
      {
        uint32_t src, ds, len;
        if (j < nj) { const uint64_t jb = sh.jobs[j]; src = (uint32_t)jb & 0x1FFFu; ds = (uint32_t)(jb >> 13) & 0x1FFFu; len = (uint32_t)(jb >> 26); }
        else { src = carL; ds = pos + TERM; len = kLit; }
        if (len != 0u) put_chunks(src, ds, len, lane, 64u, ((ds + len - 1u) >> 4) - (ds >> 4) + 1u);
      }
    }
    wave_sync();

    HS_PSTAMP(4)
    // ---- 4. the image leaves LDS once: whole 16-byte chunks at their final (unaligned) addresses, the last bytes one per lane ----
    {
      uint8_t *const dst = payload + myOffset;
      const uint32_t nFull = streamSize >> 4, tail = streamSize & 15u;
      for (uint32_t c = lane; c < nFull; c += 64u)
        st128(dst + 16u * c, lds_ld128(sh.img + 16u * c));
      if (lane < tail) dst[16u * nFull + lane] = sh.img[16u * nFull + lane];
    }
    HS_PSTAMP(5)
#ifdef HSRLE_PP_STAMPS   // (a slot per block: atomics on ONE address from 2 M waves distorted every memory wait)
    if constexpr (MODE == 1) { if (lane < 16u) { pst[6] = 1ull; pst[7] = R; pst[12] = pt0; unsigned long long v_ = 0; for (int q_ = 0; q_ < 16; q_++) if ((int)lane == q_) v_ = pst[q_]; ((unsigned long long *)sc.stamps)[16ull * b + lane] = v_; } }
#endif
  }
}

// One block per workgroup, XCD-aware order (neighbouring streams share payload lines: they meet in one L2).
template <int FAM, int MODE>
__global__ __launch_bounds__(64) void k_encode8_pp(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sizes,
                                                   const uint64_t *__restrict__ offsets, uint8_t *__restrict__ payload, PpScratch sc)
{
  __shared__ PpShared<MODE != 0, kPp8InputLds || MODE != 0> sh;
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
  if constexpr (kPp8Bpw == 1u)
  {
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
    if (b < nBlocks)
    {
      u32x4 x[4];
      pp_load(in, U, B, b, x);
      uint32_t rec0 = 0;
      if constexpr (MODE == 1) rec0 = sc.recs[(uint64_t)b * sc.recStride + threadIdx.x];   // (the first 64 records -- or garbage in front of fewer: requested with the input)
      pp_block<FAM, MODE>(in, U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
    }
  }
  else
  {
    // kPp8Bpw consecutive blocks per wave, the next block's input (and first records) requested before the current block is worked on
    const uint32_t b0 = xcd_tile(blockIdx.x, gridDim.x) * kPp8Bpw;
    if (b0 >= nBlocks) return;
    u32x4 x[4], xn[4];
    uint32_t rec0 = 0, rec0n = 0;
    pp_load(in, U, B, b0, x);
    if constexpr (MODE == 1) rec0 = sc.recs[(uint64_t)b0 * sc.recStride + threadIdx.x];
    for (uint32_t i = 0; i < kPp8Bpw; i++)
    {
      const uint32_t b = b0 + i;
      if (b >= nBlocks) break;
      const bool more = i + 1u < kPp8Bpw && b + 1u < nBlocks;
      if (more)
      {
        pp_load(in, U, B, b + 1u, xn);
        if constexpr (MODE == 1) rec0n = sc.recs[(uint64_t)(b + 1u) * sc.recStride + threadIdx.x];
      }
      pp_block<FAM, MODE>(in, U, B, b, sizes, offsets, payload, sc, sh, x, rec0);
      wave_sync();
#pragma unroll
      for (int j = 0; j < 4; j++) x[j] = xn[j];
      rec0 = rec0n;
    }
  }
}

} // namespace hsrle
