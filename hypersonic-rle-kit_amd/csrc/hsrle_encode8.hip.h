// hsrle_encode8.hip.h -- the 8 bit multi-symbol encoders (rle8_multi, rle8_packed_multi, rle8_{3,7}symlut) with the block's
// input staged through LDS and the output assembled in registers.
//
// Replaces: src/rle8_extreme_cpu.h:86-344 (wrapper, scalar tail, final block), :936-1099 (canonical AVX2 body),
//           src/rleX_Xsl.h:114-264 (process_symbol), :269-346, :421-485 (TYPE_SIZE 8 instantiation).
//
// Same execution model as everything else here: one lane = one block = one reference stream, 64 blocks per wavefront, the
// emit decisions run as the sequential state machine they are.  Data path:
//   HBM --(top-up: 4 adjacent lanes read 64 contiguous, aligned input bytes of ONE block per step)--> LDS history ring [64][256]
//   ring --(run detection: 64 positions per step: x ^ (x >> 8) == 0 on aligned 16-byte reads, the GPU form of the
//           reference's cmpeq + movemask + ctz scan, rle8_extreme_cpu.h:952-1084; bit masks of run starts / run ends)-->
//   packets: header bytes are assembled in registers, literal bytes come from the ring through a 128-bit byte funnel; both are
//           appended to a 16-byte output accumulator, and every completed 16-byte chunk goes straight to the block's staging
//           slot in HBM (the compressed side is the small side, so per-lane stores are not what limits the kernel).
// The kernel is latency bound like the decoder (throughput scales linearly with waves per CU), so LDS is spent on nothing but
// the ring: 16 KB per wave (XOR-swizzled rows, no pad, no mirror) + 1.5 KB of small tables (row scalars, merge masks, the
// accumulator hand-back of the slow literal path) = 17.9 KB = 14 LDS granules of 1 280 bytes = 9 waves per CU.
// Literal gaps that have left the ring (long runs, incompressible stretches) are read from global memory instead.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low, wave_sync

#ifndef HSRLE_ENC8_RING
#define HSRLE_ENC8_RING 256
#endif

namespace hsrle {

#ifdef HSRLE_ENC_STAMPS   // diagnostic build (never shipped): per-phase cycle sums of lane 0 of every wave of k_encode8_blocks
__device__ unsigned long long g_enc_stamps[8];
#define HS_ESTAMP(slot) { const unsigned long long tq_ = __builtin_readcyclecounter(); est[slot] += tq_ - et0; et0 = tq_; }
#else
#define HS_ESTAMP(slot)
#endif

// (which ring the encoders of 1 and 2 byte symbols use for an input: hsrle_ring_probe.hip.h)
// MONO = true: the lanes encode consecutive CHUNKS of ONE monolithic reference stream instead of independent blocks (hsrle_mono_encode.hip.h
// finds the chunk boundaries).  Chunk c covers the input bytes [monoStarts[c], monoStarts[c + 1]); every boundary is the end of a run
// that every encoder state emits (count >= LONG: SURVEY.md A.4 "reset points"), so the state in front of a chunk is known without the
// chunks before it -- lastRLE = the boundary, lastSymbol = that run's symbol (monoSyms) -- and the chunk's packets are exactly the
// packets the sequential encoder writes for these bytes.  A chunk writes no stream header and (unless it reaches the end of the input)
// no terminator, its rules see the TRUE end of the input (the AVX2 body / tail split of A.5 q1 counts from there), and its output goes
// to slots + monoSlotOff[c].
// RING: bytes of history ring per lane (256, or 128 for 16 instead of 9 waves per CU).  ringSel != nullptr: the host launches both
// instantiations and this one only runs if ringSel[0] == RING (k_ring_probe / k_ring_decide below chose from a sample of the input).
template <int FAM, bool MONO = false, int RING = HSRLE_ENC8_RING>
__global__ __launch_bounds__(64) void k_encode8_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                       uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes,
                                                       const uint64_t *__restrict__ monoStarts, const uint64_t *__restrict__ monoSyms,
                                                       const uint64_t *__restrict__ monoSlotOff, uint32_t monoSteps, uint64_t *__restrict__ monoListOut, uint32_t monoDry,
                                                       const uint32_t *__restrict__ ringSel)
{
  if (!MONO && ringSel != nullptr && ringSel[0] != (uint32_t)RING) return;
  // MONO with B != 0: the chunks are pieces of the BLOCKS of a container, not of one stream (hsrle_mono_encode.hip.h, "split encode": small
  // containers have too few blocks for one lane each).  Block starts are chunk starts; a block's first chunk writes the block's stream
  // header, its last one the terminator; the rules' "end of the input" is the end of the chunk's block.  ringSel[0] then holds the
  // number of chunks (the grid is sized for the most there can be).  Repair rounds of the list codecs (monoSteps >> 16 = r != 0) have nothing
  // to do when ringSel[7 + r] == 0 (what the guess / verify kernel in front of them counted): every workgroup returns at once.
  if constexpr (MONO) { if (B != 0u && ringSel != nullptr && (monoSteps >> 16) != 0u && ringSel[7u + (monoSteps >> 16)] == 0u) return; }
  using TR = Traits<FAM, 1, 0>;
  // Codecs with a move-to-front list: the list in front of a chunk is NOT known from the boundary run; the host hands every chunk a list
  // (monoSyms[8 * c + k]: entry k; [8 * c + 7]: encode this chunk?), gets the list behind it back (monoListOut[8 * c + k]; [.. + 7]: mtfDepth) and repeats the chunks whose
  // incoming list was not what the chunk in front left behind (hsrle_capi.hip: mono_encode_dev).  monoDry: no stores, only the list.
  [[maybe_unused]] const bool dry = MONO && monoDry != 0u;
  constexpr int Q = 64;                      // input bytes per lane and step
  constexpr int H = RING;                    // history ring per lane (power of two)
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr int K = TR::K;

  __shared__ __attribute__((aligned(16))) uint8_t hist[64 * H];
  __shared__ __attribute__((aligned(16))) uint32_t rinfo[64];
  __shared__ __attribute__((aligned(16))) uint8_t accScratch[64 * 16];   // see emit_literals
  // merge masks from a 16-entry table (as in k_decode_blocks: one ds_read_b128 instead of ~9 VALU; +1 % encode throughput)
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
#define HS_EMERGE(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = xcd_tile(blockIdx.x, gridDim.x) * 64u;    // XCD-aware tile order (hsrle_common.hip.h)
  const uint32_t b = wgFirst + lane;
  bool active = b < nBlocks;
  if constexpr (MONO) { if (B != 0u && ringSel != nullptr) active = b < ringSel[0]; }
  if constexpr (MONO && Traits<FAM, 1, 0>::kMtf) { if (active) active = monoSyms[8ull * b + 7] != 0ull; }   // the host's repair rounds switch most chunks off

  // ring byte x of row r lives at hist[(r * H) ^ hsw(r) ^ (x & HM)]: chunks XOR-swizzled by the row index (bank spread without pad)
  auto hsw_of = [](uint32_t r) -> uint32_t { return (r & 7u) << 4; };
  const uint32_t hbase = (lane * (uint32_t)H) ^ hsw_of(lane);

  uint32_t n = 0;
  [[maybe_unused]] uint32_t nTrueV = 0;                                 // MONO: bytes from the chunk start to the end of the input
  uint64_t blockAt = (uint64_t)b * B;
  if (active)
  {
    if constexpr (MONO)
    {
      blockAt = monoStarts[b];
      n = (uint32_t)(monoStarts[b + 1] - blockAt);
      nTrueV = (uint32_t)(U - blockAt);
      if (B != 0u)
      {
        const uint64_t blockEnd = (blockAt / B + 1ull) * B;
        if (blockEnd < U) nTrueV = (uint32_t)(blockEnd - blockAt);
      }
    }
    else
    {
      const uint64_t start = (uint64_t)b * B;
      n = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
    }
  }
  const uint32_t nTrue = MONO ? nTrueV : n;
  uint8_t *const slot = MONO ? slots + (active ? monoSlotOff[b] : 0ull) : slots + (uint64_t)b * slotStride;

  // ---- per-lane encoder state ----
  uint32_t avail = 0;        // input bytes [.., avail) are (or were) in the ring; the ring holds [avail - H, avail)
  uint32_t cb = 0;           // base of the window scanned next (multiple of 16)
  bool inRun = false;
  uint32_t runStart = 0, sym = 0;
  uint32_t lastRLE = 0;
  uint32_t lastSym = (MONO && active && !TR::kMtf) ? (uint32_t)(monoSyms[b] & 0xFFull) : 0u;      // Packed: lastSymbol (starts 0, A.5 q5)
  [[maybe_unused]] uint64_t lutw = (K == 3) ? 0x0000000000FF7F00ull : 0x00FE807E01FF7F00ull; // LUT: MTF list, entry k in byte k
  // MONO: how many leading list entries were put there by runs of this chunk (the rest is what the chunk was handed, in its order): the
  // list behind a chunk as a function of the list in front of it, as far as the chunk's own decisions did not depend on it
  [[maybe_unused]] uint32_t mtfDepth = 0;
  if constexpr (MONO && TR::kMtf)
  {
    if (active)
    {
      lutw = 0ull;
#pragma unroll
      for (int j = 0; j < K; j++) lutw |= (monoSyms[8ull * b + j] & 0xFFull) << (8 * j);
    }
  }
  bool ended = false;        // the end terminator has been written
  bool finished = !active;   // the whole stream is in the slot
  uint64_t winStarts = 0;    // current window: run-start bits

  // ---- output: 16-byte accumulator + stream position; completed chunks go to the slot ----
  const u32x4 zero4 = u32x4{ 0, 0, 0, 0 };
  u32x4 oacc = zero4;        // the chunk that contains stream position opos (its low opos & 15 bytes are valid)
  uint32_t opos = 0;         // stream bytes produced so far
  // The literals behind the last stored run (input without runs: the whole block) have mostly left the ring when the block is through.
  // Fetching them in the lane's own code costs a memory latency per 16 bytes with one lane of the wave active (random bytes: 0.45 TiB/s),
  // so finish_literals only NOTES a stretch of >= kNotedLiteralMin bytes and the whole wave copies the noted stretches when all its lanes
  // are through (coop_flush; first done for the Single encoders, hsrle_encode8s.hip.h).  A noted stretch ends at a 16-byte boundary of
  // the stream; the bytes behind it (< 16) are fetched into the accumulator.  Literals between two stored runs are NOT noted: with
  // the noting code inlined into handle_run the encoders lost 4 - 7 % on every input (same-box A/B), whatever the threshold.
  uint32_t pendSrc = 0, pendDst = 0, pendBytes = 0, pend2Src = 0, pend2Dst = 0, pend2Bytes = 0;
  auto store_bytes = [&](uint8_t *p, u32x4 w, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
    const uint64_t w0 = (uint64_t)w.x | ((uint64_t)w.y << 32), w1 = (uint64_t)w.z | ((uint64_t)w.w << 32);
    if (!dry)
      for (uint32_t k = lo; k < hi; k++)
        p[k] = (uint8_t)((k < 8u ? w0 >> (8u * k) : w1 >> (8u * (k - 8u))) & 0xFFull);
  };

  // timing-only diagnostic builds (output is wrong): -DHSRLE_ABLATE_ENC_STORES=1 no chunk store is ever executed; =2 one chunk in four;
  // =3 every chunk, but each lane only ever writes the first 128 bytes of its slot (the memory side without the open lines)
#if defined(HSRLE_ABLATE_ENC_STORES) && HSRLE_ABLATE_ENC_STORES == 1
#define HS_ENC_ST128(p, at, w) { if (U == 0x7FFFFFFFFFFFFFF1ull) st128((p) + (at), w); }
#elif defined(HSRLE_ABLATE_ENC_STORES) && HSRLE_ABLATE_ENC_STORES == 2
#define HS_ENC_ST128(p, at, w) { if ((((uint32_t)((p) - slot) + (at)) & 48u) == 48u) st128((p) + (at), w); }
#elif defined(HSRLE_ABLATE_ENC_STORES) && HSRLE_ABLATE_ENC_STORES == 3
#define HS_ENC_ST128(p, at, w) { st128(slot + (((uint32_t)((p) - slot) + (at)) & 112u), w); }
#else
#define HS_ENC_ST128(p, at, w) st128((p) + (at), w)
#endif
  // append the low nb (<= 12) bytes of hv
  auto append = [&](u32x4 hv, uint32_t nb) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);   // hv << c bytes
    const u32x4 w = HS_EMERGE(oacc, lowp, c);
    if (c + nb >= 16u)
    {
      if (!dry) HS_ENC_ST128(slot, (opos & ~15u), w);
      oacc = (c == 0u) ? zero4 : funnel16(hv, zero4, 16u - c);          // hv >> (16 - c) bytes
    }
    else
      oacc = w;
    opos += nb;
  };

  // packet header under construction: up to 12 bytes, little endian
  uint64_t hlo = 0;
  uint32_t hhi = 0, hn = 0;
  auto hpush = [&](uint32_t v, uint32_t k) {                            // the low k (1, 2 or 4) bytes of v
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
  auto h16 = [&](uint32_t v) { hpush(v & 0xFFFFu, 2u); };
  auto h32 = [&](uint32_t v) { hpush(v, 4u); };
  auto hflush = [&]() {
    append(u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), hhi, 0u }, hn);
    hlo = 0; hhi = 0; hn = 0;
  };

  // 16 input bytes at block position p (p may reach below 0 or beyond the input: those bytes are never used)
  auto ring_win = [&](uint32_t p) -> u32x4 {
    const uint32_t a0 = p & ~15u;
    return funnel16(lds_ld128(hist + (hbase ^ (a0 & HM))), lds_ld128(hist + (hbase ^ ((a0 + 16u) & HM))), p & 15u);
  };
  // literal bytes [from, from + len) of the block: from the ring while they are still there.  Literals that have left the ring (a long
  // stretch of runs too short to be stored) are read from global memory by a function that is kept out of line: with the two
  // sources selected per chunk inside one loop the rare path cost the common one 10 % of the kernel.
  auto emit_literals = [&](uint32_t from, uint32_t len, bool mayNote) {
    if (len == 0u) return;
    const uint32_t c = opos & 15u, total = c + len;
    const uint32_t srcp = from - c;
    uint8_t *const dst = slot + (opos & ~15u);
    if (__builtin_expect(from + (uint32_t)H >= avail + 16u, 1))
    {
      u32x4 w = HS_EMERGE(oacc, ring_win(srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        if (!dry) HS_ENC_ST128(dst, k, w);
        k += 16u;
        if (k < total) w = ring_win(srcp + k);
      }
      oacc = w;
    }
    else if (mayNote && len >= kNotedLiteralMin && (pendBytes == 0u || pend2Bytes == 0u))
    {
      // what the accumulator holds goes out now; the stretch is noted up to the last 16-byte boundary of the stream it reaches; the
      // bytes behind that boundary come into the accumulator (through LDS: see below)
      if (c != 0u) store_bytes(dst, oacc, 0u, c);
      const uint32_t tail = (opos + len) & 15u, noted = len - tail;
      if (!dry)
      {
        if (pendBytes == 0u) { pendSrc = from; pendDst = opos; pendBytes = noted; }
        else { pend2Src = from; pend2Dst = opos; pend2Bytes = noted; }
      }
      lds_st128(accScratch + lane * 16u, tail != 0u ? global_window16(in, blockAt, U, from + noted) : zero4);
      oacc = lds_ld128(accScratch + lane * 16u);
    }
    else
    {
      u32x4 w = HS_EMERGE(oacc, global_window16(in, blockAt, U, srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        if (!dry) st128(dst + k, w);
        k += 16u;
        if (k < total) w = global_window16(in, blockAt, U, srcp + k);
      }
      // the accumulator is handed back through LDS: a register that may hold a pending vector-memory load makes the compiler wait
      // for ALL outstanding loads -- the input prefetch included -- in front of every store of the common path (-10 % encode)
      lds_st128(accScratch + lane * 16u, w);
      oacc = lds_ld128(accScratch + lane * 16u);
    }
    opos += len;
  };

  // wave-converged: copy the noted stretches, 16 bytes per lane and load (any alignment), two stretches in flight; coop_flush(false) only
  // acts once some lane has both of its slots in use
  const uint64_t slotOff = (uint64_t)(slot - slots);
  auto coop_flush = [&](bool all) __attribute__((always_inline)) {
    if (__builtin_expect(__ballot(pendBytes != 0u && (all || pend2Bytes != 0u)) == 0ull, 1)) return;
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
        auto lane64 = [&](uint64_t v, uint32_t l) __attribute__((always_inline)) -> uint64_t {
          return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32);
        };
        const uint8_t *const sp0 = in + lane64(blockAt, l0) + (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l0);
        const uint8_t *const sp1 = in + lane64(blockAt, l1) + (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)l1);
        uint8_t *const dp0 = slots + lane64(slotOff, l0) + (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l0);
        uint8_t *const dp1 = slots + lane64(slotOff, l1) + (uint32_t)__builtin_amdgcn_readlane((int)pd, (int)l1);
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l0);
        const uint32_t n1 = (l1 != l0) ? (uint32_t)__builtin_amdgcn_readlane((int)pb, (int)l1) : 0u;
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
        const uint32_t r = lane & 15u;                                    // the last n & 15 bytes: lanes 0..15 / 16..31
        if (lane < 16u) { if (r < (n0 & 15u)) dp0[f0 + r] = sp0[f0 + r]; }
        else if (lane < 32u) { if (r < (n1 & 15u)) dp1[f1 + r] = sp1[f1 + r]; }
      }
    }
    pendBytes = 0u; pend2Bytes = 0u;
  };

  // ---- stream header ----
  const bool blockFirst = MONO && B != 0u && active && (blockAt % B) == 0ull;     // split encode: this chunk opens its block's stream
  if (active && (!MONO || blockFirst))
  {
    h32(MONO ? nTrue : n);                                              // (the block's length; its compressed length is patched in at placement)
    h32(0);
    if constexpr (!TR::kLut && !TR::kShort) hb(0); // mode = multi
    hflush();
  }

  // ---- per-row scalars for the lanes that serve a row (same scheme as the decoder's publish()) ----
  auto publish = [&](uint32_t v) { rinfo[(lane % (uint32_t)RPL) * (uint32_t)LPR + lane / (uint32_t)RPL] = v; };

  // ---- input top-up (4 lanes per row read 64 contiguous bytes; the loads fly during the step's scan) ----
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0;
  // per served row, once: where its input starts (+ this lane's chunk within a 64-byte piece), how many bytes are readable from there, its
  // ring row (round 3: the 64 bit products and compares were 70 VALU of every step)
  const uint8_t *rowPtr[LPR];
  uint32_t rowLim[LPR], rowLds[LPR];
#pragma unroll
  for (int q = 0; q < LPR; q++)
  {
    const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = lane % LPR;
    uint64_t at = (uint64_t)(wgFirst + r) * B;
    if constexpr (MONO)
    {
      const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)blockAt, (int)r, 64), hi32 = (uint32_t)__shfl((int)(uint32_t)(blockAt >> 32), (int)r, 64);
      at = ((uint64_t)hi32 << 32) | lo32;
    }
    at += c * 16u;
    rowPtr[q] = in + at;
    rowLim[q] = (at < U) ? (uint32_t)(((U - at) < 0xFFFFFFFFull) ? (U - at) : 0xFFFFFFFFull) : 0u;
    rowLds[q] = (r * (uint32_t)H) ^ hsw_of(r);
  }

  auto issue = [&]() {
    const uint32_t left = (n > avail) ? (n - avail + 15u) >> 4 : 0u;
    wantReq = umin((uint32_t)LPR, left);
    publish(wantReq != 0u ? (avail | wantReq) : 0u);                   // avail is a multiple of 16 while chunks are left
    wave_sync();
    uint32_t ri[LPR];
#pragma unroll
    for (int q = 0; q < LPR; q++) ri[q] = rinfo[(lane / LPR) * LPR + q];
    wave_sync();
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      const uint32_t c = lane % LPR;
      const uint32_t nreq = ri[q] & 15u, e = ri[q] & ~15u;
      const bool valid = c < nreq;
      u32x4 v = u32x4{ 0, 0, 0, 0 };
      if (valid)
      {
        const uint8_t *const g = rowPtr[q] + e;
        if (__builtin_expect(e + 16u <= rowLim[q], 1))
          v = ld128(g);
        else if (U >= 16u)
          v = funnel16(ld128(in + U - 16u), v, (uint32_t)((uint64_t)(g - in) + 16u - U));   // the input's last, partial chunk: read at U - 16, shifted down
        else
        {
          uint64_t t0 = 0, t1 = 0;
#pragma unroll 1
          for (uint32_t k = 0; k < 16u && e + k < rowLim[q]; k++)
            if (k < 8u) t0 |= (uint64_t)g[k] << (8u * k); else t1 |= (uint64_t)g[k] << (8u * (k - 8u));
          v = u32x4{ (uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)t1, (uint32_t)(t1 >> 32) };
        }
      }
      pf[q] = v;
      pfAt[q] = rowLds[q] ^ ((e + c * 16u) & HM);
    }
  };

  auto land = [&]() {
    // the ring must keep the chunk being scanned (and one before it for the byte funnel)
    const uint32_t keep = (cb >= 16u) ? cb - 16u : 0u;
    const uint32_t fit = ((uint32_t)H - (avail - keep)) >> 4;
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

  [[maybe_unused]] uint32_t nrec = 0;
  // ---- one finished run [p, e): decide, and if emitted write the packet ----
  auto handle_run = [&](uint32_t p, uint32_t e) {
    const uint32_t count = e - p;
    const uint32_t gap = p - lastRLE;
    bool same = false, body = false;
    int k;                                                             // 0 keep as literals, 1 short range field, 2 long range field
    [[maybe_unused]] uint32_t m = 0, c7 = 0, r7 = 0, cst = 0, rng = 0;

    [[maybe_unused]] int32_t sc = 0;                                   // Short: stored count (count - SMINS + 2)
    [[maybe_unused]] bool pack1 = false;                               // Short: the one-byte header form

    if constexpr (TR::kShort)
    {
      // rleX_Xsl_short.h:152-197
      rng = gap + 2u;
      m = (uint32_t)K;
#pragma unroll
      for (int j = K - 1; j >= 0; j--)
        if (((lutw >> (8 * j)) & 0xFFull) == (uint64_t)sym) m = (uint32_t)j;
      sc = (int32_t)count - (int32_t)TR::SMINS + 2;
      pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
      uint32_t pen = (K > 0 && m == (uint32_t)K) ? 1u : 0u;
      if (!pack1)
      {
        pen += 2u;
        if (!(sc <= (int32_t)TR::SMAXTC && rng <= TR::SMAXTR))
          pen += ((rng <= 0xFFFFFu) ? (rng <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
      }
      k = (count >= TR::SMINL || count >= TR::SMINS + pen) ? 1 : 0;
    }
    else if constexpr (TR::kLut)
    {
      // rleX_Xsl.h:116-132
      rng = gap + 2u;
      m = (uint32_t)K;
#pragma unroll
      for (int j = K - 1; j >= 0; j--)
        if (((lutw >> (8 * j)) & 0xFFull) == (uint64_t)sym) m = (uint32_t)j;
      cst = count - 3u + 2u;
      constexpr uint32_t MAXC = 127u, MAXR = (1u << TR::RB) - 1u;
      uint32_t pen = (rng <= 0xFFFFFu) ? (rng <= MAXR ? 0u : 2u) : 4u;  // 0xFFFFF vs 0xFFFF in the writer: A.5 q3
      pen += (cst <= 0xFFFFFu) ? (cst <= MAXC ? 0u : 2u) : 4u;
      pen += (m == (uint32_t)K) ? 1u : 0u;
      k = (count >= 1u + 10u || count >= 3u + pen) ? 1 : 0;
      c7 = (cst <= MAXC) ? cst : (cst <= 0xFFFFu ? 1u : 0u);
      r7 = (rng <= MAXR) ? rng : (rng <= 0xFFFFu ? 1u : 0u);
    }
    else if constexpr (TR::kPacked)
    {
      // body / tail split of the canonical AVX2 encoder
      rng = gap + 1u;
      const int32_t kk = (int32_t)(count - 1u) / 32;
      body = (e < nTrue) && ((int32_t)p + 1 + 32 * kk < (int32_t)nTrue - 32);
      if (body)
      {
        same = sym == lastSym;
        const bool emit = count >= 11u || (rng <= 127u && ((same && count >= 3u) || count >= 4u));
        k = emit ? (rng <= 127u ? 1 : 2) : 0;
      }
      else
        k = (count >= 11u) ? (rng <= 127u ? 1 : 2) : 0;
    }
    else
    {
      rng = gap + 1u;
      k = (count >= 6u) ? (rng <= 255u ? 1 : 2) : 0;                   // rle8_extreme_cpu.h:974
    }

    if (!k)
      return;

#ifdef HSRLE_ENC_DECIDE_ONLY   // timing-only diagnostic build (round 5 pricing): decisions and sizes only, no header assembly, no literal emission
    if constexpr (TR::kPacked && !TR::kLut && !TR::kShort)
    {
      if (body) lastSym = sym;
      const uint32_t c = count - 3u + 1u;
      const uint32_t hbytes = (c <= 127u ? 1u : 5u) + (same ? 0u : 1u) + (k == 1 ? 1u : 4u);
#if HSRLE_ENC_DECIDE_ONLY >= 2  // + one 8-byte packet record per stored run
      *(uint64_t *)(slot + 16u + 8u * nrec) = (uint64_t)opos | ((uint64_t)lastRLE << 20) | ((uint64_t)gap << 36) | ((uint64_t)(same ? 1u : 0u) << 52) | ((uint64_t)(c & 0xFFu) << 56);
      nrec++;
#endif
      opos += hbytes + gap;
      lastRLE = e;
      if (e >= nTrue) { opos += 9u; ended = true; }
      return;
    }
#endif

    // ---- header ----
    if constexpr (TR::kShort)
    {
      // rleX_Xsl_short.h:199-357
      if constexpr (K > 0)
      {
        const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
        const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
        const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
        lutw = keepHi | (low << 8) | (uint64_t)sym;
        if (m >= mtfDepth && mtfDepth < (uint32_t)K) mtfDepth++;
      }
      const uint32_t mi = (K > 0) ? m << (TR::SCB + TR::SRBP) : 0u;
      if (pack1)
        hb(mi | ((uint32_t)(sc - 2) << TR::SRBP) | gap);
      else
      {
        const uint32_t scu = (uint32_t)sc;
        const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
        const uint32_t rx = (rng <= TR::SMAXTR) ? rng : (rng <= 0xFFFFu ? 1u : 0u);
        hb(mi | (TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8));
        hb((scx << (TR::SRB - 8u)) | (rx >> 8));
        hb(rx);
        if (scx != scu) { if (scu <= 0xFFFFu) h16(scu); else h32(scu); }
        if (rx != rng) { if (rng <= 0xFFFFu) h16(rng); else h32(rng); }
      }
      if (K == 0 || m == (uint32_t)K) hb(sym);
    }
    else if constexpr (TR::kLut)
    {
      const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
      const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
      const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
      lutw = keepHi | (low << 8) | (uint64_t)sym;
      if (m >= mtfDepth && mtfDepth < (uint32_t)K) mtfDepth++;
      h16((m << (K == 3 ? 14 : 13)) | (c7 << TR::RB) | r7);
      if (m == (uint32_t)K) hb(sym);
      if (cst != c7) { if (cst <= 0xFFFFu) h16(cst); else h32(cst); }
      if (rng != r7) { if (rng <= 0xFFFFu) h16(rng); else h32(rng); }
    }
    else if constexpr (TR::kPacked)
    {
      if (body) lastSym = sym;                                          // only the body rule tracks lastSymbol (A.3)
      const uint32_t c = count - 3u + 1u, sm = same ? 0x80u : 0u;
      if (c <= 127u) hb(c | sm); else { hb(sm); h32(c); }
      if (!same) hb(sym);
      if (k == 1) hb(rng << 1); else h32((rng << 1) | 1u);
    }
    else
    {
      const uint32_t c = count - 6u + 1u;
      hb(sym);
      if (c <= 255u) hb(c); else { hb(0); h32(c); }
      if (k == 1) hb(rng); else { hb(0); h32(rng); }
    }
    hflush();

    // ---- literals ----
    emit_literals(lastRLE, gap, false);
    lastRLE = e;

    if (e >= nTrue)
    {
      // end terminator (rle8_extreme_cpu.h:203-338; rleX_Xsl.h:319-338)
      if constexpr (TR::kShort) { hb(TR::SCINV << TR::SRBP); hb(TR::STB); hb(1); h16(0); h16(0); if (K == 0) hb(0); }   // rleX_Xsl_short.h:470-501
      else if constexpr (TR::kLut) { h16((1u << TR::RB) | 1u); h16(0); h16(0); }
      else if constexpr (TR::kPacked) { hb(0x80); h32(0); h32(1); }
      else { hb(0); hb(0); h32(0); hb(0); h32(0); }
      hflush();
      ended = true;
    }
  };

  // literal terminator carrying the bytes behind the last emitted run
  auto finish_literals = [&]() {
    const uint32_t kLit = n - lastRLE;
#ifdef HSRLE_ENC_DECIDE_ONLY
    if constexpr (TR::kPacked && !TR::kLut && !TR::kShort) { opos += 9u + kLit; return; }
#endif
    if constexpr (TR::kShort) { hb(TR::SCINV << TR::SRBP); hb(TR::STB); hb(0); h16(0); h32(kLit + 2u); if (K == 0) hb(0); }   // :503-523
    else if constexpr (TR::kLut) { h16(1u << TR::RB); h16(0); h32(kLit + 2u); }
    else if constexpr (TR::kPacked) { hb(0x80); h32(0); h32(((kLit + 1u) << 1) | 1u); }
    else { hb(0); hb(0); h32(0); hb(0); h32(kLit + 1u); }
    hflush();
    emit_literals(lastRLE, kLit, true);    // (the one place that notes: see above)
  };

  // ---- main loop ----
  issue();
  land();
  wave_sync();

  uint32_t stepsLeft = MONO ? ((B != 0u) ? (monoSteps & 0xFFFFu) : monoSteps) : 2u * (B / (uint32_t)Q) + 64u;  // bounded: every step scans a window or lands input

#ifdef HSRLE_ENC_STAMPS
  unsigned long long est[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, et0 = __builtin_readcyclecounter();
  uint32_t myTrips = 0, myRuns = 0;
#endif
  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    HS_ESTAMP(0)
    issue();
    HS_ESTAMP(1)

    // ---------------- scan what is in the ring ----------------
    // Phase A (uniform): equality mask of up to 64 positions -> bit masks of run starts and run ends.
    // Phase B (per lane): one handle_run per run END -- a lane's trip count is the number of runs that end in its window.
    if (!finished)
    {
      // window [cb, cb + W): every position needs its successor byte (or the end of the input)
      const uint32_t lastStep = (avail >= n) ? 1u : 0u;
      uint32_t W = lastStep ? umin(64u, n - cb) : umin(64u, ((avail - 1u - cb) >> 4) << 4);
      if (cb >= n) W = 0;
      if (W != 0u)
      {
        uint64_t e64 = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++)
        {
          if (j * 16u < W)
          {
            const u32x4 x = lds_ld128(hist + (hbase ^ ((cb + j * 16u) & HM)));
            const uint32_t x4 = lds_ld32(hist + (hbase ^ ((cb + j * 16u + 16u) & HM)));
            e64 |= (uint64_t)zero_mask16(x.x ^ alignbyte(x.y, x.x, 1), x.y ^ alignbyte(x.z, x.y, 1), x.z ^ alignbyte(x.w, x.z, 1), x.w ^ alignbyte(x4, x.w, 1)) << (16u * j);
          }
        }
        // position i is a match only if cb + i + 1 < n (bytes at or beyond n never match)
        const uint32_t validBits = (n - cb > W) ? W : (n - cb - 1u);
        e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
        const uint64_t wmask = (W >= 64u) ? ~0ull : ((1ull << W) - 1ull);
        HS_ESTAMP(2)
        const uint64_t prev = (e64 << 1) | (inRun ? 1ull : 0ull);      // "the position before me matched"
        winStarts = e64 & ~prev;
        uint64_t pendingEnds = ~e64 & prev & wmask;                     // bit i: a run ends with position i (exclusive end cb + i + 1)

        while (pendingEnds != 0ull)
        {
#ifdef HSRLE_ENC_STAMPS
          { const unsigned long long am_ = __ballot(true); if (lane == (uint32_t)__builtin_ctzll(am_)) myTrips++; myRuns++; }
#endif
          const uint32_t i = (uint32_t)__builtin_ctzll(pendingEnds);
          const uint64_t sBelow = winStarts & ((2ull << i) - 1ull);
          uint32_t st = runStart, sy = sym;
          if (sBelow != 0ull)
          {
            st = cb + (63u - (uint32_t)__builtin_clzll(sBelow));
            sy = hist[hbase ^ (st & HM)];
          }
          sym = sy;
          handle_run(st, cb + i + 1u);
          pendingEnds &= pendingEnds - 1ull;
        }

        // the window is done: remember a run that is still open at its end (its last position matches its successor)
        if (((e64 >> (W - 1u)) & 1ull) != 0ull)
        {
          // the open run is the last one that started in this window; with no start at all the carried run goes on
          if (winStarts != 0ull)
          {
            runStart = cb + (63u - (uint32_t)__builtin_clzll(winStarts));
            sym = hist[hbase ^ (runStart & HM)];
          }
          inRun = true;
        }
        else
          inRun = false;
        cb += W;
      }

      if (cb >= n && avail >= n)
      {
        // end of input: a run that reaches the end is judged now; then the literal terminator unless the stream ended
        if (inRun) { handle_run(runStart, n); inRun = false; }
        if (!ended && n == nTrue) { finish_literals(); ended = true; }      // (a MONO chunk that does not reach the end of the input ends with its boundary run's packet)
        // the last partial chunk, then the stream size (header field compressedLength and the size table)
        if ((opos & 15u) != 0u && !dry)
          st128(slot + (opos & ~15u), oacc);
        if constexpr (!MONO) st32(slot + 4, opos);
        if (!dry) sizes[b] = opos;
        if constexpr (MONO && TR::kMtf)
        {
#pragma unroll
          for (int j = 0; j < K; j++) monoListOut[8ull * b + j] = (lutw >> (8 * j)) & 0xFFull;
          monoListOut[8ull * b + 7] = mtfDepth;
        }
        finished = true;
      }
    }

    wave_sync();
    HS_ESTAMP(3)
    land();
    wave_sync();
    HS_ESTAMP(4)
#ifdef HSRLE_ENC_STAMPS
    est[5] += 1;
#endif
  }
  coop_flush(true);
#ifdef HSRLE_ENC_STAMPS
  if (lane == 0u) { for (int q = 0; q < 6; q++) atomicAdd(g_enc_stamps + q, est[q]); atomicAdd(g_enc_stamps + 6, 1ull); }
  atomicAdd(g_enc_stamps + 7, ((unsigned long long)myTrips << 32) | myRuns);   // wave-level trips of the run-end loop | lane-level run ends
#endif
}

} // namespace hsrle
