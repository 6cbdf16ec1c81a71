// hsrle_encodeS.hip.h -- the multi-symbol encoders for 2/3/4/6/8-byte symbols (plain, Packed, 3/7-symbol LUT; sym- and
// byte-aligned) with the block's input staged through LDS and the output assembled in registers: the data path of
// k_encode8_blocks (hsrle_encode8.hip.h) with the run enumeration of the wide codecs.
//
// Replaces: src/rleX_extreme_cpu_encode.h:14-609 (16/32/64 bit), src/rle24_extreme_cpu_encode.h, src/rle48_extreme_cpu_encode.h
//           (24/48 bit), src/rleX_Xsl_multibyte_encoder.h:18-370 + src/rleX_Xsl.h:114-264 (LUT).
//
// Run enumeration (restated by oracle/hsrle_oracle.c:runs_next): with the match bits m[j] = (d[j] == d[j + S])
// (false at and beyond n - S), a run starts at the first p >= i with S consecutive set bits, i.e. at the start of a stretch of
// L >= S set bits; it ends at e = p + S + S * floor(L / S) (whole symbols) plus, for the byte-aligned variants and only while a
// whole further symbol would still fit in the input, the L mod S matching leading bytes.  The search continues at i = e
// whether or not the run was emitted.  Here every lane computes the match bits of a 64-position window from the LDS ring
// (SWAR compare of each 16-byte chunk with the bytes S further on) and walks the stretches of set bits; a stretch that is still
// open at the window end is carried by its start position.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // funnel16, merge_low, wave_sync

namespace hsrle {

// MONO = true: the lanes encode consecutive chunks of ONE monolithic stream (hsrle_mono_encode.hip.h; see k_encode8_blocks): chunk table
// instead of b * B, the symbol in front of the chunk as lastSymbol, no stream header, no terminator unless the chunk reaches the end of
// the input, and the rules that look at the end of the input (the partial-symbol extension of the byte-aligned variants, the end
// terminator) see the TRUE end.
#ifdef HSRLE_ENCS_RING
#define HSRLE_ENCS_RING_OF(S) HSRLE_ENCS_RING
#else
#define HSRLE_ENCS_RING_OF(S) ((S) >= 3 ? 128 : 256)
#endif
template <int FAM, int S, int AL, bool MONO = false, int RING = HSRLE_ENCS_RING_OF(S)>
__global__ __launch_bounds__(64) void k_encodeS_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                       uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes,
                                                       const uint64_t *__restrict__ monoStarts, uint64_t *monoSyms,
                                                       const uint64_t *__restrict__ monoSlotOff, uint32_t monoSteps, uint64_t *monoListOut, uint32_t monoDry,
                                                       const uint32_t *__restrict__ ringSel)
{
  if (!MONO && ringSel != nullptr && ringSel[0] != (uint32_t)RING) return;          // (S = 2: the host launches both rings, k_ring_decide chose: hsrle_encode8.hip.h)
  // (codecs with a move-to-front list: monoSyms[8 * c + k] = entry k of the list in front of chunk c, monoListOut likewise the list behind it;
  //  monoDry: no stores, only the list -- see k_encode8_blocks.  Neither is const nor __restrict__: when the lists are settled in the kernel
  //  (grp != 0 below) the lanes of a wave rewrite monoSyms and read each other's monoListOut between rounds.)
  [[maybe_unused]] const bool dry = MONO && monoDry != 0u;
  using TR = Traits<FAM, S, AL>;
  static_assert(S == 2 || S == 3 || S == 4 || S == 6 || S == 8, "8 bit and 128 bit symbols have their own kernels");
  static_assert(FAM == PLAIN || FAM == PACKED || FAM == LUT3 || FAM == LUT7 || (FAM >= SHORT0 && FAM <= SHORT7), "multi-symbol families only");
  constexpr int Q = 64;                      // input bytes per lane and step
  // History ring per lane (power of two).  The kernel is a latency chain, so LDS is waves: 128 bytes per lane give 16 instead of 9 waves
  // per CU -- and cost the literals in front of a run whose end lies more than ~100 bytes behind them a fetch from global memory.  With
  // symbols of 3 bytes and more the runs are that long anyway (8 GiB, same-box A/B, run-distributed / video-shaped: rle32_byte_packed
  // +17 % / +32 %, rle64_3symlut_byte +21 % / +26 %); with 1 and 2 byte symbols the run-distributed data loses (rle16_sym -11 %,
  // the 8 bit codecs -17 %) what the video-shaped data gains (+28 ... +31 %): those choose per input.  HSRLE_ENCS_RING overrides (A/B builds).
  // For S = 2 the host chooses per input (RING).
  constexpr int H = RING;
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr int K = TR::K;
  constexpr uint32_t SU = (uint32_t)S;

  __shared__ __attribute__((aligned(16))) uint8_t hist[64 * H];
  __shared__ __attribute__((aligned(16))) uint32_t rinfo[64];
  __shared__ __attribute__((aligned(16))) uint8_t accScratch[64 * 16];   // see emit_literals
  // merge masks from a 16-entry table (as in k_decode_blocks / k_encode8_blocks)
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
#define HS_SMERGE(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = xcd_tile(blockIdx.x, gridDim.x) * 64u;    // XCD-aware tile order (hsrle_common.hip.h)
  if constexpr (MONO) { if (B != 0u && ringSel != nullptr && (monoSteps >> 16) != 0u && ringSel[7u + (monoSteps >> 16)] == 0u) return; }   // an empty repair round (see k_encode8_blocks)
  // Split encode of a list codec with the lists settled IN the kernel (ringSel[1] = grp != 0): a wave takes the chunks of grp whole blocks
  // (their chunks are consecutive: lane = chunk - first chunk of the wave's first block; ringSel + 64 = first chunk of every block, ringSel[2] = blocks;
  // the host chooses grp so that grp blocks never have more than 64 chunks), so every chunk
  // a chunk's list depends on sits in the same wave -- after a pass the lanes compose the lists from what the chunks in front of them left
  // (as k_split_list_guess does between launches) and the chunks whose list changed go round again: no proof launches that find nothing
  // (4 x 21 us of the 88 MB frame's 332).  All other modes make one round.
  [[maybe_unused]] const uint32_t grp = (MONO && B != 0u && ringSel != nullptr) ? ringSel[1] : 0u;
  for (uint32_t listRound = 0;; listRound++)
  {
  uint32_t b = wgFirst + lane;
  bool active = b < nBlocks;
  [[maybe_unused]] uint32_t blockFirstChunk = 0;
  [[maybe_unused]] bool mine = false;
  if constexpr (MONO)
  {
    if (grp != 0u)
    {
      const uint32_t *const first = ringSel + 64;
      const uint32_t blk0 = blockIdx.x * grp, blk1 = umin(blk0 + grp, ringSel[2]);
      active = false;
      if (blk0 < blk1)
      {
        const uint32_t c0 = first[blk0];
        if (lane < first[blk1] - c0) { b = c0 + lane; active = true; mine = true; blockFirstChunk = first[(uint32_t)(monoStarts[b] / B)]; }
      }
    }
    else if (B != 0u && ringSel != nullptr) active = b < ringSel[0];   // split encode (see k_encode8_blocks): chunks of a container's blocks, their number on the device
  }
  if constexpr (MONO && Traits<FAM, S, AL>::kMtf) { if (active) active = ld_fresh64(monoSyms + 8ull * b + 7) != 0ull; }   // repair rounds switch most chunks off

  // ring byte x of row r lives at hist[(r * H) ^ hsw(r) ^ (x & HM)]: chunks XOR-swizzled by the row index (bank spread without pad)
  auto hsw_of = [](uint32_t r) -> uint32_t { return (r & 7u) << 4; };
  const uint32_t hbase = (lane * (uint32_t)H) ^ hsw_of(lane);

  uint32_t n = 0;
  [[maybe_unused]] uint32_t nTrueV = 0;
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
  uint32_t from = 0;         // run search position i: match bits below it are ignored
  bool open = false;         // a stretch of set match bits is open ...
  uint32_t sStart = 0;       // ... since this position
  uint32_t sy0 = 0, sy1 = 0; // the symbol at sStart (low S bytes)
  uint32_t lastRLE = 0;
  [[maybe_unused]] uint32_t la0 = 0, la1 = 0;  // Packed: last emitted symbol (starts as zeros)
  if constexpr (MONO && !Traits<FAM, S, AL>::kMtf) { if (active) { const uint64_t v = monoSyms[b]; la0 = (uint32_t)v; la1 = (uint32_t)(v >> 32); } }
  bool ended = false;        // the end terminator has been written
  bool finished = !active;   // the whole stream is in the slot

  [[maybe_unused]] uint32_t lut0[K ? K : 1], lut1[K ? K : 1];          // LUT: move-to-front list, entry k = {lut0[k], lut1[k]}
  if constexpr (TR::kMtf)
  {
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const uint32_t b4 = init[k] * 0x01010101u;
      lut0[k] = (S >= 4) ? b4 : (b4 & ((1u << (8 * (S & 3))) - 1u));
      lut1[k] = (S == 8) ? b4 : (S == 6 ? (b4 & 0xFFFFu) : 0u);
    }
  }

  [[maybe_unused]] uint32_t mtfDepth = 0;   // MONO: leading list entries that came from runs of this chunk (see k_encode8_blocks)
  if constexpr (MONO && TR::kMtf)
  {
    if (active)
    {
#pragma unroll
      for (int k = 0; k < K; k++) { const uint64_t v = ld_fresh64(monoSyms + 8ull * b + k); lut0[k] = (uint32_t)v; lut1[k] = (uint32_t)(v >> 32); }
    }
  }

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
  [[maybe_unused]] bool tailNoted = false;   // MONO: the stream's last bytes are part of a noted stretch (no store of the last partial chunk)
  auto store_bytes = [&](uint8_t *p, u32x4 w, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
    const uint64_t w0 = (uint64_t)w.x | ((uint64_t)w.y << 32), w1 = (uint64_t)w.z | ((uint64_t)w.w << 32);
    if (!dry)
      for (uint32_t k = lo; k < hi; k++)
        p[k] = (uint8_t)((k < 8u ? w0 >> (8u * k) : w1 >> (8u * (k - 8u))) & 0xFFull);
  };

  // append the low nb (<= 16) bytes of hv
  auto append = [&](u32x4 hv, uint32_t nb) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);   // hv << c bytes
    const u32x4 w = HS_SMERGE(oacc, lowp, c);
    if (c + nb >= 16u)
    {
      if (!dry) st128(slot + (opos & ~15u), w);
      oacc = (c == 0u) ? zero4 : funnel16(hv, zero4, 16u - c);          // hv >> (16 - c) bytes
    }
    else
      oacc = w;
    opos += nb;
  };

  // packet header under construction: up to 16 bytes, little endian (flushed early when a field would not fit)
  uint64_t hlo = 0, hhi = 0;
  uint32_t hn = 0;
  auto hflush = [&]() {
    if (hn != 0u)
      append(u32x4{ (uint32_t)hlo, (uint32_t)(hlo >> 32), (uint32_t)hhi, (uint32_t)(hhi >> 32) }, hn);
    hlo = 0; hhi = 0; hn = 0;
  };
  auto hpush = [&](uint64_t v, uint32_t k) {                            // the low k (1..8) bytes of v; v has no bits above them
    if (hn + k > 16u) hflush();
    if (hn < 8u)
    {
      const uint32_t sh = hn * 8u;
      hlo |= v << sh;
      if (hn + k > 8u) hhi |= v >> (64u - sh);                          // hn >= 1 here, so the shift is < 64
    }
    else
      hhi |= v << ((hn - 8u) * 8u);
    hn += k;
  };
  auto hb = [&](uint32_t v) { hpush((uint64_t)(v & 0xFFu), 1u); };
  auto h16 = [&](uint32_t v) { hpush((uint64_t)(v & 0xFFFFu), 2u); };
  auto h32 = [&](uint32_t v) { hpush((uint64_t)v, 4u); };
  auto hsym = [&](uint32_t s0, uint32_t s1) { hpush((uint64_t)s0 | ((uint64_t)s1 << 32), SU); };

  // 16 input bytes at block position p (p may reach below 0 or beyond the input: those bytes are never used)
  auto ring_win = [&](uint32_t p) -> u32x4 {
    const uint32_t a0 = p & ~15u;
    return funnel16(lds_ld128(hist + (hbase ^ (a0 & HM))), lds_ld128(hist + (hbase ^ ((a0 + 16u) & HM))), p & 15u);
  };
  // literal bytes [at, at + len) of the block: from the ring while they are still there.  Literals that have left the ring (a long
  // stretch of runs too short to be stored) are read from global memory by a function that is kept out of line: with the two
  // sources selected per chunk inside one loop the rare path cost the common one 10 % of the kernel.
  auto emit_literals = [&](uint32_t at, uint32_t len, bool mayNote) {
    if (len == 0u) return;
    const uint32_t c = opos & 15u, total = c + len;
    const uint32_t srcp = at - c;
    uint8_t *const dst = slot + (opos & ~15u);
    if (__builtin_expect(at + (uint32_t)H >= avail + 16u, 1))
    {
      u32x4 w = HS_SMERGE(oacc, ring_win(srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        if (!dry) st128(dst + k, w);
        k += 16u;
        if (k < total) w = ring_win(srcp + k);
      }
      oacc = w;
    }
    else if (mayNote && (MONO || len >= kNotedLiteralMin) && (pendBytes == 0u || pend2Bytes == 0u))
    {
      // what the accumulator holds goes out now; the stretch is noted up to the last 16-byte boundary of the stream it reaches; the
      // bytes behind that boundary come into the accumulator (through LDS: see below)
      if (c != 0u) store_bytes(dst, oacc, 0u, c);
      if constexpr (MONO)
      {
        // Chunk encoders: whatever has left the ring is noted, whole -- these are the last bytes of the stream (only finish_literals may
        // note), so the accumulator is not needed again and the lane neither loads nor waits.  Fetching a 112 .. 255 byte tail in the
        // lane's own code cost a drained memory pipeline per 16 bytes with one lane active: 12 us per block end, 83 % of the split
        // encode kernel of the 88 MB frame (rle64_3symlut_byte, 200 us).
        if (!dry)
        {
          if (pendBytes == 0u) { pendSrc = at; pendDst = opos; pendBytes = len; }
          else { pend2Src = at; pend2Dst = opos; pend2Bytes = len; }
        }
        tailNoted = true;
      }
      else
      {
        const uint32_t tail = (opos + len) & 15u, noted = len - tail;
        if (!dry)
        {
          if (pendBytes == 0u) { pendSrc = at; pendDst = opos; pendBytes = noted; }
          else { pend2Src = at; pend2Dst = opos; pend2Bytes = noted; }
        }
        lds_st128(accScratch + lane * 16u, tail != 0u ? global_window16(in, blockAt, U, at + noted) : zero4);
        oacc = lds_ld128(accScratch + lane * 16u);
      }
    }
    else
    {
      u32x4 w = HS_SMERGE(oacc, global_window16(in, blockAt, U, srcp), c);
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
    hflush();
  }

  // ---- per-row scalars for the lanes that serve a row (same scheme as the decoder's publish()) ----
  auto publish = [&](uint32_t v) { rinfo[(lane % (uint32_t)RPL) * (uint32_t)LPR + lane / (uint32_t)RPL] = v; };

  // ---- input top-up (4 lanes per row read 64 contiguous bytes; the loads fly during the step's scan) ----
  u32x4 pf[LPR];
  uint32_t pfAt[LPR];
  uint32_t wantReq = 0;
  [[maybe_unused]] uint64_t rowAt[LPR];                                 // MONO: input position of the LPR rows this lane serves
  if constexpr (MONO)
  {
#pragma unroll
    for (int q = 0; q < LPR; q++)
    {
      const int r = (int)((uint32_t)q * RPL + lane / LPR);
      const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)blockAt, r, 64), hi32 = (uint32_t)__shfl((int)(uint32_t)(blockAt >> 32), r, 64);
      rowAt[q] = ((uint64_t)hi32 << 32) | lo32;
    }
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
      const uint32_t r = (uint32_t)q * RPL + lane / LPR, c = lane % LPR;
      const uint32_t nreq = ri[q] & 15u, e = ri[q] & ~15u;
      const bool valid = c < nreq;
      const uint64_t g = (MONO ? rowAt[q] : (uint64_t)(wgFirst + r) * B) + e + c * 16u;
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

  // ---- one enumerated run [p, e) of the symbol {s0, s1}: decide, and if emitted write the packet ----
  auto handle_run = [&](uint32_t p, uint32_t e, uint32_t s0, uint32_t s1) {
    const uint32_t count = e - p;
    const uint32_t gap = p - lastRLE;

    if constexpr (TR::kShort)
    {
      // rleX_Xsl_short.h:152-357 (process_symbol of the Short family)
      const uint32_t range = gap + 2u;
      [[maybe_unused]] uint32_t m = (uint32_t)K;
#pragma unroll
      for (int k = K - 1; k >= 0; k--)
        if (lut0[k] == s0 && lut1[k] == s1) m = (uint32_t)k;
      const int32_t sc = TR::kAligned ? (int32_t)(count / SU) - (int32_t)(TR::SMINS / SU) + 2 : (int32_t)count - (int32_t)TR::SMINS + 2;
      const bool pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
      uint32_t pen = (K > 0 && m == (uint32_t)K) ? SU : 0u;
      if (!pack1)
      {
        pen += 2u;
        if (!(sc <= (int32_t)TR::SMAXTC && range <= TR::SMAXTR))
          pen += ((range <= 0xFFFFFu) ? (range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
      }
      if (!(count >= TR::SMINL || count >= TR::SMINS + pen))
        return;

      if constexpr (K > 0)
      {
        const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
        for (int k = K - 1; k >= 1; k--)
          if ((uint32_t)k <= limit) { lut0[k] = lut0[k - 1]; lut1[k] = lut1[k - 1]; }
        lut0[0] = s0; lut1[0] = s1;
        if (m >= mtfDepth && mtfDepth < (uint32_t)K) mtfDepth++;
      }

      const uint32_t mi = (K > 0) ? m << (TR::SCB + TR::SRBP) : 0u;
      if (pack1)
        hb(mi | ((uint32_t)(sc - 2) << TR::SRBP) | gap);
      else
      {
        const uint32_t scu = (uint32_t)sc;
        const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
        const uint32_t rx = (range <= TR::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
        hb(mi | (TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8));
        hb((scx << (TR::SRB - 8u)) | (rx >> 8));
        hb(rx);
        if (scx != scu) { if (scu <= 0xFFFFu) h16(scu); else h32(scu); }
        if (rx != range) { if (range <= 0xFFFFu) h16(range); else h32(range); }
      }
      if (K == 0 || m == (uint32_t)K) hsym(s0, s1);
    }
    else if constexpr (TR::kLut)
    {
      // rleX_Xsl.h:116-195 (process_symbol); SURVEY.md A.3
      constexpr uint32_t RB = TR::RB, MAXC = 127u, MAXR = (1u << RB) - 1u;
      const uint32_t range = gap + 2u;
      uint32_t m = (uint32_t)K;
#pragma unroll
      for (int k = K - 1; k >= 0; k--)
        if (lut0[k] == s0 && lut1[k] == s1) m = (uint32_t)k;
      const uint32_t c = TR::kAligned ? (count / SU - 3u / SU + 2u) : (count - 3u + 2u);
      // the penalty uses 0xFFFFF where the writer uses 0xFFFF (A.5 q3; rleX_Xsl.h:130 vs :195)
      uint32_t pen = (range <= 0xFFFFFu) ? (range <= MAXR ? 0u : 2u) : 4u;
      pen += (c <= 0xFFFFFu) ? (c <= MAXC ? 0u : 2u) : 4u;
      pen += (m == (uint32_t)K) ? 1u : 0u;
      if (!(count >= SU + 10u || count >= 3u + pen))
        return;

      // move to front (rleX_Xsl.h:134-188)
      const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
      for (int k = K - 1; k >= 1; k--)
        if ((uint32_t)k <= limit) { lut0[k] = lut0[k - 1]; lut1[k] = lut1[k - 1]; }
      lut0[0] = s0; lut1[0] = s1;
      if (m >= mtfDepth && mtfDepth < (uint32_t)K) mtfDepth++;

      const uint32_t c7 = (c <= MAXC) ? c : (c <= 0xFFFFu ? 1u : 0u);
      const uint32_t r7 = (range <= MAXR) ? range : (range <= 0xFFFFu ? 1u : 0u);
      h16((m << (K == 3 ? 14 : 13)) | (c7 << RB) | r7);
      if (m == (uint32_t)K) hsym(s0, s1);
      if (c != c7) { if (c <= 0xFFFFu) h16(c); else h32(c); }
      if (range != r7) { if (range <= 0xFFFFu) h16(range); else h32(range); }
    }
    else
    {
      // rleX_extreme_cpu_encode.h:165-313; thresholds SURVEY.md A.2
      const uint32_t range = gap + 1u;
      bool same = false;
      if constexpr (TR::kPacked) same = (s0 == la0 && s1 == la1);
      bool shortOk;
      if constexpr (!TR::kPacked) shortOk = range <= TR::MAXRANGE && count >= TR::SHORT;
      else shortOk = range <= TR::MAXRANGE && ((count >= TR::SHORT && same) || count >= TR::MEDIUM);
      const int k = shortOk ? 1 : (count >= TR::LONG ? 2 : 0);
      if (!k)
        return;
      if constexpr (TR::kPacked) { la0 = s0; la1 = s1; }

      const uint32_t c = TR::kAligned ? (count / SU - TR::SHORT / SU + 1u) : (count - TR::SHORT + 1u);
      if constexpr (!TR::kPacked)
      {
        hsym(s0, s1);
        if (c <= 255u) hb(c); else { hb(0); h32(c); }
      }
      else
      {
        const uint32_t sm = same ? 0x80u : 0u;
        if (c <= 127u) hb(c | sm); else { hb(sm); h32(c); }
        if (!same) hsym(s0, s1);
      }
      if constexpr (TR::kRange7)
      {
        if (k == 1) hb(range << 1); else h32((range << 1) | 1u);
      }
      else
      {
        if (k == 1) hb(range); else { hb(0); h32(range); }
      }
    }
    hflush();

    // ---- literals ----
    emit_literals(lastRLE, gap, false);
    lastRLE = e;

    if (e >= nTrue)
    {
      // end terminator (rleX_extreme_cpu_encode.h:373-609; rleX_Xsl_multibyte_encoder.h:329-370)
      if constexpr (TR::kShort) { hb(TR::SCINV << TR::SRBP); hb(TR::STB); hb(1); h16(0); h16(0); if (K == 0) hb(0); }   // one zero byte, whatever S
      else if constexpr (TR::kLut) { h16((1u << TR::RB) | 1u); h16(0); h16(0); }
      else
      {
        if constexpr (!TR::kPacked) { hsym(0, 0); hb(0); h32(0); } else { hb(0x80); h32(0); }
        if constexpr (TR::kRange7) h32(1); else { hb(0); h32(0); }
      }
      hflush();
      ended = true;
    }
  };

  // literal terminator carrying the bytes behind the last emitted run
  auto finish_literals = [&]() {
    const uint32_t kLit = n - lastRLE;
    if constexpr (TR::kShort) { hb(TR::SCINV << TR::SRBP); hb(TR::STB); hb(0); h16(0); h32(kLit + 2u); if (K == 0) hsym(0, 0); }   // a whole zero symbol here
    else if constexpr (TR::kLut) { h16(1u << TR::RB); h16(0); h32(kLit + 2u); }
    else
    {
      if constexpr (!TR::kPacked) { hsym(0, 0); hb(0); h32(0); } else { hb(0x80); h32(0); }
      if constexpr (TR::kRange7) h32(((kLit + 1u) << 1) | 1u); else { hb(0); h32(kLit + 1u); }
    }
    hflush();
    emit_literals(lastRLE, kLit, true);    // (the one place that notes: see above)
  };

  // a stretch of L set match bits that started at sStart has ended: if it is a run (L >= S), enumerate it
  auto close_stretch = [&](uint32_t L) -> uint32_t {                    // returns the run end (the new search position), or 0
    if (L < SU)
      return 0u;
    const uint32_t whole = sStart + SU + SU * (L / SU);
    uint32_t e = whole;
    if constexpr (!TR::kAligned)
      if (whole + SU <= nTrue) e = sStart + SU + L;
    handle_run(sStart, e, sy0, sy1);
    return e;
  };

  // ---- main loop ----
  issue();
  land();
  wave_sync();

  uint32_t stepsLeft = MONO ? ((B != 0u) ? (monoSteps & 0xFFFFu) : monoSteps) : 2u * (B / (uint32_t)Q) + 64u;  // bounded: every step scans a window or lands input

  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u) break;
    issue();

    if (!finished)
    {
      // window [cb, cb + W): position j needs the bytes up to j + S (or the end of the input)
      const bool lastStep = avail >= n;
      uint32_t W = 0;
      if (cb < n)
      {
        if (lastStep) W = umin(64u, n - cb);
        else if (avail > cb + SU) W = umin(64u, ((avail - SU - cb) >> 4) << 4);
      }

      if (W != 0u)
      {
        uint64_t m = 0;
        u32x4 x = lds_ld128(hist + (hbase ^ (cb & HM)));
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++)
        {
          if (j * 16u < W)
          {
            const u32x4 nx = lds_ld128(hist + (hbase ^ ((cb + j * 16u + 16u) & HM)));
            const u32x4 y = funnel16(x, nx, SU);                         // the bytes S further on
            const uint32_t zm16 = zero_mask16(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
            m |= (uint64_t)zm16 << (16u * j);
            x = nx;
          }
        }
        // position j matches only if j + S < n (bytes at or beyond n never match) and lies inside the window
        const uint32_t validBits = (n > cb + SU) ? umin(W, n - SU - cb) : 0u;
        m &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
        // positions below the search position are ignored
        if (from > cb) m &= (from - cb >= 64u) ? 0ull : ~((1ull << (from - cb)) - 1ull);

        uint32_t pos = 0;
        for (;;)
        {
          if (!open)
          {
            const uint64_t rest = (pos < 64u) ? (m >> pos) : 0ull;
            if (rest == 0ull) break;
            pos += (uint32_t)__builtin_ctzll(rest);
            sStart = cb + pos;
            const u32x4 sv = ring_win(sStart);
            sy0 = (S >= 4) ? sv.x : (sv.x & ((1u << (8 * (S & 3))) - 1u));
            sy1 = (S == 8) ? sv.y : (S == 6 ? (sv.y & 0xFFFFu) : 0u);
            open = true;
          }
          // the stretch ends at the first clear bit at or behind pos
          const uint64_t inv = (pos < 64u) ? (~m >> pos) : 0ull;
          const uint32_t z = (inv != 0ull) ? pos + (uint32_t)__builtin_ctzll(inv) : 64u;
          if (z >= W && !(lastStep && cb + W >= n))
            break;                                                       // still open at the window end: goes on in the next window
          const uint32_t zc = umin(z, W);
          open = false;
          pos = zc;
          const uint32_t e = close_stretch(cb + zc - sStart);
          if (e != 0u)
          {
            from = e;
            if (e > cb + zc)
            {
              const uint32_t cut = e - cb;
              if (cut >= 64u) { m = 0; pos = 64u; }
              else { m &= ~((1ull << cut) - 1ull); pos = cut; }
            }
          }
          if (pos >= W) break;
        }
        cb += W;
      }

      if (cb >= n && avail >= n)
      {
        // end of input: the literal terminator unless the stream ended with a run
        if (!ended && n == nTrue) { finish_literals(); ended = true; }
        // the last partial chunk, then the stream size (header field compressedLength and the size table)
        if ((opos & 15u) != 0u && !dry && !(MONO && tailNoted))
          st128(slot + (opos & ~15u), oacc);
        if constexpr (!MONO) st32(slot + 4, opos);
        if (!dry) sizes[b] = opos;
        if constexpr (MONO && TR::kMtf)
        {
#pragma unroll
          for (int k = 0; k < K; k++) monoListOut[8ull * b + k] = (uint64_t)lut0[k] | ((uint64_t)lut1[k] << 32);
          monoListOut[8ull * b + 7] = mtfDepth;
        }
        finished = true;
      }
    }

    wave_sync();
    land();
    wave_sync();
  }
  coop_flush(true);
  if constexpr (MONO && Traits<FAM, S, AL>::kMtf)
  {
    if (grp == 0u) break;
    // the lists in front of this wave's chunks from what the chunks left behind (all of a block's chunks are here: no other wave writes
    // what is read below); a chunk whose list comes out different is switched on for the next round
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wave_sync();
    bool changed = false;
    if (mine)
    {
      MonoListAcc a; a.n = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) a.e[k] = 0;
      for (uint32_t i = b; i > blockFirstChunk && a.n < (uint32_t)K; i--)
      {
        uint64_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = ld_fresh64(monoListOut + 8ull * (i - 1u) + k);
        a.add(t, (uint32_t)K);
      }
      for (uint32_t k = 0; k < (uint32_t)K && a.n < (uint32_t)K; k++) a.add_one(mono_default_entry(k, (uint32_t)S), (uint32_t)K);
      uint64_t *const g = monoSyms + 8ull * b;
#pragma unroll
      for (int k = 0; k < 7; k++)
        if (k < K && ld_fresh64(g + k) != a.e[k]) { changed = true; g[k] = a.e[k]; }
      g[7] = changed ? 1ull : 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (__ballot(changed) == 0ull) break;
    // a block's chunks settle one per round at worst and a wave holds 64 chunks at most: round 64 cannot find a change.  Should it ever,
    // the launch dies (the host sees a launch failure) rather than leave a stream behind that was encoded from a wrong list
    if (listRound >= 64u) __builtin_trap();
    wave_sync();
  }
  else
    break;
  }
}

} // namespace hsrle
