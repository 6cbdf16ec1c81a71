// hsrle_encode8.hip.h -- the 8 bit multi-symbol encoders (rle8_multi, rle8_packed_multi, rle8_{3,7}symlut) with the block's
// input staged through LDS and the output assembled in registers.
//
// Replaces: src/rle8_extreme_cpu.h:86-344 (wrapper, scalar tail, final block), :936-1099 (canonical AVX2 body),
//           src/rleX_Xsl.h:114-264 (process_symbol), :269-346, :421-485 (TYPE_SIZE 8 instantiation).
//
// Same execution model as everything else here: one lane = one block = one reference stream, 64 blocks per wavefront, the
// emit decisions run as the sequential state machine they are.  Data path:
//   HBM --(top-up: 4 adjacent lanes read 64 contiguous, aligned input bytes of ONE block per trip)--> LDS history ring [64][256]
//   ring --(match bits: when chunks have landed their lane computes E[i] = (d[i] == d[i - 1]) for them -- SWAR compare + v_dot4
//           as the movemask, the GPU form of the reference's cmpeq + movemask, rle8_extreme_cpu.h:952-1084 -- into a bit ring
//           in LDS, 256 positions per lane)-->
//   trips: the loop is RUN-synchronous, not window-synchronous: in every trip EVERY lane finds its own next run (two 64-bit
//           windows of the bit ring + ctz: start, end) and judges it, so a wave makes max-over-lanes(runs per block) trips
//           (~75 on the run-distributed buffer) instead of the sum over 64-byte windows of max-over-lanes(runs per window) (~290):
//           the lanes drift apart in their blocks by bytes, never by runs, and every row of the ring tops up at its own pace.
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

#ifdef HSRLE_E8_STATS
// diagnostic build only (tools/e8_stats.py): what the waves of the last launches executed
__device__ unsigned long long g_e8stats[32];
#define E8S(x) x
#else
#define E8S(x)
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
__global__ __launch_bounds__(128) void k_encode8_blocks(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                       uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes,
                                                       const uint64_t *__restrict__ monoStarts, const uint64_t *__restrict__ monoSyms,
                                                       const uint64_t *__restrict__ monoSlotOff, uint32_t monoSteps, uint64_t *__restrict__ monoListOut, uint32_t monoDry,
                                                       const uint32_t *__restrict__ ringSel,
                                                       uint64_t *__restrict__ fuseOffsets, uint8_t *__restrict__ fusePayload, unsigned long long *__restrict__ fuseTiles)
{
  if (ringSel != nullptr && ringSel[0] != (uint32_t)RING) return;
  // fuseOffsets != nullptr (block containers): the workgroup PLACES its 64 streams itself when they are done -- their offsets come from a
  // decoupled look-back over one word per workgroup (fuseTiles, zeroed by the host), the copy from the staging slots to the payload is done
  // by both waves together.  Replaces the size scan and k_compact's pass over the payload (1.85 ms of 7.0 for 8 GiB): the copies of the
  // workgroups that are done run under the encoding of the others.  The look-back wants the workgroups in dispatch order, so the fused form
  // gives up the XCD-aware tile order.
  const bool fuse = !MONO && fuseOffsets != nullptr;
  using TR = Traits<FAM, 1, 0>;
  // Codecs with a move-to-front list: the list in front of a chunk is NOT known from the boundary run; the host hands every chunk a list
  // (monoSyms[8 * c + k]: entry k; [8 * c + 7]: encode this chunk?), gets the list behind it back (monoListOut[8 * c + k]; [.. + 7]: mtfDepth) and repeats the chunks whose
  // incoming list was not what the chunk in front left behind (hsrle_capi.hip: mono_encode_dev).  monoDry: no stores, only the list.
  [[maybe_unused]] const bool dry = MONO && monoDry != 0u;
  constexpr int Q = 128;                     // input bytes per row and loader round (at most)
  constexpr int H = RING;                    // history ring per lane (power of two)
  constexpr int LPR = Q / 16, RPL = 64 / LPR;
  constexpr uint32_t HM = (uint32_t)H - 1u;
  constexpr int K = TR::K;

  // ring byte x of row r lives at hist[r * RS + (x & HM)]; the row's first chunk is mirrored behind its end, so that the 20-byte window of
  // a literal copy (five dword reads at ANY dword of the row: ring_win) never wraps
  constexpr uint32_t RS = (uint32_t)H + 16u;
  __shared__ __attribute__((aligned(16))) uint8_t hist[64 * RS];
  // match bits of the ring's positions: bit (i & 31) of ebits[((i >> 5) % EW) * 64 + lane] = (d[i] == d[i - 1]); dword-transposed so
  // that every lane stays in its own LDS bank whichever dword of its row it reads
  constexpr uint32_t EW = (uint32_t)H / 32u;
  __shared__ __attribute__((aligned(16))) uint32_t ebits[EW * 64u];
  // merge masks from a 16-entry table (as in k_decode_blocks: one ds_read_b128 instead of ~9 VALU; +1 % encode throughput)
  __shared__ __attribute__((aligned(16))) uint8_t mlut[16 * 16];
  // the two waves of the workgroup talk through two words per row, both only ever grow: the encoder publishes the first ring position it
  // still wants to read (keepPub), the loader what has landed and has its match bits (availPub)
  __shared__ uint32_t keepPub[64], availPub[64];
  __shared__ uint32_t abortFlag;
  // (relaxed atomic accesses: a plain LDS read could be hoisted out of the polling loops, a volatile one is compiled into a FLAT load that
  //  waits on vmcnt -- i.e. for every store the wave has in flight)
#define HS_LDS_GET(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define HS_LDS_PUT(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  if (threadIdx.x < 64u) { keepPub[threadIdx.x] = 0u; availPub[threadIdx.x] = 0u; }
  if (threadIdx.x == 0u) abortFlag = 0u;
  // fused: the workgroups take their tiles in the order they START (a ticket counter behind the look-back words): the eight XCDs work
  // through the grid at their own pace, so "workgroup i - 1" may not even be dispatched while workgroup i waits for its sizes
  __shared__ uint32_t tileTicket;
  if (fuse && threadIdx.x == 0u) tileTicket = (uint32_t)atomicAdd(&fuseTiles[gridDim.x], 1ull);
  if (threadIdx.x < 16u)
  {
    const uint32_t c = threadIdx.x;
    const uint64_t part = ~(~0ull << (8u * (c & 7u)));
    const bool hiHalf = c >= 8u;
    const uint32_t p0 = (uint32_t)part, p1 = (uint32_t)(part >> 32);
    lds_st128(mlut + c * 16u, u32x4{ hiHalf ? ~0u : p0, hiHalf ? ~0u : p1, hiHalf ? p0 : 0u, hiHalf ? p1 : 0u });
  }
  __syncthreads();                                                      // the only barrier: from here on the waves run at their own pace
#define HS_EMERGE(keep, fresh, c) merge_low_m(keep, fresh, lds_ld128(mlut + ((c) << 4)))

  const uint32_t lane = threadIdx.x & 63u;
  const bool isLoader = threadIdx.x >= 64u;
  const uint32_t tileNo = fuse ? tileTicket : xcd_tile(blockIdx.x, gridDim.x);               // XCD-aware tile order (hsrle_common.hip.h)
  const uint32_t wgFirst = tileNo * 64u;
  const uint32_t b = wgFirst + lane;
  bool active = b < nBlocks;
  if constexpr (MONO && Traits<FAM, 1, 0>::kMtf) { if (active) active = monoSyms[8ull * b + 7] != 0ull; }   // the host's repair rounds switch most chunks off

  uint8_t *const row = hist + lane * RS;

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
    }
    else
    {
      const uint64_t start = (uint64_t)b * B;
      n = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
    }
  }
  const uint32_t nTrue = MONO ? nTrueV : n;
  uint8_t *const slot = MONO ? slots + (active ? monoSlotOff[b] : 0ull) : slots + (uint64_t)b * slotStride;

  constexpr uint32_t kReach = (uint32_t)H - 16u;                       // the loader keeps avail - keep <= kReach: nothing at or above `keep` is overwritten
  constexpr uint32_t kNeed = ((uint32_t)H >= 256u) ? 144u : (uint32_t)H / 2u + 16u;   // what the scan wants in front of it (start within 64, end within 64 more, one chunk)

  if (isLoader)
  {
    // ================= loader wave: HBM -> ring, match bits =================
    // 8 adjacent lanes read 128 contiguous, aligned input bytes of ONE row per round; lane r owns row r's counters.  The loader has no stores
    // and nothing but its loads in flight, and it runs AHEAD of the encoder wave by up to kReach bytes per row, so neither wave ever waits
    // for the other's memory latency (one wave doing both waited for its loads -- and, vmcnt being one counter, for its stores -- every trip).
    u32x4 pf[LPR];
    uint32_t pfAt[LPR];
    bool pfValid[LPR];
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
      rowLds[q] = r * RS;
    }
    uint32_t avail = 0;      // input bytes [.., avail) are (or were) in the ring; the ring holds [avail - H, avail)
    uint32_t mk = 0;         // match bits exist for the positions below mk (multiple of 16)
    uint32_t pd = 0;         // the last dword of the chunk below mk (its top byte is the predecessor of position mk)
    uint32_t idle = 0;
    __builtin_amdgcn_s_setprio(3);                                      // the encoder wave waits for this one, never the other way round
    while (__ballot(avail < n) != 0ull)
    {
      // chunks this round: as many as are left and as fit in front of what the encoder still wants to read
      const uint32_t keep = HS_LDS_GET(keepPub[lane]);
      const uint32_t left = (n > avail) ? (n - avail + 15u) >> 4 : 0u;
      const uint32_t behind = avail - keep;
      const uint32_t wantReq = umin(umin((uint32_t)LPR, left), (behind < kReach) ? (kReach - behind) >> 4 : 0u);
      // The ring is a short time buffer (kReach bytes = what the encoder consumes in about four trips), so a round starts as soon as any row
      // has room: waiting for fuller rounds starved the encoder (26 instead of 7 trips per lane without a run).
      if (__ballot(wantReq != 0u) == 0ull)
      {
        // every ring of the wave is full: wait for the encoder (bounded: a stuck encoder raises abortFlag, and so does this loop)
        if (HS_LDS_GET(abortFlag) != 0u || ++idle > (1u << 22)) { HS_LDS_PUT(abortFlag, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      // the lanes that serve a row take its request from the row's owner (ds_bpermute: no LDS memory)
      const uint32_t req = wantReq != 0u ? (avail | wantReq) : 0u;     // avail is a multiple of 16 while chunks are left
      uint32_t ri[LPR];
#pragma unroll
      for (int q = 0; q < LPR; q++) ri[q] = (uint32_t)__shfl((int)req, (int)((uint32_t)q * RPL + lane / LPR), 64);
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
#ifdef HSRLE_X_NTLOAD
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(g));
#else
            v = ld128(g);
#endif
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
        pfValid[q] = valid;
        pfAt[q] = rowLds[q] + ((e + c * 16u) & HM);
      }
#pragma unroll
      for (int q = 0; q < LPR; q++)
        if (pfValid[q])
        {
          lds_st128(hist + pfAt[q], pf[q]);
          if ((pfAt[q] - rowLds[q]) == 0u) lds_st128(hist + pfAt[q] + (uint32_t)H, pf[q]);   // the mirror of the row's first chunk
        }
      avail = umin(avail + (wantReq << 4), n);
      wave_sync();
      // match bits of the chunks that are new in this lane's row: E[i] = (d[i] == d[i - 1]) -- x ^ (x shifted by one byte) per dword, the
      // exact SWAR zero-byte test, and v_dot4_u32_u8 with the weights 1 2 4 8 / 16 .. 128 as the movemask: 5 VALU per dword.  (Bits of
      // positions at or beyond n are never looked at: the scan stops at `avail` <= n.)
#pragma unroll
      for (int q = 0; q < LPR; q++)
      {
        if (mk < avail)
        {
          const u32x4 x = lds_ld128(row + (mk & HM));
          const uint32_t t0 = x.x ^ alignbyte(x.x, pd, 3), t1 = x.y ^ alignbyte(x.y, x.x, 3), t2 = x.z ^ alignbyte(x.z, x.y, 3), t3 = x.w ^ alignbyte(x.w, x.z, 3);
          const uint32_t n0 = (((t0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t0) & 0x80808080u, n1 = (((t1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t1) & 0x80808080u;
          const uint32_t n2 = (((t2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t2) & 0x80808080u, n3 = (((t3 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t3) & 0x80808080u;
          // 0x80 per differing byte -> 128 * (8 bit mask) per dword pair
          const uint32_t a01 = __builtin_amdgcn_udot4(n1, 0x80402010u, __builtin_amdgcn_udot4(n0, 0x08040201u, 0u, false), false);
          const uint32_t a23 = __builtin_amdgcn_udot4(n3, 0x80402010u, __builtin_amdgcn_udot4(n2, 0x08040201u, 0u, false), false);
          uint32_t e16 = ~(((a23 << 8) | a01) >> 7);
          if (mk == 0u) e16 &= ~1u;                                     // position 0 has no predecessor
          reinterpret_cast<uint16_t *>(ebits)[(((mk >> 5) & (EW - 1u)) * 64u + lane) * 2u + ((mk >> 4) & 1u)] = (uint16_t)e16;
          pd = x.w;
          mk += 16u;
        }
      }
      wave_sync();
      HS_LDS_PUT(availPub[lane], avail);                                        // behind the data and the bits (LDS accesses of a wave execute in order)
    }
    if (!fuse) return;
  }
  else
  {
  // ================= encoder wave =================
  // ---- per-lane encoder state ----
  uint32_t avail = 0;        // what the loader has published for this row: input bytes [.., avail) are in the ring and have their match bits
  uint32_t keepFrom = 0;     // what this lane has published: the loader overwrites nothing at or above it
  uint32_t cur = 0;          // scan position: every run that starts below cur has been judged (inRun: the run from runStart reaches cur)
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
  E8S(uint32_t sTrips = 0; uint32_t sRunLanes = 0; uint32_t sRunTrips = 0; uint32_t sEmitLanes = 0; uint32_t sLitWave = 0; uint32_t sLitLanes = 0; uint32_t sGlobalTrips = 0;
      uint32_t sMaskPasses = 0; uint32_t sMaskLanes = 0; uint32_t sLoadLanes = 0; uint32_t sStarve = 0; uint32_t sHdrStore = 0; uint32_t tLit = 0; uint32_t tGlob = 0; uint32_t tEmit = 0; uint32_t tHdrSt = 0;
      uint32_t sFinTrips = 0; uint32_t sActive = 0;)

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

  // append the low nb (<= 12) bytes of hv
  auto append = [&](u32x4 hv, uint32_t nb) {
    const uint32_t c = opos & 15u;
    const u32x4 lowp = (c == 0u) ? hv : funnel16(zero4, hv, 16u - c);   // hv << c bytes
    const u32x4 w = HS_EMERGE(oacc, lowp, c);
    if (c + nb >= 16u)
    {
      E8S(tHdrSt++;)
#ifndef HSRLE_X_NOSTORE
      if (!dry) st128(slot + (opos & ~15u), w);
#endif
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

  // 16 input bytes at block position p (p may reach below 0 or beyond the input: those bytes are never used): five dwords from the dword
  // that holds p (ds_read2_b32 / ds_read_b32: full speed at any dword, unlike the wider reads -- tools/ubench/lds_align.hip) and four
  // v_alignbyte; with two aligned 16-byte reads the dword selection cost 11 v_cndmask per window
  auto ring_win = [&](uint32_t p) -> u32x4 {
    const uint32_t *const w = reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(row + (p & HM & ~3u), 4));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4], sh = p & 3u;
    return u32x4{ alignbyte(d1, d0, sh), alignbyte(d2, d1, sh), alignbyte(d3, d2, sh), alignbyte(d4, d3, sh) };
  };
  // literal bytes [from, from + len) of the block: from the ring while they are still there.  Literals that have left the ring (a long
  // stretch of runs too short to be stored) are read from global memory by a function that is kept out of line: with the two
  // sources selected per chunk inside one loop the rare path cost the common one 10 % of the kernel.
  auto emit_literals = [&](uint32_t from, uint32_t len, bool mayNote) {
    if (len == 0u) return;
    const uint32_t c = opos & 15u, total = c + len;
    const uint32_t srcp = from - c;
    uint8_t *const dst = slot + (opos & ~15u);
    if (__builtin_expect(from >= keepFrom, 1))
    {
      u32x4 w = HS_EMERGE(oacc, ring_win(srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        E8S(tLit++;)
#if defined(HSRLE_X_SAMELINE)
        st128(slot, w);
#elif defined(HSRLE_X_DENSE)
        st128(slots + (uint64_t)wgFirst * slotStride + lane * 16u, w);
#elif defined(HSRLE_X_NTSTORE)
        if (!dry) __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(dst + k));
#elif !defined(HSRLE_X_NOSTORE)
        if (!dry) st128(dst + k, w);
#else
        if (w.x == 0x12345678u && w.y == 0x9ABCDEF0u) st128(dst + k, w);
#endif
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
      oacc = tail != 0u ? global_window16(in, blockAt, U, from + noted) : zero4;
      __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): see below
    }
    else
    {
      E8S(tGlob = 1;)
      u32x4 w = HS_EMERGE(oacc, global_window16(in, blockAt, U, srcp), c);
      uint32_t k = 0;
      while (k + 16u <= total)
      {
        if (!dry) st128(dst + k, w);
        k += 16u;
        if (k < total) w = global_window16(in, blockAt, U, srcp + k);
      }
      // the loads are waited for HERE: an accumulator that may still be the target of a pending vector-memory load where the paths join
      // makes the compiler wait for ALL outstanding loads -- the input prefetch included -- in front of every store of the common path
      // (-10 % encode; round 2 handed the value back through 1 KB of LDS instead, which the bit ring needs now)
      __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
      oacc = w;
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
  if (active && !MONO)
  {
    h32(n);
    h32(0);
    if constexpr (!TR::kLut && !TR::kShort) hb(0); // mode = multi
    hflush();
  }

  // E[c .. c + 64) (the caller knows how many of them exist)
  auto ewin = [&](uint32_t c) -> uint64_t {
    const uint32_t k = c >> 5, sh = c & 31u;
    const uint32_t d0 = ebits[((k) & (EW - 1u)) * 64u + lane], d1 = ebits[((k + 1u) & (EW - 1u)) * 64u + lane], d2 = ebits[((k + 2u) & (EW - 1u)) * 64u + lane];
    return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
  };

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

#ifdef HSRLE_X_NOEMIT
    k = 0;
#endif
    if (!k)
      return;
    E8S(tEmit = 1;)

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
    if constexpr (TR::kShort) { hb(TR::SCINV << TR::SRBP); hb(TR::STB); hb(0); h16(0); h32(kLit + 2u); if (K == 0) hb(0); }   // :503-523
    else if constexpr (TR::kLut) { h16(1u << TR::RB); h16(0); h32(kLit + 2u); }
    else if constexpr (TR::kPacked) { hb(0x80); h32(0); h32(((kLit + 1u) << 1) | 1u); }
    else { hb(0); hb(0); h32(0); hb(0); h32(kLit + 1u); }
    hflush();
    emit_literals(lastRLE, kLit, true);    // (the one place that notes: see above)
  };

  // ---- main loop ----
  // bounded: a trip judges a run (>= 2 bytes), or moves the cursor over >= 16 positions, or finds that nothing new has landed (the
  // loader is bounded too); a wave that gives up raises abortFlag so that its loader does not wait for it
  uint32_t stepsLeft = (MONO ? 32u * monoSteps : B + 64u) + (1u << 22);

  while (__ballot(!finished) != 0ull)
  {
    if (stepsLeft-- == 0u || HS_LDS_GET(abortFlag) != 0u) { HS_LDS_PUT(abortFlag, 1u); break; }
    avail = HS_LDS_GET(availPub[lane]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");            // the ring and the bit ring are read behind this
    const uint32_t curIn = cur;

    // ---------------- one run per lane and trip ----------------
    if (!finished)
    {
      const bool allIn = avail >= n;
#ifdef HSRLE_X_NOSCAN
      cur = avail; inRun = false;
#endif
      const uint32_t known = avail;                                     // match bits exist below `known` (avail <= n); allIn: position n does not match
      if (!inRun)
      {
        // the next run starts at the first i >= cur with E[i + 1]
        if (cur + 1u < known)
        {
          const uint64_t w = ewin(cur + 1u);
          const uint32_t lim = umin(known - (cur + 1u), 64u);
          const uint32_t j = (w != 0ull) ? (uint32_t)__builtin_ctzll(w) : 64u;
          if (j < lim)
          {
            runStart = cur + j;
            sym = row[runStart & HM];
            inRun = true;
            cur = runStart + 1u;
          }
          else
            cur += lim;
        }
        else if (allIn)
          cur = n;
      }
      bool haveRun = false;
      uint32_t runEnd = 0;
      if (inRun)
      {
        // ... and ends at the first i >= cur without E[i] (cur > runStart; E[runStart + 1] is set)
        const uint64_t w = ~ewin(cur);
        const uint32_t lim = umin(known - cur, 64u);
        const uint32_t j = (w != 0ull) ? (uint32_t)__builtin_ctzll(w) : 64u;
        if (j < lim) { haveRun = true; runEnd = cur + j; }
        else if (allIn && cur + lim >= n) { haveRun = true; runEnd = n; }
        else cur += lim;
      }
      if (haveRun)
      {
        handle_run(runStart, runEnd);
        inRun = false;
        cur = runEnd;
      }
      E8S(if (!haveRun && !(allIn && !inRun && cur >= n)) sStarve += (avail - cur < 64u && !allIn) ? 1u : 0u;)

      if (allIn && !inRun && cur >= n)
      {
        // end of input: the literal terminator unless the stream ended with a run's packet
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

#ifdef HSRLE_E8_STATS
    {
      sTrips++;
      sActive += (uint32_t)__builtin_popcountll(__ballot(!finished));
      const uint64_t br = __ballot(tEmit != 0u || tLit != 0u || tHdrSt != 0u);
      (void)br;
      sEmitLanes += (uint32_t)__builtin_popcountll(__ballot(tEmit != 0u));
      sGlobalTrips += (__ballot(tGlob != 0u) != 0ull) ? 1u : 0u;
      uint32_t m = tLit;
      for (int dd = 32; dd >= 1; dd >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, dd, 64); m = m > o ? m : o; }
      sLitWave += m;
      uint32_t sum = tLit;
      for (int dd = 32; dd >= 1; dd >>= 1) sum += (uint32_t)__shfl_xor((int)sum, dd, 64);
      sLitLanes += sum;
      uint32_t hs = tHdrSt;
      for (int dd = 32; dd >= 1; dd >>= 1) hs += (uint32_t)__shfl_xor((int)hs, dd, 64);
      sHdrStore += hs;
      tLit = 0; tGlob = 0; tEmit = 0; tHdrSt = 0;
    }
#endif
    // what the loader may overwrite: nothing from lastRLE on (the literals in front of the next run are copied from the ring when the
    // run ends) -- unless the scan needs the room: kNeed positions in front of it come first, and a literal stretch that is given up
    // here is fetched from global memory by its lane (emit_literals looks at keepFrom)
    {
      const uint32_t want = inRun ? kNeed / 2u : kNeed;                 // in a run: only its end is looked for
      const uint32_t floorKeep = (cur + want > kReach) ? cur + want - kReach : 0u;
      const uint32_t k = finished ? 0xFFFFFFFFu : (lastRLE > floorKeep ? lastRLE : floorKeep);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // this trip's ring reads are done
      if (k > keepFrom) { keepFrom = k; HS_LDS_PUT(keepPub[lane], k); }          // (never back: what was given up may be gone)
    }
    // nothing moved in any lane: the loader's data is still on its way
    if (__ballot(cur != curIn || finished) == 0ull) __builtin_amdgcn_s_sleep(2);
  }
#ifdef HSRLE_E8_STATS
  {
    uint32_t st = sStarve;
    for (int dd = 32; dd >= 1; dd >>= 1) st += (uint32_t)__shfl_xor((int)st, dd, 64);
    if (lane == 0u)
    {
      atomicAdd(&g_e8stats[0], 1ull); atomicAdd(&g_e8stats[1], (unsigned long long)sTrips); atomicAdd(&g_e8stats[2], (unsigned long long)sActive);
      atomicAdd(&g_e8stats[3], (unsigned long long)sEmitLanes); atomicAdd(&g_e8stats[4], (unsigned long long)sGlobalTrips); atomicAdd(&g_e8stats[5], (unsigned long long)sLitWave);
      atomicAdd(&g_e8stats[6], (unsigned long long)sLitLanes); atomicAdd(&g_e8stats[7], (unsigned long long)sHdrStore); atomicAdd(&g_e8stats[8], (unsigned long long)sLoadLanes);
      atomicAdd(&g_e8stats[9], (unsigned long long)st); atomicAdd(&g_e8stats[10], (unsigned long long)sMaskPasses); atomicAdd(&g_e8stats[11], (unsigned long long)sMaskLanes);
    }
  }
#endif
  coop_flush(true);
  if (!fuse) return;

  // ---- placement, part 1 (encoder wave): this workgroup's offsets by decoupled look-back ----
  {
    constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_PREFIX = 2ull << 62, VALUE = (1ull << 62) - 1ull;
    const uint64_t sz = active ? (uint64_t)opos : 0ull;
    uint64_t incl = sz;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1)
    {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, dd, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), dd, 64);
      if (lane >= (uint32_t)dd) incl += ((uint64_t)hi << 32) | lo;
    }
    const uint64_t total = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)incl, 63, 64);
    // (the look-back words carry their value with them, so relaxed device-scope accesses are all it takes: a release / acquire pair at
    //  device scope writes back / invalidates the whole L2 of the XCD on this chip -- 19.8 instead of 6 ms per 8 GiB)
    const uint32_t tile = tileNo;
    if (lane == 0u && tile != 0u) __hip_atomic_store(&fuseTiles[tile], FLAG_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    int64_t lookAt = (int64_t)tile - 1;
    uint32_t spins = 0;
    while (lookAt >= 0)
    {
      const int64_t idx = lookAt - (int64_t)lane;
      const unsigned long long v = (idx >= 0) ? __hip_atomic_load(&fuseTiles[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLAG_PREFIX;
      const uint32_t flag = (uint32_t)(v >> 62);
      const uint64_t pm = __ballot(flag == 2u);
      const uint32_t firstPrefix = pm != 0ull ? (uint32_t)__builtin_ctzll(pm) : 64u;
      const uint64_t upTo = firstPrefix < 63u ? ((2ull << firstPrefix) - 1ull) : ~0ull;
      if ((__ballot(flag == 0u) & upTo) != 0ull)
      {
        if (++spins > (1u << 24)) break;                               // (cannot happen: every earlier workgroup is resident or done)
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      uint64_t part = (lane <= firstPrefix) ? (uint64_t)(v & VALUE) : 0ull;
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1)
      {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)part, dd, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(part >> 32), dd, 64);
        part += ((uint64_t)hi << 32) | lo;
      }
      excl += part;
      if (firstPrefix < 64u) break;
      lookAt -= 64;
    }
    if (lane == 0u) __hip_atomic_store(&fuseTiles[tile], FLAG_PREFIX | ((excl + total) & VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t at = excl + incl - sz;
    if (active) fuseOffsets[b] = at;
    if (active && b + 1u == nBlocks) fuseOffsets[nBlocks] = at + sz;
    // the rows' places for both waves (the ring is free now)
    reinterpret_cast<uint64_t *>(hist)[lane] = at;
    reinterpret_cast<uint32_t *>(hist + 512)[lane] = (uint32_t)sz;
  }
  }
  __syncthreads();

  // ---- placement, part 2 (both waves): slot streams -> payload, destination-aligned 16-byte stores; four rows per wave and pass, two
  //      chunks per row and lane in flight ----
  {
    const uint32_t w = threadIdx.x >> 6;
    for (uint32_t r0 = w * 4u; r0 < 64u; r0 += 8u)
    {
      const uint8_t *src[4];
      uint8_t *dst[4];
      uint32_t head[4], body[4], size[4];
      uint32_t most = 0;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const uint32_t r = r0 + (uint32_t)j;
        const uint64_t at = reinterpret_cast<const uint64_t *>(hist)[r];
        size[j] = reinterpret_cast<const uint32_t *>(hist + 512)[r];
        src[j] = slots + (uint64_t)(wgFirst + r) * slotStride;
        dst[j] = fusePayload + at;
        head[j] = umin((uint32_t)((16u - ((uint32_t)(uintptr_t)dst[j] & 15u)) & 15u), size[j]);
        body[j] = (size[j] - head[j]) & ~15u;
        most = body[j] > most ? body[j] : most;
      }
      most = (uint32_t)__builtin_amdgcn_readfirstlane((int)most);       // (the same in every lane: the rows' sizes come from LDS)
      for (uint32_t k0 = 0; k0 < most; k0 += 2048u)
      {
        u32x4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int u = 0; u < 2; u++)
          {
            const uint32_t k = k0 + (uint32_t)u * 1024u + lane * 16u;
            if (k < body[j]) v[j][u] = ld128(src[j] + head[j] + k);
          }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int u = 0; u < 2; u++)
          {
            const uint32_t k = k0 + (uint32_t)u * 1024u + lane * 16u;
            if (k < body[j]) st128(dst[j] + head[j] + k, v[j][u]);
          }
      }
      // the bytes in front of the first and behind the last whole chunk: lanes 0..15 / 16..31 of row j's quarter... one row per 16 lanes
      {
        const uint32_t j = lane >> 4, x = lane & 15u;
        const uint8_t *sp = src[0]; uint8_t *dp = dst[0]; uint32_t hd = head[0], bd = body[0], sz = size[0];
        if (j == 1u) { sp = src[1]; dp = dst[1]; hd = head[1]; bd = body[1]; sz = size[1]; }
        if (j == 2u) { sp = src[2]; dp = dst[2]; hd = head[2]; bd = body[2]; sz = size[2]; }
        if (j == 3u) { sp = src[3]; dp = dst[3]; hd = head[3]; bd = body[3]; sz = size[3]; }
        if (x < hd) dp[x] = sp[x];
        const uint32_t tail = sz - hd - bd;
        if (x < tail) dp[hd + bd + x] = sp[hd + bd + x];
      }
    }
  }
}

} // namespace hsrle
