// hsrle_encode8r.hip.h -- the 8 bit multi-symbol block encoders (rle8_multi, rle8_packed_multi, rle8_{3,7}symlut and the Short family:
// rle8_multi_short, rle8_{1,3,7}symlut_short) as a RUN LIST encoder: the parallel parts of a block's encode are done by the whole wave,
// only the emit decisions by one lane per block.
//
// Replaces: src/rle8_extreme_cpu.h:86-344 (wrapper, scalar tail, final block), :936-1099 (canonical AVX2 body); src/rleX_Xsl.h:114-264
//           (process_symbol of the LUT codecs), :269-346, :421-485 (TYPE_SIZE 8); src/rleX_Xsl_short.h:152-357 (process_symbol of the Short
//           family), :470-523 (its terminators) -- the same streams as k_encode8_blocks (hsrle_encode8.hip.h), whose handle_run is restated
//           here as a decision (the walk) and a header (the emission); this file only orders the work differently.
//
// k_encode8_blocks gives every block to one lane for the whole encode: run detection, the emit decisions, the packet headers and the
// literal copies all sit in one per-lane state machine, 64 of them in lock step -- ~1 400 VALU instructions per 64-byte step of a wave,
// most of them executed for the lanes that happen to have another run or another 16 literal bytes in this step.  But only the emit
// DECISION is sequential (a run is stored or not depending on the distance to the last stored run and, Packed, on its
// symbol); and a run that is not stored leaves the state alone.  So, per wave and batch of blocks:
//   A  (whole wave, one block at a time)  64 lanes x 64 bytes = the block; equality mask against the successor byte (zero_mask16), the
//      ends of all runs of at least T bytes (T = 3 Packed / 6 plain: shorter ones are never stored) and their starts (wave prefix
//      maximum of the stretch starts), compacted by a wave prefix sum into the batch's candidate list in LDS: start, count, symbol;
//   B  (one lane per block)  walks its block's candidates with the encoder state (lastRLE, lastSymbol, stream position): stored or
//      not, which header form, where in the stream -- ~25 instructions per candidate, nothing else;
//   C  (whole wave, one block at a time)  one lane per STORED run writes the packet header and the literals in front of the run into an
//      output tile in LDS (global loads placed so that their dwords are the tile's dwords; <= 64 bytes in the lane, longer gaps by the
//      whole wave; the block's terminator + last literals are one more item), then the tile goes out in whole lines.  64 items at a
//      time, the list reads + loads of the next 64 issued before the tile writes of these.
// A batch is as many consecutive blocks of the wave's 64 as fit into the candidate list (a 4 KiB block has at most 4096 / 3
// candidates; run-distributed data: 63 per block, 62 of them stored).  Blocks of 1 .. 4 KiB (one 64-byte window per lane).
//
// Where it is used: containers of fewer than 131 072 blocks of 1 .. 4 KiB -- too few blocks for one lane each.  A wave takes
// ceil(blocks / 2304) blocks (9 waves per CU resident: the whole container at once), so the 88 MB frame's 21 600 blocks are 2 160 waves
// of 10 blocks: 144 us where the split encode (cuts inside the blocks, chunk table, ring encoders' chunk mode, placement) took 220;
// 64 MiB run-distributed 111 against 213 (rle8_multi: 110 / 197, 97 / 194).
// Where it is NOT used: big containers.  8 GiB run-distributed, rle8_packed_multi: 7.4 ms of kernel against the ring encoder's 5.2
// (-DHSRLE_EXPERIMENTS builds: HSRLE_RUNLIST=1 sends them here all the same).  Measured on the way: stores straight from
// the lanes (16-byte pieces at byte offsets, 5 per stored run) 6.6 ms; four runs per lane in flight 6.9 ms; the output tile (this file)
// 7.2 - 7.4 ms.  The wave's instruction count fell as planned (615 VALU per block against ~1 400), but a wave is alone with its block:
// every phase ends in a wait (list in LDS -> loads -> tile -> store) that 9 waves per CU do not cover (SQ_WAIT_ANY 64 % of the wave
// cycles), where the ring encoder keeps 64 independent streams per wave in flight.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // wave_sync

namespace hsrle {

constexpr uint32_t kRunListCap = 1408u;        // candidates per wave and batch (>= 4096 / 3: any one block fits)
constexpr uint32_t kRunListMaxBlock = 4096u;
constexpr uint32_t kRunListOutTile = 4352u;    // >= the slot of a 4 KiB block (rle_compress_bounds(4096) rounded up to 16: 4304)

template <int FAM>
__global__ __launch_bounds__(64) void k_encode8_runlist(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                        uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes, uint32_t bpw,
                                                        const uint32_t *__restrict__ sel)
{
  if (sel != nullptr && sel[0] != 1u) return;                          // (big containers of 3 / 4 byte symbols: k_list_decide chose the ring encoder, hsrle_ring_probe.hip.h)
  static_assert(FAM == PLAIN || FAM == PACKED || FAM == LUT3 || FAM == LUT7 || (FAM >= SHORT0 && FAM <= SHORT7), "rle8_multi / rle8_packed_multi / rle8_{3,7}symlut / their Short family");
  using TR = Traits<FAM, 1, 0>;
  constexpr bool PK = FAM == PACKED, LT = TR::kLut, SH = TR::kShort;
  constexpr int K = TR::K;
  constexpr uint32_t T = SH ? TR::SMINS : ((FAM == PLAIN) ? 6u : 3u);    // the shortest run any state stores (Short with a list: 2)
  constexpr uint32_t CAP = (T == 2u) ? 2112u : kRunListCap;             // (runs of two bytes: up to 2048 candidates in a 4 KiB block)
  constexpr uint32_t kStored = 1u, kSame = 2u, kLong = 4u;

  __shared__ uint32_t cand[CAP];              // start | count << 16
  __shared__ __attribute__((aligned(16))) uint32_t info[CAP];   // stream offset of the packet | gap << 14 | flags << 27 (phases B, C)
  __shared__ uint8_t csym[CAP];
  __shared__ __attribute__((aligned(16))) uint8_t otile[kRunListOutTile];   // the stream of the block under phase C
  __shared__ uint32_t bOff[65];               // first candidate of the batch's blocks
  static_assert(CAP * 4u >= kRunListMaxBlock, "the tile lives in info[]");
  uint8_t *const tile = (uint8_t *)info;      // the block under phase A (symbol lookups): phase A never touches info[]

  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = xcd_tile(blockIdx.x, gridDim.x) * bpw;          // bpw (<= 64) blocks per wave: small containers spread over the device
  const uint32_t wgLast = umin(wgFirst + bpw, nBlocks);
  if (wgFirst >= nBlocks) return;

  auto block_len = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t {
    const uint64_t at = (uint64_t)b * B;
    return (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  };
  // 16 input bytes (zeros beyond the input)
  auto load16 = [&](uint64_t g) __attribute__((always_inline)) -> u32x4 {
    if (g + 16u <= U) return ld128(in + g);
    return load16_edge(in, (int64_t)g, U);
  };
  struct Win { u32x4 a, b, c, d; };
  auto load_block = [&](uint32_t b) __attribute__((always_inline)) -> Win {
    Win w; w.a = w.b = w.c = w.d = u32x4{ 0, 0, 0, 0 };
    const uint32_t n = block_len(b);
    if (lane * 64u < n)
    {
      const uint64_t g = (uint64_t)b * B + lane * 64u;
      w.a = load16(g); w.b = load16(g + 16u); w.c = load16(g + 32u); w.d = load16(g + 48u);
    }
    return w;
  };

  // ---- building a block's stream in the output tile ----
  // 16 input bytes at p (zeros beyond the input: only the very last bytes of the last block get there)
  auto load16p = [&](const uint8_t *p) __attribute__((always_inline)) -> u32x4 { return load16((uint64_t)(p - in)); };
  auto tile_dwords = [&](uint32_t q, u32x4 v, uint32_t nd) __attribute__((always_inline)) {               // the low nd (0..4) dwords of v at the dword-aligned tile offset q
    uint32_t *const o = (uint32_t *)(otile + q);
    if (nd > 0u) o[0] = v.x;
    if (nd > 1u) o[1] = v.y;
    if (nd > 2u) o[2] = v.z;
    if (nd > 3u) o[3] = v.w;
  };
  auto tile_bytes = [&](uint32_t q, uint32_t w, uint32_t nb) __attribute__((always_inline)) {             // the low nb (0..3) bytes of w at tile offset q
    if (nb > 0u) otile[q] = (uint8_t)w;
    if (nb > 1u) otile[q + 1u] = (uint8_t)(w >> 8);
    if (nb > 2u) otile[q + 2u] = (uint8_t)(w >> 16);
  };
  // len (<= 64) literal bytes from src to tile offset d, by this lane: the global loads are placed so that their dwords are the tile's
  auto lane_copy = [&](const uint8_t *src, uint32_t d, uint32_t len) __attribute__((always_inline)) {
    uint32_t h = (4u - (d & 3u)) & 3u;
    if (h > len) h = len;
    const uint32_t rem = len - h, full = rem >> 4, r = rem & 15u;
    const uint8_t *const p = src + h;
    const uint32_t q = d + h;
    u32x4 v0, v1, v2, v3;
    const uint32_t head = (h != 0u) ? load16p(src).x : 0u;               // (a stored run of >= 3 bytes follows the literals)
    if (full > 0u) v0 = ld128(p);
    if (full > 1u) v1 = ld128(p + 16u);
    if (full > 2u) v2 = ld128(p + 32u);
    if (full > 3u) v3 = ld128(p + 48u);
    u32x4 vt = u32x4{ 0, 0, 0, 0 };
    if (r != 0u) vt = load16p(p + 16u * full);
    tile_bytes(d, head, h);
    if (full > 0u) tile_dwords(q, v0, 4u);
    if (full > 1u) tile_dwords(q + 16u, v1, 4u);
    if (full > 2u) tile_dwords(q + 32u, v2, 4u);
    if (full > 3u) tile_dwords(q + 48u, v3, 4u);
    const uint32_t nd = r >> 2, qt = q + 16u * full;
    tile_dwords(qt, vt, nd);
    tile_bytes(qt + 4u * nd, nd == 0u ? vt.x : (nd == 1u ? vt.y : (nd == 2u ? vt.z : vt.w)), r & 3u);
  };
  // len literal bytes by the whole wave (src, d, len uniform)
  auto wave_copy = [&](const uint8_t *src, uint32_t d, uint32_t len) __attribute__((always_inline)) {
    uint32_t h = (4u - (d & 3u)) & 3u;
    if (h > len) h = len;
    const uint32_t rem = len - h, full = rem >> 4, r = rem & 15u;
    const uint8_t *const p = src + h;
    const uint32_t q = d + h;
    for (uint32_t t = lane; t < full; t += 64u) tile_dwords(q + 16u * t, ld128(p + 16u * t), 4u);
    if (lane == 63u)
    {
      if (h != 0u) tile_bytes(d, load16p(src).x, h);
      if (r != 0u)
      {
        const u32x4 vt = load16p(p + 16u * full);
        const uint32_t nd = r >> 2, qt = q + 16u * full;
        tile_dwords(qt, vt, nd);
        tile_bytes(qt + 4u * nd, nd == 0u ? vt.x : (nd == 1u ? vt.y : (nd == 2u ? vt.z : vt.w)), r & 3u);
      }
    }
  };
  auto tile_hdr = [&](uint32_t q, uint64_t lo, uint64_t hi, uint32_t n) __attribute__((always_inline)) {    // n (<= 12) header bytes, any alignment
#pragma unroll
    for (uint32_t i = 0; i < 12u; i++)
      if (i < n) otile[q + i] = (uint8_t)((i < 8u ? lo >> (8u * i) : hi >> (8u * (i - 8u))) & 0xFFull);
  };

  // ---- phases B and C for the blocks [b0, b1) whose candidates are in the list ----
  auto flush = [&](uint32_t b0, uint32_t b1) __attribute__((always_inline)) {
    wave_sync();
    // B: one lane per block
    const uint32_t myBlock = b0 + lane;
    const bool walker = myBlock < b1;
    uint32_t n = 0, lastRLE = 0, lastSym = 0, opos = TR::kHeaderSize;     // (behind the stream header: size, compressed size, and the mode byte 0 = multi of the codecs without a list)
    [[maybe_unused]] uint64_t lutw = 0;                                   // LUT: the move-to-front list, entry k in byte k (rleX_Xsl.h: starts as 00 7F FF 01 7E 80 FE)
    if constexpr (TR::kMtf) lutw = 0x00FE807E01FF7F00ull & ((1ull << (8 * K)) - 1ull);
    if (walker)
    {
      n = block_len(myBlock);
      const uint32_t c1 = bOff[lane + 1u];
      for (uint32_t c = bOff[lane]; c < c1; c++)
      {
        const uint32_t cv = cand[c], st = cv & 0xFFFFu, count = cv >> 16;
        [[maybe_unused]] const uint32_t sym = csym[c];
        const uint32_t e = st + count, gap = st - lastRLE, rng = gap + 1u;
        uint32_t flags = 0, hl = 0;
        if constexpr (SH)
        {
          // rleX_Xsl_short.h:152-357 (k_encode8_blocks handle_run)
          const uint32_t range = gap + 2u;
          [[maybe_unused]] uint32_t m = (uint32_t)K;
          if constexpr (K > 0)
          {
#pragma unroll
            for (int q = K - 1; q >= 0; q--)
              if (((lutw >> (8 * q)) & 0xFFull) == (uint64_t)sym) m = (uint32_t)q;
          }
          const int32_t sc = (int32_t)count - (int32_t)TR::SMINS + 2;
          const bool pack1 = gap <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
          uint32_t pen = (K > 0 && m == (uint32_t)K) ? 1u : 0u;
          if (!pack1)
          {
            pen += 2u;
            if (!(sc <= (int32_t)TR::SMAXTC && range <= TR::SMAXTR))
              pen += ((range <= 0xFFFFFu) ? (range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((sc <= 0xFFFFF) ? (sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
          }
          if (count >= TR::SMINL || count >= TR::SMINS + pen)
          {
            if constexpr (K > 0)
            {
              const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
              const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
              const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
              lutw = keepHi | (low << 8) | (uint64_t)sym;
            }
            const uint32_t scu = (uint32_t)sc;
            flags = kStored | (m << 1);
            hl = pack1 ? 1u : 3u + (scu > TR::SMAXTC ? (scu <= 0xFFFFu ? 2u : 4u) : 0u) + (range > TR::SMAXTR ? (range <= 0xFFFFu ? 2u : 4u) : 0u);
            if (K == 0 || m == (uint32_t)K) hl += 1u;
          }
        }
        else if constexpr (LT)
        {
          // rleX_Xsl.h:116-195 (k_encode8_blocks handle_run)
          constexpr uint32_t MAXC = 127u, MAXR = (1u << TR::RB) - 1u;
          const uint32_t range = gap + 2u, cst = count - 1u;
          uint32_t m = (uint32_t)K;
#pragma unroll
          for (int q = K - 1; q >= 0; q--)
            if (((lutw >> (8 * q)) & 0xFFull) == (uint64_t)sym) m = (uint32_t)q;
          uint32_t pen = (range <= 0xFFFFFu) ? (range <= MAXR ? 0u : 2u) : 4u;   // 0xFFFFF vs 0xFFFF in the writer: A.5 q3
          pen += (cst <= 0xFFFFFu) ? (cst <= MAXC ? 0u : 2u) : 4u;
          pen += (m == (uint32_t)K) ? 1u : 0u;
          if (count >= 11u || count >= 3u + pen)
          {
            const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
            const uint64_t keepHi = lutw & ~((1ull << (8u * (limit + 1u))) - 1ull);
            const uint64_t low = lutw & ((1ull << (8u * limit)) - 1ull);
            lutw = keepHi | (low << 8) | (uint64_t)sym;
            flags = kStored | (m << 1);
            hl = 2u + (m == (uint32_t)K ? 1u : 0u) + (cst > MAXC ? (cst <= 0xFFFFu ? 2u : 4u) : 0u) + (range > MAXR ? (range <= 0xFFFFu ? 2u : 4u) : 0u);
          }
        }
        else if constexpr (PK)
        {
          // body / tail split of the canonical AVX2 encoder (k_encode8_blocks handle_run)
          const int32_t kk = (int32_t)(count - 1u) / 32;
          const bool body = (e < n) && ((int32_t)st + 1 + 32 * kk < (int32_t)n - 32);
          bool same = false, emit;
          if (body)
          {
            same = sym == lastSym;
            emit = count >= 11u || (rng <= 127u && ((same && count >= 3u) || count >= 4u));
          }
          else
            emit = count >= 11u;
          if (emit)
          {
            if (body) lastSym = sym;
            flags = kStored | (same ? kSame : 0u) | (rng <= 127u ? 0u : kLong);
            hl = ((count - 2u) <= 127u ? 1u : 5u) + (same ? 0u : 1u) + (rng <= 127u ? 1u : 4u);
          }
        }
        else
        {
          flags = kStored | (rng <= 255u ? 0u : kLong);                   // rle8_extreme_cpu.h:974: count >= 6 is all it takes
          hl = 1u + ((count - 5u) <= 255u ? 1u : 5u) + (rng <= 255u ? 1u : 5u);
        }
        info[c] = opos | (gap << 14) | (flags << 27);                     // (flags: stored | same << 1 | long range << 2, or stored | list index << 1)
        if (flags != 0u) { opos += hl + gap; lastRLE = e; }
      }
    }
    const bool ended = walker && lastRLE >= n;                            // a stored run reached the end: the end terminator, no literals
    const uint32_t termOff = opos;
    const uint32_t termLen = SH ? (ended ? 7u : 9u) + (K == 0 ? 1u : 0u) : (LT ? (ended ? 6u : 8u) : (PK ? 9u : 11u));
    const uint32_t kLit = (!walker || ended) ? 0u : n - lastRLE;
    const uint32_t size = termOff + termLen + kLit;
    const uint32_t endedI = ended ? 1u : 0u;
    if (walker) sizes[myBlock] = size;
    wave_sync();

    // C: block by block, 64 items at a time (a block's stored runs, then its terminator + last literals as one more item), the stream in
    // the output tile, then out in whole lines.  Two stages, software pipelined: stage 1 of the NEXT 64 items (list reads, headers, the
    // literal loads -- always the same five loads per lane, so that the wait in front of stage 2 can leave them in flight) is issued
    // before stage 2 of these (tile writes, copy-out).
    struct Stage
    {
      bool on, lng, tail; uint32_t off, hn, d, gap, head; uint64_t lo, hi; const uint8_t *src; u32x4 v0, v1, v2, v3;
    };
    auto stage1 = [&](Stage &s, uint32_t j, uint32_t cb, bool live) __attribute__((always_inline)) {
      const uint32_t b = b0 + j;
      const uint64_t blockAt = (uint64_t)b * B;
      const uint32_t c1 = bOff[j + 1u], c = cb + lane;
      s.on = false; s.lng = false; s.tail = false; s.off = 0; s.hn = 0; s.d = 0; s.gap = 0; s.lo = 0; s.hi = 0; s.src = in;
      // (values of the lane that walked this block; read here, where all lanes are active: a readlane of something computed under a
      //  lane mask that excludes the source lane reads a stale register)
      const uint32_t jTerm = (uint32_t)__builtin_amdgcn_readlane((int)termOff, (int)j), jLit = (uint32_t)__builtin_amdgcn_readlane((int)kLit, (int)j);
      const uint32_t jLast = (uint32_t)__builtin_amdgcn_readlane((int)lastRLE, (int)j);
      const bool jEnded = __builtin_amdgcn_readlane((int)endedI, (int)j) != 0;
      const uint32_t jTermLen = (uint32_t)__builtin_amdgcn_readlane((int)termLen, (int)j);
      if (live && c < c1)
      {
        const uint32_t iv = info[c], flags = iv >> 27;
        if (flags & kStored)
        {
          const uint32_t cv = cand[c], st = cv & 0xFFFFu, count = cv >> 16, sym = csym[c];
          const uint32_t off = iv & 0x3FFFu, rng = ((iv >> 14) & 0x1FFFu) + 1u;
          uint64_t lo, hi = 0; uint32_t hn;
          if constexpr (SH)
          {
            // one byte (list index | count | range), or three (list index | all ones, count, range) + the values that do not fit; then the symbol unless the list has it
            const uint32_t m = flags >> 1, gapv = rng - 1u, range = rng + 1u;                      // (rng = gap + 1 here)
            const int32_t sc = (int32_t)count - (int32_t)TR::SMINS + 2;
            const bool pack1 = gapv <= TR::SMAXPR && (uint32_t)(sc - 2) <= TR::SMAXPC;
            const uint32_t scu = (uint32_t)sc, mi = (K > 0) ? m << (TR::SCB + TR::SRBP) : 0u;
            if (pack1) { lo = (uint64_t)((mi | ((uint32_t)(sc - 2) << TR::SRBP) | gapv) & 0xFFu); hn = 1u; }
            else
            {
              const uint32_t scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u), rx = (range <= TR::SMAXTR) ? range : (range <= 0xFFFFu ? 1u : 0u);
              const uint32_t b0 = (mi | (TR::SCINV << TR::SRBP) | ((scx << (TR::SRB - 8u)) >> 8)) & 0xFFu, b1 = ((scx << (TR::SRB - 8u)) | (rx >> 8)) & 0xFFu, b2 = rx & 0xFFu;
              lo = (uint64_t)(b0 | (b1 << 8) | (b2 << 16));
              hn = 3u;
              if (scx != scu) { lo |= (uint64_t)scu << 24; hn += (scu <= 0xFFFFu) ? 2u : 4u; }                 // hn <= 7
              if (rx != range)
              {
                lo |= (uint64_t)range << (8u * hn);
                if (hn > 4u) hi = (uint64_t)range >> (64u - 8u * hn);
                hn += (range <= 0xFFFFu) ? 2u : 4u;                                                           // hn <= 11
              }
            }
            if (K == 0 || m == (uint32_t)K)
            {
              if (hn < 8u) lo |= (uint64_t)sym << (8u * hn); else hi |= (uint64_t)sym << (8u * (hn - 8u));
              hn += 1u;
            }
          }
          else if constexpr (LT)
          {
            // u16 (list index | count field | range field) [symbol] [count u16 / u32] [range u16 / u32]
            constexpr uint32_t MAXC = 127u, MAXR = (1u << TR::RB) - 1u;
            const uint32_t m = flags >> 1, range = rng + 1u, cst = count - 1u;     // (rng = gap + 1 here)
            const uint32_t c7 = (cst <= MAXC) ? cst : (cst <= 0xFFFFu ? 1u : 0u), r7 = (range <= MAXR) ? range : (range <= 0xFFFFu ? 1u : 0u);
            lo = (uint64_t)(((m << (K == 3 ? 14 : 13)) | (c7 << TR::RB) | r7) & 0xFFFFu);
            hn = 2u;
            if (m == (uint32_t)K) { lo |= (uint64_t)sym << 16; hn = 3u; }
            if (cst != c7) { lo |= (uint64_t)cst << (8u * hn); hn += (cst <= 0xFFFFu) ? 2u : 4u; }      // hn <= 3: fits the low word
            if (range != r7)
            {
              lo |= (uint64_t)range << (8u * hn);                          // hn <= 7
              if (hn > 4u) hi = (uint64_t)range >> (64u - 8u * hn);
              hn += (range <= 0xFFFFu) ? 2u : 4u;
            }
          }
          else if constexpr (PK)
          {
            // count byte (| same bit) [count u32] [symbol] range byte / u32
            const uint32_t cc = count - 2u, sm = (flags & kSame) ? 0x80u : 0u;
            lo = (cc <= 127u) ? (uint64_t)(cc | sm) : ((uint64_t)sm | ((uint64_t)cc << 8));
            hn = (cc <= 127u) ? 1u : 5u;
            if (!(flags & kSame)) { lo |= (uint64_t)sym << (8u * hn); hn += 1u; }
            const uint64_t rv = (flags & kLong) ? (uint64_t)((rng << 1) | 1u) : (uint64_t)(rng << 1);
            lo |= rv << (8u * hn);
            if (hn > 4u) hi = rv >> (64u - 8u * hn);
            hn += (flags & kLong) ? 4u : 1u;
          }
          else
          {
            // symbol, count byte / 0 + u32, range byte / 0 + u32
            const uint32_t cc = count - 5u;
            lo = (uint64_t)sym;
            if (cc <= 255u) { lo |= (uint64_t)cc << 8; hn = 2u; } else { lo |= (uint64_t)cc << 16; hn = 6u; }
            if (!(flags & kLong)) { lo |= (uint64_t)rng << (8u * hn); hn += 1u; }
            else
            {
              hn += 1u;                                                    // the zero byte
              lo |= (uint64_t)rng << (8u * hn);
              if (hn > 4u) hi = (uint64_t)rng >> (64u - 8u * hn);
              hn += 4u;
            }
          }
          s.on = true; s.off = off; s.hn = hn; s.lo = lo; s.hi = hi; s.gap = rng - 1u;
          s.src = in + blockAt + st - s.gap;
          s.d = off + hn;
        }
      }
      else if (live && c == c1)
      {
        // the block's last item: terminator + the literals behind the last stored run
        if constexpr (SH)
        {
          // rleX_Xsl_short.h:470-523: all-ones count byte, STB, then 1, u16 0, u16 0 (end) or 0, u16 0, u32 literals + 2; one more zero byte in the codec without a list
          const uint32_t b0 = (TR::SCINV << TR::SRBP) & 0xFFu;
          if (jEnded) { s.lo = (uint64_t)(b0 | (TR::STB << 8) | (1u << 16)); s.hi = 0; }
          else { s.lo = (uint64_t)(b0 | (TR::STB << 8)) | ((uint64_t)(jLit + 2u) << 40); s.hi = (uint64_t)(jLit + 2u) >> 24; }
        }
        else if constexpr (LT)
        {
          // end: u16 (1 << RB) | 1, u16 0, u16 0;  literals: u16 1 << RB, u16 0, u32 literals + 2
          if (jEnded) { s.lo = (uint64_t)((1u << TR::RB) | 1u); s.hi = 0; }
          else { s.lo = (uint64_t)(1u << TR::RB) | ((uint64_t)(jLit + 2u) << 32); s.hi = 0; }
        }
        else if constexpr (PK)
        {
          // 0x80, u32 0, then u32 1 (end) or u32 ((literals + 1) << 1 | 1)
          const uint32_t v = jEnded ? 1u : (((jLit + 1u) << 1) | 1u);
          s.lo = 0x80ull | ((uint64_t)v << 40); s.hi = (uint64_t)(v >> 24);
        }
        else
        {
          // 0, 0, u32 0, 0, then u32 0 (end) or u32 (literals + 1)
          const uint32_t v = jEnded ? 0u : jLit + 1u;
          s.lo = (uint64_t)v << 56; s.hi = (uint64_t)(v >> 8);
        }
        s.on = true; s.tail = true; s.off = jTerm; s.hn = jTermLen; s.gap = jLit;
        s.src = in + blockAt + jLast;
        s.d = jTerm + jTermLen;
      }
      s.lng = s.on && s.gap > 64u;
      // the loads may run up to 67 bytes past the literals: not in the last blocks of the input (those copy in stage 2)
      const bool fast = blockAt + (uint64_t)B + 80ull <= U;
      const bool ld = s.on && !s.lng && s.gap != 0u && fast;
      const uint8_t *const a = ld ? s.src : in;
      uint32_t h = (4u - (s.d & 3u)) & 3u;
      if (h > s.gap) h = s.gap;
      const uint8_t *const p = a + (ld ? h : 0u);
      s.head = ld32(a);
      s.v0 = ld128(p); s.v1 = ld128(p + 16u); s.v2 = ld128(p + 32u); s.v3 = ld128(p + 48u);
    };
    auto stage2 = [&](Stage &s, uint32_t j, uint32_t cb) __attribute__((always_inline)) {
      const uint32_t b = b0 + j;
      const uint64_t blockAt = (uint64_t)b * B;
      const bool fast = blockAt + (uint64_t)B + 80ull <= U;
      const uint32_t jSize = (uint32_t)__builtin_amdgcn_readlane((int)size, (int)j);
      if (s.on)
      {
        tile_hdr(s.off, s.lo, s.hi, s.hn);
        if (s.tail) { ((uint32_t *)otile)[0] = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)j); ((uint32_t *)otile)[1] = jSize; if constexpr (TR::kHeaderSize == 9u) otile[8] = 0; }
        if (!s.lng && s.gap != 0u)
        {
          if (fast)
          {
            uint32_t h = (4u - (s.d & 3u)) & 3u;
            if (h > s.gap) h = s.gap;
            const uint32_t rem = s.gap - h, full = rem >> 4, r = rem & 15u, q = s.d + h;
            tile_bytes(s.d, s.head, h);
            if (full > 0u) tile_dwords(q, s.v0, 4u);
            if (full > 1u) tile_dwords(q + 16u, s.v1, 4u);
            if (full > 2u) tile_dwords(q + 32u, s.v2, 4u);
            if (full > 3u) tile_dwords(q + 48u, s.v3, 4u);
            u32x4 vt = s.v3;                                               // (not a ?: chain of the members: that is an indexed access and puts the set in scratch memory)
            if (full == 0u) vt = s.v0; else if (full == 1u) vt = s.v1; else if (full == 2u) vt = s.v2;
            const uint32_t nd = r >> 2, qt = q + 16u * full;
            tile_dwords(qt, vt, nd);
            tile_bytes(qt + 4u * nd, nd == 0u ? vt.x : (nd == 1u ? vt.y : (nd == 2u ? vt.z : vt.w)), r & 3u);
          }
          else
            lane_copy(s.src, s.d, s.gap);
        }
      }
      // literals of more than 64 bytes: the whole wave, one after the other
      uint64_t todo = __ballot(s.lng);
      while (todo != 0ull)
      {
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint64_t sp = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)s.src, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)s.src >> 32), l) << 32);
        wave_copy((const uint8_t *)sp, (uint32_t)__builtin_amdgcn_readlane((int)s.d, l), (uint32_t)__builtin_amdgcn_readlane((int)s.gap, l));
      }
      if (cb + 64u > bOff[j + 1u])
      {
        // the block's last items: its stream is complete
        wave_sync();
        uint8_t *const slot = slots + (uint64_t)b * slotStride;
        for (uint32_t k = lane * 16u; k < jSize; k += 1024u) st128(slot + k, lds_ld128(otile + k));
        wave_sync();
      }
    };
    {
      // (two register sets in turn: a copy of the set that is being loaded would wait for its loads)
      uint32_t j = 0, cb = bOff[0];
      auto advance = [&]() __attribute__((always_inline)) -> bool {
        if (cb + 64u <= bOff[j + 1u]) cb += 64u; else { j++; cb = bOff[j]; }
        return b0 + j < b1;
      };
      bool have = b0 < b1;
      Stage sa, sb;
      stage1(sa, 0u, cb, have);
      while (have)
      {
        const uint32_t ja = j, cba = cb;
        const bool haveB = advance();
        stage1(sb, haveB ? j : 0u, haveB ? cb : 0u, haveB);
        stage2(sa, ja, cba);
        if (!haveB) break;
        const uint32_t jb = j, cbb = cb;
        have = advance();
        stage1(sa, have ? j : 0u, have ? cb : 0u, have);
        stage2(sb, jb, cbb);
      }
    }
  };

  // ---- phase A over the wave's blocks ----
  uint32_t batchFirst = wgFirst, used = 0;
  Win nx = load_block(wgFirst), nx2 = nx;
  if (wgFirst + 1u < wgLast) nx2 = load_block(wgFirst + 1u);
  for (uint32_t b = wgFirst; b < wgLast; b++)
  {
    const Win w = nx;
    nx = nx2;
    if (b + 2u < wgLast) nx2 = load_block(b + 2u);                       // two blocks ahead: the loads of a block fly for two blocks' phase A
    const uint32_t n = block_len(b);
    const uint32_t myAt = lane * 64u;
    lds_st128(tile + myAt, w.a); lds_st128(tile + myAt + 16u, w.b); lds_st128(tile + myAt + 32u, w.c); lds_st128(tile + myAt + 48u, w.d);   // (for the symbol lookups)

    // equality with the successor byte: bit i = d[myAt + i] == d[myAt + i + 1], only where both lie in the block
    const uint32_t nxt = (uint32_t)__shfl_down((int)w.a.x, 1, 64);       // (lane 63's successor lies outside the block: masked below)
    const uint32_t m0 = zero_mask16(w.a.x ^ alignbyte(w.a.y, w.a.x, 1), w.a.y ^ alignbyte(w.a.z, w.a.y, 1), w.a.z ^ alignbyte(w.a.w, w.a.z, 1), w.a.w ^ alignbyte(w.b.x, w.a.w, 1));
    const uint32_t m1 = zero_mask16(w.b.x ^ alignbyte(w.b.y, w.b.x, 1), w.b.y ^ alignbyte(w.b.z, w.b.y, 1), w.b.z ^ alignbyte(w.b.w, w.b.z, 1), w.b.w ^ alignbyte(w.c.x, w.b.w, 1));
    const uint32_t m2 = zero_mask16(w.c.x ^ alignbyte(w.c.y, w.c.x, 1), w.c.y ^ alignbyte(w.c.z, w.c.y, 1), w.c.z ^ alignbyte(w.c.w, w.c.z, 1), w.c.w ^ alignbyte(w.d.x, w.c.w, 1));
    const uint32_t m3 = zero_mask16(w.d.x ^ alignbyte(w.d.y, w.d.x, 1), w.d.y ^ alignbyte(w.d.z, w.d.y, 1), w.d.z ^ alignbyte(w.d.w, w.d.z, 1), w.d.w ^ alignbyte(nxt, w.d.w, 1));
    uint64_t e64 = (uint64_t)(m0 | (m1 << 16)) | ((uint64_t)(m2 | (m3 << 16)) << 32);
    const uint32_t inBlock = (n > myAt) ? umin(64u, n - myAt) : 0u;      // positions of this lane inside the block
    const uint32_t validBits = (n > myAt + 1u) ? umin(64u, n - 1u - myAt) : 0u;
    e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t wmask = (inBlock >= 64u) ? ~0ull : ((1ull << inBlock) - 1ull);

    // the match bits in front of every position: bit i of (e64 << k | pe >> (64 - k)) = e[myAt + i - k]
    uint64_t pe = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(e64 >> 32), 1, 64) << 32;     // only the top bits of the lane below are needed
    if (lane == 0u) pe = 0ull;
    const uint64_t h1 = (e64 << 1) | (pe >> 63);
    uint64_t longEnough = h1;                                            // T - 1 match bits in front
    if constexpr (T >= 3u) longEnough &= (e64 << 2) | (pe >> 62);
    if constexpr (T == 6u) longEnough &= ((e64 << 3) | (pe >> 61)) & ((e64 << 4) | (pe >> 60)) & ((e64 << 5) | (pe >> 59));
    const uint64_t ustarts = e64 & ~h1;                                  // a stretch of match bits begins: the run's first byte
    const uint64_t fends = ~e64 & longEnough & wmask;                    // the last byte of a run of at least T bytes

    // two wave scans in one shuffle per step: exclusive prefix sum of the candidates (low 12 bits: <= 2048) and exclusive prefix maximum
    // of the last stretch start + 1 (above them)
    const uint32_t cnt = (uint32_t)__builtin_popcountll(fends);
    const uint32_t lastStart1 = (ustarts != 0ull) ? myAt + 64u - (uint32_t)__builtin_clzll(ustarts) : 0u;   // position + 1, 0 = none
    uint32_t sum = cnt, mx = lastStart1;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1)
    {
      const uint32_t pv = (uint32_t)__shfl_up((int)(sum | (mx << 12)), d, 64);
      if (lane >= d) { sum += pv & 0xFFFu; mx = (pv >> 12) > mx ? (pv >> 12) : mx; }
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);
    uint32_t rank = sum - cnt;
    uint32_t carryStart1 = (uint32_t)__shfl_up((int)mx, 1, 64);
    if (lane == 0u) carryStart1 = 0u;

    if (used + total > CAP)
    {
      if (lane == 0u) bOff[b - batchFirst] = used;
      flush(batchFirst, b);
      batchFirst = b; used = 0u;
      // the tile lives in info[], which the flush has just used: this block's bytes once more (kept in registers across the flush they
      // cost the kernel a wave per SIMD)
      const Win again = load_block(b);
      lds_st128(tile + myAt, again.a); lds_st128(tile + myAt + 16u, again.b); lds_st128(tile + myAt + 32u, again.c); lds_st128(tile + myAt + 48u, again.d);
    }
    if (lane == 0u) bOff[b - batchFirst] = used;

    wave_sync();
    uint64_t f = fends;
    while (f != 0ull)
    {
      const uint32_t i = (uint32_t)__builtin_ctzll(f);
      f &= f - 1ull;
      const uint64_t sBelow = ustarts & ((1ull << i) - 1ull);
      const uint32_t st = (sBelow != 0ull) ? myAt + 63u - (uint32_t)__builtin_clzll(sBelow) : carryStart1 - 1u;
      const uint32_t count = myAt + i + 1u - st;
      cand[used + rank] = st | (count << 16);
      csym[used + rank] = tile[myAt + i];                                // (every byte of the run is the symbol)
      rank++;
    }
    wave_sync();
    used += total;
  }
  if (lane == 0u) bOff[wgLast - batchFirst] = used;
  flush(batchFirst, wgLast);
}

} // namespace hsrle
