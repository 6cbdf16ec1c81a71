// hsrle_encodeSr.hip.h -- the run list encoder (hsrle_encode8r.hip.h: the whole wave per block, one lane per block only for the emit
// decisions) for the codecs of 2, 3, 4, 6 and 8 byte symbols: plain, Packed, 3 / 7 symbol LUT and the Short family (0 / 1 / 3 / 7 symbol list);
// sym- and byte-aligned.
//
// Replaces: src/rleX_extreme_cpu_encode.h:14-609 (32 / 64 bit), src/rle48_extreme_cpu_encode.h, src/rleX_Xsl_multibyte_encoder.h:18-370 +
//           src/rleX_Xsl.h:114-264 (LUT), src/rleX_Xsl_short.h:152-357 + :470-523 (Short: process_symbol, terminators) -- the same streams as
//           k_encodeS_blocks (hsrle_encodeS.hip.h), whose handle_run is restated here as a decision (walk) and a header (emission).
//
// What differs from the 8 bit version:
//   A  match bits m[j] = (d[j] == d[j + S]) (false from n - S on); a candidate is a stretch of at least S set bits: (start s, first
//      clear position z, the S bytes at s).  Run enumeration continues at the END of a run whether or not it was stored,
//      and that end can lie up to S bytes behind z -- inside the next stretch.  So the run a stretch yields depends on the run in front
//      of it; the walker resolves that: p = max(s, search position), no run if fewer than S bits are left, the symbol rotated by
//      (p - s) mod S bytes (the bytes of a stretch have period S).
//   B  the walk carries the search position, lastRLE, the Packed codecs' last symbol or the LUT codecs' move-to-front list; it leaves the
//      run [p, e), its symbol and where its packet goes.
//   C  headers of up to 19 bytes (Short, three-byte form with both 32 bit fields and an 8 byte symbol).
// Used for containers of fewer than 131 072 blocks of 1 .. 4 KiB (hsrle_launch.h: run_list_applies); the 88 MB frame of BASELINE config 3
// (rle64_3symlut_byte) is the case it was built for.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h" // wave_sync
#include <type_traits>

namespace hsrle {

// candidates per wave and batch: any one 4 KiB block fits (a candidate takes S + 1 positions at least: 819 / 585 / 455), and no more than
// that -- LDS is waves (S = 8: 14.7 KB, 10 waves per CU; S = 2: 21.6 KB, 7)
#ifdef HSRLE_RL_STAMPS
// diagnostic build: cycles per phase, summed over the waves' first lanes: [0] phase A, [1] phase B, [2] phase C, [3] waves, [4] flushes, [5] candidates
__device__ unsigned long long g_rl_stamps[8];
#define HS_RLSTAMP(i) { const unsigned long long t_ = __builtin_readcyclecounter(); rlst[i] += t_ - rlt; rlt = t_; }
#else
#define HS_RLSTAMP(i)
#endif
constexpr uint32_t run_list_cap(int S) { return S == 8 ? 832u : (S == 6 ? 832u : (S == 4 ? 832u : (S == 3 ? 1024u : 1408u))); }

template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_encodeS_runlist(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks,
                                                        uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes, uint32_t bpw,
                                                        const uint32_t *__restrict__ sel)
{
  if (sel != nullptr && sel[0] != 1u) return;                          // (big containers of 3 / 4 byte symbols: k_list_decide chose the ring encoder, hsrle_ring_probe.hip.h)
  static_assert(FAM == PLAIN || FAM == PACKED || FAM == LUT3 || FAM == LUT7 || (FAM >= SHORT0 && FAM <= SHORT7), "plain, Packed, LUT, Short");
  static_assert(S == 2 || S == 3 || S == 4 || S == 6 || S == 8, "2 .. 8 byte symbols");
  using TR = Traits<FAM, S, AL>;
  constexpr uint32_t SU = (uint32_t)S;
  constexpr int K = TR::K;
  constexpr uint32_t CAP = run_list_cap(S);
  constexpr uint64_t SYMMASK = (S == 8) ? ~0ull : ((1ull << (8 * (S & 7))) - 1ull);

  __shared__ uint32_t cand[CAP];              // A: stretch start | first clear position << 16;  B: run start p | run end e << 16
  constexpr uint32_t INFO = CAP > 1024u ? CAP : 1024u;
  __shared__ __attribute__((aligned(16))) uint32_t info[INFO];   // stream offset of the packet | gap << 14 | stored << 27 | LUT index or same << 28
  using sym_t = typename std::conditional<(S <= 4), uint32_t, uint64_t>::type;
  __shared__ sym_t csym[CAP];                 // A: the S bytes at the stretch start;  B: the run's symbol
  __shared__ __attribute__((aligned(16))) uint8_t otile[4352];   // the stream of the block under phase C (>= the slot of a 4 KiB block)
  __shared__ uint32_t bOff[65];               // first candidate of the batch's blocks
  uint8_t *const tile = (uint8_t *)info;      // the block under phase A (symbol lookups): phase A never touches info[]

#ifdef HSRLE_RL_STAMPS
  unsigned long long rlst[6] = { 0, 0, 0, 0, 0, 0 }, rlt = __builtin_readcyclecounter();
#endif
  const uint32_t lane = threadIdx.x;
  const uint32_t wgFirst = xcd_tile(blockIdx.x, gridDim.x) * bpw;          // bpw (<= 64) blocks per wave
  const uint32_t wgLast = umin(wgFirst + bpw, nBlocks);
  if (wgFirst >= nBlocks) return;

  auto block_len = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t {
    const uint64_t at = (uint64_t)b * B;
    return (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  };
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

  // ---- building a block's stream in the output tile (as in k_encode8_runlist) ----
  auto load16p = [&](const uint8_t *p) __attribute__((always_inline)) -> u32x4 { return load16((uint64_t)(p - in)); };
  auto tile_dwords = [&](uint32_t q, u32x4 v, uint32_t nd) __attribute__((always_inline)) {
    uint32_t *const o = (uint32_t *)(otile + q);
    if (nd > 0u) o[0] = v.x;
    if (nd > 1u) o[1] = v.y;
    if (nd > 2u) o[2] = v.z;
    if (nd > 3u) o[3] = v.w;
  };
  auto tile_bytes = [&](uint32_t q, uint32_t w, uint32_t nb) __attribute__((always_inline)) {
    if (nb > 0u) otile[q] = (uint8_t)w;
    if (nb > 1u) otile[q + 1u] = (uint8_t)(w >> 8);
    if (nb > 2u) otile[q + 2u] = (uint8_t)(w >> 16);
  };
  auto lane_copy = [&](const uint8_t *src, uint32_t d, uint32_t len) __attribute__((always_inline)) {
    uint32_t h = (4u - (d & 3u)) & 3u;
    if (h > len) h = len;
    const uint32_t rem = len - h, full = rem >> 4, r = rem & 15u;
    const uint8_t *const p = src + h;
    const uint32_t q = d + h;
    u32x4 v0, v1, v2, v3;
    const uint32_t head = (h != 0u) ? load16p(src).x : 0u;
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
  auto tile_hdr = [&](uint32_t q, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t n) __attribute__((always_inline)) {    // n (<= 20) header bytes, any alignment
#pragma unroll
    for (uint32_t i = 0; i < 20u; i++)
      if (i < n) otile[q + i] = (uint8_t)((i < 8u ? w0 >> (8u * i) : (i < 16u ? w1 >> (8u * (i - 8u)) : w2 >> (8u * (i - 16u)))) & 0xFFull);
  };
  // a header under construction: up to 24 bytes, little endian
  struct Hdr
  {
    uint64_t w0 = 0, w1 = 0, w2 = 0; uint32_t n = 0;
    __device__ __forceinline__ void push(uint64_t v, uint32_t k)           // the low k (1..8) bytes of v; v has no bits above them
    {
      const uint32_t sh = (n & 7u) * 8u;
      const uint64_t lo = v << sh, hi = sh ? (v >> (64u - sh)) : 0ull;
      if (n < 8u) { w0 |= lo; w1 |= hi; } else if (n < 16u) { w1 |= lo; w2 |= hi; } else w2 |= lo;
      n += k;
    }
    __device__ __forceinline__ void b8(uint32_t v) { push((uint64_t)(v & 0xFFu), 1u); }
    __device__ __forceinline__ void b16(uint32_t v) { push((uint64_t)(v & 0xFFFFu), 2u); }
    __device__ __forceinline__ void b32(uint32_t v) { push((uint64_t)v, 4u); }
  };
  // field sizes shared by the walk (lengths) and the emission (bytes): rleX_Xsl.h:116-195, rleX_extreme_cpu_encode.h:165-313
  auto lut_fields = [&](uint32_t count, uint32_t gap, uint32_t &c, uint32_t &c7, uint32_t &range, uint32_t &r7) __attribute__((always_inline)) {
    constexpr uint32_t MAXC = 127u, MAXR = (1u << TR::RB) - 1u;
    range = gap + 2u;
    c = TR::kAligned ? (count / SU - 3u / SU + 2u) : (count - 3u + 2u);
    c7 = (c <= MAXC) ? c : (c <= 0xFFFFu ? 1u : 0u);
    r7 = (range <= MAXR) ? range : (range <= 0xFFFFu ? 1u : 0u);
  };

  // Short family (rleX_Xsl_short.h:152-357): stored count, the one-byte form, the three-byte form's field values
  struct ShortF { int32_t sc; bool pack1; uint32_t scx, rx, range; };
  auto short_fields = [&](uint32_t count, uint32_t gap) __attribute__((always_inline)) -> ShortF {
    ShortF f;
    f.range = gap + 2u;
    f.sc = TR::kAligned ? (int32_t)(count / SU) - (int32_t)(TR::SMINS / SU) + 2 : (int32_t)count - (int32_t)TR::SMINS + 2;
    f.pack1 = gap <= TR::SMAXPR && (uint32_t)(f.sc - 2) <= TR::SMAXPC;
    const uint32_t scu = (uint32_t)f.sc;
    f.scx = (scu <= TR::SMAXTC) ? scu : (scu <= 0xFFFFu ? 1u : 0u);
    f.rx = (f.range <= TR::SMAXTR) ? f.range : (f.range <= 0xFFFFu ? 1u : 0u);
    return f;
  };

  // ---- phases B and C for the blocks [b0, b1) whose candidates are in the list ----
  auto flush = [&](uint32_t b0, uint32_t b1) __attribute__((always_inline)) {
    wave_sync();
    HS_RLSTAMP(0)
#ifdef HSRLE_RL_STAMPS
    rlst[4] += 1; rlst[5] += bOff[b1 - b0];
#endif
    // B: one lane per block
    const uint32_t myBlock = b0 + lane;
    const bool walker = myBlock < b1;
    uint32_t n = 0, lastRLE = 0, from = 0, opos = 8u;                     // (behind the stream header: size, compressed size)
    [[maybe_unused]] uint64_t la = 0;                                     // Packed: the last stored symbol (starts as zeros)
    [[maybe_unused]] uint64_t lut[K ? K : 1];
    if constexpr (TR::kMtf)
    {
      constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };
#pragma unroll
      for (int k = 0; k < K; k++) lut[k] = (0x0101010101010101ull * init[k]) & SYMMASK;
    }
    if (walker)
    {
      n = block_len(myBlock);
      const uint32_t c1 = bOff[lane + 1u];
      for (uint32_t c = bOff[lane]; c < c1; c++)
      {
        const uint32_t cv = cand[c], s = cv & 0xFFFFu, z = cv >> 16;
        const uint32_t p = s > from ? s : from;
        uint32_t iv = 0;
        if (z > p && z - p >= SU)
        {
          const uint32_t L = z - p;
          const uint32_t whole = p + SU + SU * (L / SU);
          uint32_t e = whole;
          if constexpr (!TR::kAligned)
            if (whole + SU <= n) e = p + SU + L;
          from = e;
          // the symbol at p: the one at s rotated by (p - s) mod S bytes
          uint64_t sym = csym[c];
          const uint32_t rot = (p - s) % SU;
          if (rot != 0u) sym = ((sym >> (8u * rot)) | (sym << (8u * (SU - rot)))) & SYMMASK;
          const uint32_t count = e - p, gap = p - lastRLE;
          bool stored; uint32_t hl, tag = 0;
          if constexpr (TR::kShort)
          {
            const ShortF f = short_fields(count, gap);
            [[maybe_unused]] uint32_t m = (uint32_t)K;
            if constexpr (K > 0)
            {
#pragma unroll
              for (int k = K - 1; k >= 0; k--)
                if (lut[k] == sym) m = (uint32_t)k;
            }
            uint32_t pen = (K > 0 && m == (uint32_t)K) ? SU : 0u;
            const uint32_t scu = (uint32_t)f.sc;
            if (!f.pack1)
            {
              pen += 2u;
              if (!(f.sc <= (int32_t)TR::SMAXTC && f.range <= TR::SMAXTR))
                pen += ((f.range <= 0xFFFFFu) ? (f.range <= TR::SMAXTR ? 0u : 2u) : 4u) + ((f.sc <= 0xFFFFF) ? (f.sc <= (int32_t)TR::SMAXTC ? 0u : 2u) : 4u);
            }
            stored = count >= TR::SMINL || count >= TR::SMINS + pen;
            if constexpr (K > 0)
            {
              if (stored)
              {
                const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
                for (int k = K - 1; k >= 1; k--)
                  if ((uint32_t)k <= limit) lut[k] = lut[k - 1];
                lut[0] = sym;
              }
            }
            hl = f.pack1 ? 1u : 3u + (f.scx != scu ? (scu <= 0xFFFFu ? 2u : 4u) : 0u) + (f.rx != f.range ? (f.range <= 0xFFFFu ? 2u : 4u) : 0u);
            if (K == 0 || m == (uint32_t)K) hl += SU;
            tag = m;
          }
          else if constexpr (TR::kLut)
          {
            uint32_t cc, c7, range, r7;
            lut_fields(count, gap, cc, c7, range, r7);
            uint32_t m = (uint32_t)K;
#pragma unroll
            for (int k = K - 1; k >= 0; k--)
              if (lut[k] == sym) m = (uint32_t)k;
            // the penalty uses 0xFFFFF where the writer uses 0xFFFF (A.5 q3; rleX_Xsl.h:130 vs :195)
            uint32_t pen = (range <= 0xFFFFFu) ? (range <= ((1u << TR::RB) - 1u) ? 0u : 2u) : 4u;
            pen += (cc <= 0xFFFFFu) ? (cc <= 127u ? 0u : 2u) : 4u;
            pen += (m == (uint32_t)K) ? 1u : 0u;
            stored = count >= SU + 10u || count >= 3u + pen;
            if (stored)
            {
              const uint32_t limit = (m == (uint32_t)K) ? (uint32_t)K - 1u : m;
#pragma unroll
              for (int k = K - 1; k >= 1; k--)
                if ((uint32_t)k <= limit) lut[k] = lut[k - 1];
              lut[0] = sym;
            }
            hl = 2u + (m == (uint32_t)K ? SU : 0u) + (cc != c7 ? (cc <= 0xFFFFu ? 2u : 4u) : 0u) + (range != r7 ? (range <= 0xFFFFu ? 2u : 4u) : 0u);
            tag = m;
          }
          else
          {
            const uint32_t range = gap + 1u;
            const bool same = TR::kPacked && sym == la;
            bool shortOk;
            if constexpr (!TR::kPacked) shortOk = range <= TR::MAXRANGE && count >= TR::SHORT;
            else shortOk = range <= TR::MAXRANGE && ((count >= TR::SHORT && same) || count >= TR::MEDIUM);
            stored = shortOk || count >= TR::LONG;
            if constexpr (TR::kPacked) { if (stored) la = sym; }
            const uint32_t cc = TR::kAligned ? (count / SU - TR::SHORT / SU + 1u) : (count - TR::SHORT + 1u);
            if constexpr (!TR::kPacked) hl = SU + (cc <= 255u ? 1u : 5u);
            else hl = (cc <= 127u ? 1u : 5u) + (same ? 0u : SU);
            if constexpr (TR::kRange7) hl += shortOk ? 1u : 4u; else hl += shortOk ? 1u : 5u;
            tag = (same ? 1u : 0u) | (shortOk ? 2u : 0u);
          }
          if (stored)
          {
            iv = opos | (gap << 14) | (1u << 27) | (tag << 28);
            cand[c] = p | (e << 16);
            csym[c] = (sym_t)sym;
            opos += hl + gap;
            lastRLE = e;
          }
        }
        info[c] = iv;
      }
    }
    const bool ended = walker && lastRLE >= n;                            // a stored run reached the end: the end terminator, no literals
    const uint32_t termOff = opos;
    // terminators: Short 3 + 2 + (2 | 4) + (one zero byte | a zero symbol, 0-symbol codecs only); LUT 2 + 2 + (2 | 4); plain S + 1 + 4 + 1 + 4; Packed 1 + 4 + (4 | 1 + 4)
    const uint32_t termLen = TR::kShort ? (ended ? 7u + (K == 0 ? 1u : 0u) : 9u + (K == 0 ? SU : 0u))
                                        : (TR::kLut ? (ended ? 6u : 8u) : ((TR::kPacked ? 5u : SU + 5u) + (TR::kRange7 ? 4u : 5u)));
    const uint32_t kLit = (!walker || ended) ? 0u : n - lastRLE;
    const uint32_t size = termOff + termLen + kLit;
    const uint32_t endedI = ended ? 1u : 0u;
    if (walker) sizes[myBlock] = size;
    wave_sync();
    HS_RLSTAMP(1)

    // C: as in k_encode8_runlist
    struct Stage
    {
      bool on, lng, tail; uint32_t off, hn, d, gap, head; uint64_t w0, w1, w2; const uint8_t *src; u32x4 v0, v1, v2, v3;
    };
    auto stage1 = [&](Stage &st, uint32_t j, uint32_t cb, bool live) __attribute__((always_inline)) {
      const uint32_t b = b0 + j;
      const uint64_t blockAt = (uint64_t)b * B;
      const uint32_t c1 = bOff[j + 1u], c = cb + lane;
      st.on = false; st.lng = false; st.tail = false; st.off = 0; st.hn = 0; st.d = 0; st.gap = 0; st.w0 = 0; st.w1 = 0; st.w2 = 0; st.src = in;
      // (values of the lane that walked this block; read here, where all lanes are active)
      const uint32_t jTerm = (uint32_t)__builtin_amdgcn_readlane((int)termOff, (int)j), jLit = (uint32_t)__builtin_amdgcn_readlane((int)kLit, (int)j);
      const uint32_t jLast = (uint32_t)__builtin_amdgcn_readlane((int)lastRLE, (int)j), jTermLen = (uint32_t)__builtin_amdgcn_readlane((int)termLen, (int)j);
      const bool jEnded = __builtin_amdgcn_readlane((int)endedI, (int)j) != 0;
      if (live && c < c1)
      {
        const uint32_t iv = info[c];
        if ((iv >> 27) & 1u)
        {
          const uint32_t cv = cand[c], p = cv & 0xFFFFu, e = cv >> 16, count = e - p;
          const uint64_t sym = csym[c];
          const uint32_t off = iv & 0x3FFFu, gap = (iv >> 14) & 0x1FFFu, tag = iv >> 28;
          Hdr h;
          if constexpr (TR::kShort)
          {
            const ShortF f = short_fields(count, gap);
            const uint32_t scu = (uint32_t)f.sc;
            const uint32_t mi = (K > 0) ? tag << (TR::SCB + TR::SRBP) : 0u;
            if (f.pack1)
              h.b8(mi | ((uint32_t)(f.sc - 2) << TR::SRBP) | gap);
            else
            {
              h.b8(mi | (TR::SCINV << TR::SRBP) | ((f.scx << (TR::SRB - 8u)) >> 8));
              h.b8((f.scx << (TR::SRB - 8u)) | (f.rx >> 8));
              h.b8(f.rx);
              if (f.scx != scu) { if (scu <= 0xFFFFu) h.b16(scu); else h.b32(scu); }
              if (f.rx != f.range) { if (f.range <= 0xFFFFu) h.b16(f.range); else h.b32(f.range); }
            }
            if (K == 0 || tag == (uint32_t)K) h.push(sym, SU);
          }
          else if constexpr (TR::kLut)
          {
            uint32_t cc, c7, range, r7;
            lut_fields(count, gap, cc, c7, range, r7);
            h.b16((tag << (K == 3 ? 14 : 13)) | (c7 << TR::RB) | r7);
            if (tag == (uint32_t)K) h.push(sym, SU);
            if (cc != c7) { if (cc <= 0xFFFFu) h.b16(cc); else h.b32(cc); }
            if (range != r7) { if (range <= 0xFFFFu) h.b16(range); else h.b32(range); }
          }
          else
          {
            const uint32_t range = gap + 1u;
            const bool same = (tag & 1u) != 0u, shortOk = (tag & 2u) != 0u;
            const uint32_t cc = TR::kAligned ? (count / SU - TR::SHORT / SU + 1u) : (count - TR::SHORT + 1u);
            if constexpr (!TR::kPacked)
            {
              h.push(sym, SU);
              if (cc <= 255u) h.b8(cc); else { h.b8(0); h.b32(cc); }
            }
            else
            {
              const uint32_t sm = same ? 0x80u : 0u;
              if (cc <= 127u) h.b8(cc | sm); else { h.b8(sm); h.b32(cc); }
              if (!same) h.push(sym, SU);
            }
            if constexpr (TR::kRange7) { if (shortOk) h.b8(range << 1); else h.b32((range << 1) | 1u); }
            else { if (shortOk) h.b8(range); else { h.b8(0); h.b32(range); } }
          }
          st.on = true; st.off = off; st.hn = h.n; st.w0 = h.w0; st.w1 = h.w1; st.w2 = h.w2; st.gap = gap;
          st.src = in + blockAt + p - gap;
          st.d = off + h.n;
        }
      }
      else if (live && c == c1)
      {
        // the block's last item: terminator + the literals behind the last stored run
        Hdr h;
        if constexpr (TR::kShort)
        {
          // rleX_Xsl_short.h:470-523: the three-byte form with count field 0 / 1, then u16 0 + u16 0 + ONE zero byte (end) or u16 0 + u32 literals + 2 +
          // a whole zero symbol (literals) -- the zero byte / symbol only in the codecs without a list
          h.b8(TR::SCINV << TR::SRBP); h.b8(TR::STB);
          if (jEnded) { h.b8(1); h.b16(0); h.b16(0); if (K == 0) h.b8(0); }
          else { h.b8(0); h.b16(0); h.b32(jLit + 2u); if (K == 0) h.push(0ull, SU); }
        }
        else if constexpr (TR::kLut)
        {
          // rleX_Xsl_multibyte_encoder.h:329-370: end (1 << RB) | 1, 0, 0 (u16 each); literals 1 << RB, 0 (u16), literals + 2 (u32)
          if (jEnded) { h.b16((1u << TR::RB) | 1u); h.b16(0); h.b16(0); }
          else { h.b16(1u << TR::RB); h.b16(0); h.b32(jLit + 2u); }
        }
        else
        {
          // rleX_extreme_cpu_encode.h:373-609
          if constexpr (!TR::kPacked) { h.push(0ull, SU); h.b8(0); h.b32(0); } else { h.b8(0x80); h.b32(0); }
          if (jEnded) { if constexpr (TR::kRange7) h.b32(1); else { h.b8(0); h.b32(0); } }
          else { if constexpr (TR::kRange7) h.b32(((jLit + 1u) << 1) | 1u); else { h.b8(0); h.b32(jLit + 1u); } }
        }
        st.on = true; st.tail = true; st.off = jTerm; st.hn = jTermLen; st.w0 = h.w0; st.w1 = h.w1; st.w2 = h.w2; st.gap = jLit;
        st.src = in + blockAt + jLast;
        st.d = jTerm + jTermLen;
      }
      st.lng = st.on && st.gap > 64u;
      // the loads may run up to 67 bytes past the literals: not in the last blocks of the input (those copy in stage 2)
      const bool fast = blockAt + (uint64_t)B + 80ull <= U;
      const bool ld = st.on && !st.lng && st.gap != 0u && fast;
      const uint8_t *const a = ld ? st.src : in;
      uint32_t h = (4u - (st.d & 3u)) & 3u;
      if (h > st.gap) h = st.gap;
      const uint8_t *const q = a + (ld ? h : 0u);
      st.head = ld32(a);
      st.v0 = ld128(q); st.v1 = ld128(q + 16u); st.v2 = ld128(q + 32u); st.v3 = ld128(q + 48u);
    };
    auto stage2 = [&](Stage &st, uint32_t j, uint32_t cb) __attribute__((always_inline)) {
      const uint32_t b = b0 + j;
      const uint64_t blockAt = (uint64_t)b * B;
      const bool fast = blockAt + (uint64_t)B + 80ull <= U;
      const uint32_t jSize = (uint32_t)__builtin_amdgcn_readlane((int)size, (int)j);
      const uint32_t jn = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)j);
      if (st.on)
      {
        tile_hdr(st.off, st.w0, st.w1, st.w2, st.hn);
        if (st.tail) { ((uint32_t *)otile)[0] = jn; ((uint32_t *)otile)[1] = jSize; }
        if (!st.lng && st.gap != 0u)
        {
          if (fast)
          {
            uint32_t h = (4u - (st.d & 3u)) & 3u;
            if (h > st.gap) h = st.gap;
            const uint32_t rem = st.gap - h, full = rem >> 4, r = rem & 15u, q = st.d + h;
            tile_bytes(st.d, st.head, h);
            if (full > 0u) tile_dwords(q, st.v0, 4u);
            if (full > 1u) tile_dwords(q + 16u, st.v1, 4u);
            if (full > 2u) tile_dwords(q + 32u, st.v2, 4u);
            if (full > 3u) tile_dwords(q + 48u, st.v3, 4u);
            u32x4 vt = st.v3;                                               // (not a ?: chain of the members: that is an indexed access and puts the set in scratch memory)
            if (full == 0u) vt = st.v0; else if (full == 1u) vt = st.v1; else if (full == 2u) vt = st.v2;
            const uint32_t nd = r >> 2, qt = q + 16u * full;
            tile_dwords(qt, vt, nd);
            tile_bytes(qt + 4u * nd, nd == 0u ? vt.x : (nd == 1u ? vt.y : (nd == 2u ? vt.z : vt.w)), r & 3u);
          }
          else
            lane_copy(st.src, st.d, st.gap);
        }
      }
      // literals of more than 64 bytes: the whole wave, one after the other
      uint64_t todo = __ballot(st.lng);
      while (todo != 0ull)
      {
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint64_t sp = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)st.src, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)st.src >> 32), l) << 32);
        wave_copy((const uint8_t *)sp, (uint32_t)__builtin_amdgcn_readlane((int)st.d, l), (uint32_t)__builtin_amdgcn_readlane((int)st.gap, l));
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
    HS_RLSTAMP(2)
  };

  // ---- phase A over the wave's blocks ----
  uint32_t batchFirst = wgFirst, used = 0;
  Win nx = load_block(wgFirst), nx2 = nx;
  if (wgFirst + 1u < wgLast) nx2 = load_block(wgFirst + 1u);
  for (uint32_t b = wgFirst; b < wgLast; b++)
  {
    const Win w = nx;
    nx = nx2;
    if (b + 2u < wgLast) nx2 = load_block(b + 2u);                       // two blocks ahead
    const uint32_t n = block_len(b);
    const uint32_t myAt = lane * 64u;
    lds_st128(tile + myAt, w.a); lds_st128(tile + myAt + 16u, w.b); lds_st128(tile + myAt + 32u, w.c); lds_st128(tile + myAt + 48u, w.d);   // (for the symbol lookups)

    // match bits: bit i = d[myAt + i] == d[myAt + i + S], only where both lie in the block
    uint32_t x[19];
    x[0] = w.a.x; x[1] = w.a.y; x[2] = w.a.z; x[3] = w.a.w; x[4] = w.b.x; x[5] = w.b.y; x[6] = w.b.z; x[7] = w.b.w;
    x[8] = w.c.x; x[9] = w.c.y; x[10] = w.c.z; x[11] = w.c.w; x[12] = w.d.x; x[13] = w.d.y; x[14] = w.d.z; x[15] = w.d.w;
    x[16] = (uint32_t)__shfl_down((int)w.a.x, 1, 64); x[17] = (uint32_t)__shfl_down((int)w.a.y, 1, 64);   // (lane 63: masked below)
    x[18] = 0u;
    constexpr int QD = S / 4, RD = S % 4;
    uint32_t t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = x[k] ^ (RD ? alignbyte(x[k + QD + 1], x[k + QD], RD) : x[k + QD]);
    const uint32_t m0 = zero_mask16(t[0], t[1], t[2], t[3]), m1 = zero_mask16(t[4], t[5], t[6], t[7]);
    const uint32_t m2 = zero_mask16(t[8], t[9], t[10], t[11]), m3 = zero_mask16(t[12], t[13], t[14], t[15]);
    uint64_t e64 = (uint64_t)(m0 | (m1 << 16)) | ((uint64_t)(m2 | (m3 << 16)) << 32);
    const uint32_t inBlock = (n > myAt) ? umin(64u, n - myAt) : 0u;
    const uint32_t validBits = (n > myAt + SU) ? umin(64u, n - SU - myAt) : 0u;
    e64 &= (validBits >= 64u) ? ~0ull : ((1ull << validBits) - 1ull);
    const uint64_t wmask = (inBlock >= 64u) ? ~0ull : ((1ull << inBlock) - 1ull);

    uint64_t pe = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(e64 >> 32), 1, 64) << 32;     // only the top bits of the lane below are needed
    if (lane == 0u) pe = 0ull;
    const uint64_t h1 = (e64 << 1) | (pe >> 63);
    uint64_t longEnough = h1;                                            // S set bits in front of the position
#pragma unroll
    for (uint32_t k = 2; k <= SU; k++) longEnough &= (e64 << k) | (pe >> (64u - k));
    const uint64_t ustarts = e64 & ~h1;                                  // a stretch of match bits begins
    const uint64_t fends = ~e64 & longEnough & wmask;                    // the first clear position behind a stretch of at least S bits

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
      const uint32_t s = (sBelow != 0ull) ? myAt + 63u - (uint32_t)__builtin_clzll(sBelow) : carryStart1 - 1u;
      cand[used + rank] = s | ((myAt + i) << 16);
      // the S bytes at s (the tile is read in aligned dwords; a stretch's bytes lie inside the block)
      const uint32_t *const tw = (const uint32_t *)(tile + (s & ~3u));
      const uint32_t r = s & 3u, d0 = tw[0], d1 = tw[1], d2 = ((s & ~3u) + 12u <= 4096u) ? tw[2] : 0u;
      const uint32_t lo = r ? alignbyte(d1, d0, r) : d0, hi = r ? alignbyte(d2, d1, r) : d1;
      csym[used + rank] = (sym_t)(((uint64_t)lo | ((uint64_t)hi << 32)) & SYMMASK);
      rank++;
    }
    wave_sync();
    used += total;
  }
  if (lane == 0u) bOff[wgLast - batchFirst] = used;
  flush(batchFirst, wgLast);
#ifdef HSRLE_RL_STAMPS
  if (lane == 0u) { for (int q = 0; q < 3; q++) atomicAdd(g_rl_stamps + q, rlst[q]); atomicAdd(g_rl_stamps + 3, 1ull); atomicAdd(g_rl_stamps + 4, rlst[4]); atomicAdd(g_rl_stamps + 5, rlst[5]); }
#endif
}

} // namespace hsrle
