// experiments/hsrle_decode_wave.hip.h -- NOT PART OF THE SHIPPED LIBRARY (-DHSRLE_EXPERIMENTS builds only: measured slower than the split
// decode).  One WAVE per block: the decoder for containers with too few blocks to fill the chip with one lane each
// (BASELINE config 3: the 88 MB frame in 4 KiB blocks is 21 600 blocks = 338 waves of the block kernel, 1.3 per CU).
//
// Replaces, for such containers, the same reference loops as k_decode_blocks (src/rle8_extreme_cpu.h:1546-2434,
// src/rleX_extreme_cpu_decode.h:27-164, src/rleX_Xsl.h:530-1844, src/rleX_Xsl_short.h:1255-1700); SURVEY.md 8f-3 / VERDICT r01 item 2.
//
//   1. the block's stream -> LDS (coalesced 16-byte loads)
//   2. lane 0 hops through the packets (parse_packet + the symbol state of hsrle_index.hip.h: the chain is serial, but a hop from LDS
//      is ~100 instructions) and leaves a descriptor per packet: output position, literal source, literal / run bytes, fill pattern
//   3. all lanes: output-centric expansion.  Lane l takes the 16-byte output chunks l, l + 64, ...: binary search for the packet that
//      holds the chunk's first byte, then the pieces of the packets that overlap the chunk -- literal bytes by an unaligned LDS read,
//      run bytes as the pattern rotated to the chunk's phase -- are merged under byte masks and stored once, whole and aligned.
// Blocks with more packets than the descriptor table holds go through 2 / 3 in batches; the descriptors that overlap the last,
// incomplete chunk of a batch are carried into the next one.  Nothing outside [0, uncompressedSize) is written.
#pragma once

#include "../hsrle_common.hip.h"
#include "../hsrle_decode.hip.h"
#include "../hsrle_index.hip.h"

namespace hsrle {

constexpr uint32_t kWaveDecodeDescriptors = 256;     // per batch (32 bytes each)
constexpr uint32_t kWaveDecodeMaxBlock = 16384;      // larger blocks: the block kernel (with the split decode, hsrle_index.hip.h)

__host__ __device__ inline uint32_t wave_decode_stream_cap(uint32_t B) { return ((B + (B >> 6) + 64u + 15u) & ~15u) + 96u; }
__host__ __device__ inline uint32_t wave_decode_lds_bytes(uint32_t B) { return 16u + wave_decode_stream_cap(B) + 17u * 16u + kWaveDecodeDescriptors * 32u + 32u; }

template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_decode_wave_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ payloadEnd,
                                                           uint8_t *__restrict__ out, uint64_t U, uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t allowSingle,
                                                           uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int KE = IndexState<FAM>::KE;
  constexpr int KS = KE > 0 ? KE : 1;
  constexpr uint32_t D = kWaveDecodeDescriptors;
  extern __shared__ __attribute__((aligned(16))) uint8_t wlds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t i = xcd_tile(blockIdx.x, gridDim.x);
  if (i >= blockCount) return;
  const uint32_t b = firstBlock + i;
  const uint32_t cap = wave_decode_stream_cap(B);
  uint8_t *const sbuf = wlds;                                            // [16 pad | stream copy]
  u32x4 *const lowm = (u32x4 *)(wlds + 16u + cap);                       // lowm[k]: the low k bytes set (k = 0 .. 16)
  u32x4 *const desc = (u32x4 *)(wlds + 16u + cap + 17u * 16u);           // [2 k] = { out, literal source, literal bytes, run bytes }, [2 k + 1] = pattern
  volatile uint32_t *const ctl = (volatile uint32_t *)(wlds + 16u + cap + 17u * 16u + D * 32u);   // [0] descriptors, [1] output bytes behind them, [2] done, [3] error

  const uint64_t payloadBytes = (uint64_t)(payloadEnd - payload);
  const uint64_t off0 = offsets[b], off1 = offsets[b + 1];
  const uint64_t start = (uint64_t)b * B;
  const uint32_t blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  uint32_t err = 0;
  if (off0 > off1 || off1 > payloadBytes || off1 - off0 + 32u > (uint64_t)cap - 64u) err = DEC_ERR_HEADER;   // (a stream longer than the bound of its block size is not a stream of this container)
  const uint32_t C = (uint32_t)(off1 - off0);
  const uint8_t *const s = payload + off0;
  uint32_t p0 = TR::kHeaderSize, sgl = 0, singleSym = 0;
  if (err == 0u)
  {
    if (C < TR::kHeaderSize + 2u || ld32(s) != blen || ld32(s + 4) != C) err = DEC_ERR_HEADER;
    else if constexpr (TR::kShortSingle) { singleSym = s[8]; p0 = 9; }
    else if constexpr (S == 1 && !TR::kLut && !TR::kShort)
    {
      const uint32_t mode = s[8];
      if (mode == 1u) { if (allowSingle) { sgl = 1; singleSym = s[9]; p0 = 10; } else err = DEC_ERR_MODE; }
      else if (mode != 0u) err = DEC_ERR_MODE;
    }
  }
  if (err != 0u)
  {
    if (lane == 0u && status != nullptr) atomicOr(status, err);
    return;
  }

  // ---- 1. stream -> LDS; mask table ----
  const uint64_t w0 = off0 & ~15ull;
  const uint32_t delta = 16u + (uint32_t)(off0 - w0);                    // LDS offset of the stream's byte 0
  {
    const uint32_t total = (uint32_t)(off1 - w0) + 64u;                  // (+ what a parse may read behind the last packet)
    for (uint32_t c = lane * 16u; c < total; c += 64u * 16u)
      lds_st128(sbuf + 16u + c, (w0 + c + 16u <= payloadBytes + HSRLE_TAIL_PAD_BYTES) ? ld128(payload + w0 + c) : u32x4{ 0, 0, 0, 0 });
    if (lane == 0u) lds_st128(sbuf, u32x4{ 0, 0, 0, 0 });
    if (lane <= 16u)
    {
      const uint32_t k = lane;
      const uint64_t lo = (k >= 8u) ? ~0ull : ~(~0ull << (8u * k)), hi = (k <= 8u) ? 0ull : ((k >= 16u) ? ~0ull : ~(~0ull << (8u * (k - 8u))));
      lowm[k] = u32x4{ (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32) };
    }
  }
  __syncthreads();

  // ---- walker state (lane 0) ----
  uint32_t x = p0, o = 0;
  uint32_t st[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) st[j] = IDX_INIT | (uint32_t)j;
  uint32_t curSym = IDX_INIT;
  uint32_t carried = 0;                                                  // descriptors kept from the batch before
  uint32_t doneOut = 0;                                                  // output bytes [0, doneOut) are written (multiple of 16 until the end)
  const LdsReader rd{ sbuf, delta };
  uint8_t *const dst = out + start;

  for (uint32_t batch = 0; batch < 4096u; batch++)
  {
    // ---- 2. lane 0: up to D descriptors ----
    if (lane == 0u)
    {
      uint32_t cnt = carried, bad = 0, fin = 0;
      while (cnt < D)
      {
        const Pkt k = parse_packet<FAM, S, AL>(rd, x, C, sgl != 0u);
        if (k.bad) { bad = 1; break; }
        state_apply<KE>(st, k.op, k.symAt);
        if (k.hasSym) curSym = k.symAt;
        u32x4 sym;
        if (sgl != 0u || TR::kShortSingle) sym = u32x4{ singleSym & 0xFFu, 0, 0, 0 };
        else if constexpr (KE > 0) sym = index_symbol<S>(rd, st[0], FAM == PACKED);
        else sym = index_symbol<S>(rd, curSym, true);
        u32x4 pv;
        if constexpr (S == 1) { const uint32_t r = (sym.x & 0xFFu) * 0x01010101u; pv = u32x4{ r, r, r, r }; }
        else pv = make_pattern<S>(sym);
        const uint32_t body = x + k.used;
        if ((uint64_t)o + k.lit + k.run > (uint64_t)blen || (k.lit == 0u && k.run == 0u && !k.last)) { bad = 1; break; }
        desc[2u * cnt] = u32x4{ o, body, k.lit, k.run };
        desc[2u * cnt + 1u] = pv;
        cnt++;
        o += k.lit + k.run;
        x = body + k.lit;
        if (k.last) { fin = 1; break; }
      }
      if (fin && o != blen) bad = 1;
      ctl[0] = cnt; ctl[1] = o; ctl[2] = fin; ctl[3] = bad;
    }
    __syncthreads();
    const uint32_t cnt = ctl[0], oEnd = ctl[1], fin = ctl[2], bad = ctl[3];
    if (bad != 0u)
    {
      if (lane == 0u && status != nullptr) atomicOr(status, (uint32_t)DEC_ERR_STREAM);
      return;
    }

    // ---- 3. all lanes: the chunks [doneOut, limit) ----
    const uint32_t limit = fin ? oEnd : (oEnd & ~15u);
    for (uint32_t cb = doneOut + lane * 16u; cb < limit; cb += 64u * 16u)
    {
      const uint32_t ce = (cb + 16u < limit) ? cb + 16u : limit;
      // the last descriptor whose packet starts at or before cb
      uint32_t lo = 0, hi = cnt;                                         // invariant: desc[lo].out <= cb, desc[hi].out > cb (hi == cnt: none)
      while (hi - lo > 1u)
      {
        const uint32_t mid = (lo + hi) >> 1;
        if (desc[2u * mid].x <= cb) lo = mid; else hi = mid;
      }
      u32x4 acc = u32x4{ 0, 0, 0, 0 };
      uint32_t pos = cb, k = lo;
      while (pos < ce && k < cnt)
      {
        const u32x4 d = desc[2u * k];
        const uint32_t litEnd = d.x + d.z, pkEnd = litEnd + d.w;
        if (pos >= pkEnd) { k++; continue; }
        u32x4 v;
        uint32_t pe;
        if (pos < litEnd)
        {
          pe = (ce < litEnd) ? ce : litEnd;
          v = lds_read16_w8(sbuf, delta + d.y + (cb - d.x));            // stream bytes for the chunk's positions (cb >= d.x - 15: the pad in front covers it)
        }
        else
        {
          pe = (ce < pkEnd) ? ce : pkEnd;
          const u32x4 pv = desc[2u * k + 1u];
          // phase of the pattern at cb: (cb - litEnd) mod S, cb may lie up to 15 bytes in front of the run
          const uint32_t ph = (cb + 16u * (uint32_t)S - litEnd) % (uint32_t)S;
          if constexpr (S == 3 || S == 6)
          {
            uint32_t e0, e1, e2;
            pattern_dwords12(pv, ph, e0, e1, e2);
            v = u32x4{ e0, e1, e2, e0 };
          }
          else
            v = funnel16(pv, pv, ph & 15u);
        }
        const u32x4 ma = lowm[pos - cb], mb = lowm[pe - cb];
        const u32x4 m = u32x4{ mb.x & ~ma.x, mb.y & ~ma.y, mb.z & ~ma.z, mb.w & ~ma.w };
        acc = u32x4{ (acc.x & ~m.x) | (v.x & m.x), (acc.y & ~m.y) | (v.y & m.y), (acc.z & ~m.z) | (v.z & m.z), (acc.w & ~m.w) | (v.w & m.w) };
        pos = pe;
      }
      if (ce - cb == 16u)
        __builtin_nontemporal_store(acc, (u32x4_unaligned *)(dst + cb));
      else
      {
        const uint64_t a0 = (uint64_t)acc.x | ((uint64_t)acc.y << 32), a1 = (uint64_t)acc.z | ((uint64_t)acc.w << 32);
        for (uint32_t t = 0; t < ce - cb; t++)
          dst[cb + t] = (uint8_t)((t < 8u ? a0 >> (8u * t) : a1 >> (8u * (t - 8u))) & 0xFFull);
      }
    }
    if (fin != 0u) return;
    doneOut = limit;
    __syncthreads();
    // carry the descriptors that reach beyond `limit` (the packets of the incomplete chunk: at most 16)
    if (lane == 0u)
    {
      uint32_t kc = cnt;
      while (kc > 0u)
      {
        const u32x4 d = desc[2u * (kc - 1u)];
        if (d.x + d.z + d.w > limit) kc--; else break;
      }
      carried = cnt - kc;
      for (uint32_t t = 0; t < carried; t++)
      {
        const u32x4 a = desc[2u * (kc + t)], c2 = desc[2u * (kc + t) + 1u];
        desc[2u * t] = a; desc[2u * t + 1u] = c2;
      }
    }
    __syncthreads();
  }
  if (lane == 0u && status != nullptr) atomicOr(status, (uint32_t)DEC_ERR_STREAM);
}

template <int FAM, int S, int AL>
inline hipError_t launch_decode_wave(const DecodeArgs &a, uint32_t allowSingle, hipStream_t st)
{
  hipLaunchKernelGGL((k_decode_wave_blocks<FAM, S, AL>), dim3(a.blockCount), dim3(64), wave_decode_lds_bytes(a.B), st, a.payload, a.offsets, a.payloadEnd, a.out, a.U, a.B, a.firstBlock,
                     a.blockCount, allowSingle, a.status);
  return hipGetLastError();
}

} // namespace hsrle
