// hsrle_expand.hip.h -- second half of the packet-list decode of small containers (hsrle_index.hip.h: k_container_packets writes the lists).
//
// Replaces: the copy / fill halves of the reference's decode loops -- src/rle8_extreme_cpu.h:1825-1913 (MEMCPY of the literals, MEMSET of the
//           run), src/rleX_extreme_cpu_decode.h:129-162, src/rleX_Xsl.h:580-760, src/rle128_extreme_cpu.h:600-802 -- for a block whose packet
//           boundaries are already known.  One kernel per SYMBOL WIDTH: the entries are the same for every family.
//
// One workgroup (one wave) per block.  The block's list goes into LDS; then every lane builds 16 OUTPUT bytes at a time: binary search for
// the packet that covers the chunk's first byte, then packet by packet -- literal bytes with one unaligned 16-byte load from the stream
// (placed so that the byte for output position p lands in the chunk's byte p & 15: no shifting), run bytes from the symbol pattern
// rotated to the chunk's phase -- until the chunk is full.  A wave stores 1 KiB of contiguous output per trip.  Nothing here depends on
// what another lane does: the chain of the format was paid for once, in the walk.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h"   // funnel16, merge_low, wave_sync
#include "hsrle_index.hip.h"    // kPktInitField, packet_list_cap

namespace hsrle {

// 32 bytes of the run's pattern from phase 0: P0 = bytes 0..15, P1 = bytes 16..31 of sym sym sym ...
template <int S>
__device__ __forceinline__ void run_pattern32(u32x4 sv, u32x4 &P0, u32x4 &P1)
{
  if constexpr (S == 1) { const uint32_t v = (sv.x & 0xFFu) * 0x01010101u; P0 = u32x4{ v, v, v, v }; P1 = P0; }
  else if constexpr (S == 2) { const uint32_t v = (sv.x & 0xFFFFu) * 0x00010001u; P0 = u32x4{ v, v, v, v }; P1 = P0; }
  else if constexpr (S == 4) { P0 = u32x4{ sv.x, sv.x, sv.x, sv.x }; P1 = P0; }
  else if constexpr (S == 8) { P0 = u32x4{ sv.x, sv.y, sv.x, sv.y }; P1 = P0; }
  else if constexpr (S == 16) { P0 = sv; P1 = sv; }
  else if constexpr (S == 3)
  {
    const uint32_t s = sv.x & 0xFFFFFFu;
    const uint32_t u0 = s | (s << 24), u1 = (s >> 8) | (s << 16), u2 = (s >> 16) | (s << 8);    // s0 s1 s2 s0 | s1 s2 s0 s1 | s2 s0 s1 s2
    P0 = u32x4{ u0, u1, u2, u0 }; P1 = u32x4{ u1, u2, u0, u1 };
  }
  else
  {
    static_assert(S == 6, "symbols of 1, 2, 3, 4, 6, 8 or 16 bytes");
    const uint32_t a = sv.x, b = sv.y & 0xFFFFu;
    const uint32_t u0 = a, u1 = b | (a << 16), u2 = (a >> 16) | (b << 16);                      // s0 s1 s2 s3 | s4 s5 s0 s1 | s2 s3 s4 s5
    P0 = u32x4{ u0, u1, u2, u0 }; P1 = u32x4{ u1, u2, u0, u1 };
  }
}

template <int S>
__global__ __launch_bounds__(64) void k_expand_packets(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, uint8_t *__restrict__ out, uint64_t U, uint32_t B,
                                                       uint32_t firstBlock, uint32_t blockCount, const uint64_t *__restrict__ lists, const uint32_t *__restrict__ counts)
{
  extern __shared__ __attribute__((aligned(16))) uint64_t ent[];
  const uint32_t lane = threadIdx.x;
  const uint32_t local = xcd_tile(blockIdx.x, gridDim.x);
  if (local >= blockCount) return;
  const uint32_t n = counts[local];
  if (n < 2u) return;                                                   // (0: the walk refused the block; 1: only a sentinel)
  const uint32_t cap = packet_list_cap(B);
  const uint64_t *const list = lists + (uint64_t)local * cap;
  for (uint32_t i = lane; i < n; i += 64u) ent[i] = list[i];
  const uint32_t b = firstBlock + local;
  const uint8_t *const strm = payload + offsets[b];
  uint8_t *const dst = out + (uint64_t)b * B;
  wave_sync();

  const uint32_t endOut = (uint32_t)ent[n - 1u] & 0x7FFFu;
  for (uint32_t o = lane * 16u; o < endOut; o += 1024u)
  {
    // the last real packet (index < n - 1) that starts at or below o
    uint32_t lo = 0u, hi = n - 1u;
    while (hi - lo > 1u)
    {
      const uint32_t mid = (lo + hi) >> 1;
      if (((uint32_t)ent[mid] & 0x7FFFu) <= o) lo = mid; else hi = mid;
    }
    uint32_t i = lo, pos = o, d = 0u;
    const uint32_t chunkEnd = umin(o + 16u, endOut);
    u32x4 acc = u32x4{ 0, 0, 0, 0 };
    for (;;)
    {
      const uint64_t e = ent[i];
      const uint32_t ps = (uint32_t)e & 0x7FFFu, lit = (uint32_t)(e >> 15) & 0x7FFFu, body = (uint32_t)(e >> 30) & 0x7FFFu, symf = (uint32_t)(e >> 45) & 0x7FFFu;
      const uint32_t pe = (uint32_t)ent[i + 1u] & 0x7FFFu, litEnd = ps + lit;
      if (pos < litEnd)
      {
        // the stream byte of output position pos belongs into chunk byte d: window from d bytes in front of it (those bytes are replaced
        // by what acc holds; they exist: a block's stream is preceded by the container's header and offset table at least)
        const u32x4 v = ld128(strm + body + (pos - ps) - d);
        acc = (d != 0u) ? merge_low(acc, v, d) : v;
        pos = umin(litEnd, chunkEnd);
        d = pos - o;
      }
      if (pos < chunkEnd && pos < pe)
      {
        u32x4 sv;
        if (symf >= kPktInitField) { const uint32_t v = (symf & 0xFFu) * 0x01010101u; sv = u32x4{ v, v, v, v }; }
        else sv = ld128(strm + symf);
        u32x4 P0, P1;
        run_pattern32<S>(sv, P0, P1);
        // chunk byte j holds run byte (o + j - litEnd) mod S
        const uint32_t ph = (o >= litEnd) ? (o - litEnd) % (uint32_t)S : ((uint32_t)S - (litEnd - o) % (uint32_t)S) % (uint32_t)S;
        const u32x4 pat = (ph != 0u) ? funnel16(P0, P1, ph) : P0;
        acc = (d != 0u) ? merge_low(acc, pat, d) : pat;
        pos = umin(pe, chunkEnd);
        d = pos - o;
      }
      if (pos >= chunkEnd) break;
      i++;
    }
    if (o + 16u <= endOut)
      __builtin_nontemporal_store(acc, (u32x4 *)(dst + o));
    else
    {
      const uint32_t w[4] = { acc.x, acc.y, acc.z, acc.w };
      for (uint32_t j = 0; j < endOut - o; j++) dst[o + j] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
    }
  }
}

template <int S>
inline hipError_t launch_expand_packets_s(const DecodeArgs &a, const uint64_t *lists, const uint32_t *counts, hipStream_t st)
{
  const uint32_t lds = packet_list_cap(a.B) * 8u;
  hipLaunchKernelGGL((k_expand_packets<S>), dim3(a.blockCount), dim3(64), lds, st, a.payload, a.offsets, a.out, a.U, a.B, a.firstBlock, a.blockCount, lists, counts);
  return hipGetLastError();
}

inline hipError_t launch_expand_packets(int S, const DecodeArgs &a, const uint64_t *lists, const uint32_t *counts, hipStream_t st)
{
  switch (S)
  {
  case 1: return launch_expand_packets_s<1>(a, lists, counts, st);
  case 2: return launch_expand_packets_s<2>(a, lists, counts, st);
  case 3: return launch_expand_packets_s<3>(a, lists, counts, st);
  case 4: return launch_expand_packets_s<4>(a, lists, counts, st);
  case 6: return launch_expand_packets_s<6>(a, lists, counts, st);
  case 8: return launch_expand_packets_s<8>(a, lists, counts, st);
  case 16: return launch_expand_packets_s<16>(a, lists, counts, st);
  default: return hipErrorInvalidValue;
  }
}

} // namespace hsrle
