// hsrle_ring_probe.hip.h -- the history ring of the ring encoders of 1 and 2 byte symbols is chosen per input
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

// Which ring the encoders of 1 and 2 byte symbols should use for THIS input (the wider ones always use 128: hsrle_encodeS.hip.h).
// With 128 bytes the literals in front of a run are fetched from global memory when the run's end lies more than ~100 bytes behind
// their start; that costs more than the extra waves bring when gaps AND runs are long (run-distributed(8): -11 ... -17 %) and nothing
// when the gaps are short (video-shaped: +28 ... +31 %).  The probe samples up to 256 blocks: e = positions whose symbol repeats,
// r = run starts; from the mean gap g = (n - e) / r and mean run length l = e / r + S it estimates the 16-byte fetches per KiB a
// 128-byte ring would add, r * min(1, (g + l) / 128) * ceil(g / 16), and k_ring_decide picks 128 below kRingFetchesPerKiB.
constexpr uint32_t kRingFetchesPerKiB = 13;     // run-distributed(8): ~21, video-shaped: ~8 (S = 1)
template <int S>
__global__ __launch_bounds__(64) void k_ring_probe(const uint8_t *__restrict__ in, uint64_t U, uint32_t B, uint32_t nBlocks, uint32_t *__restrict__ sel)
{
  // sel[1] += sampled positions, sel[2] += repeating positions, sel[3] += run starts
  const uint32_t samples = gridDim.x;
  const uint32_t b = (uint32_t)(((uint64_t)blockIdx.x * nBlocks) / samples);
  const uint64_t at = (uint64_t)b * B;
  const uint32_t n = (uint32_t)((U - at) < (uint64_t)B ? (U - at) : (uint64_t)B);
  const uint32_t span = n < 4096u ? n : 4096u;                            // the first 4 KiB of the block
  uint32_t e = 0, r = 0, cnt = 0;
  for (uint32_t p = threadIdx.x * 16u; p + 24u <= span; p += 64u * 16u)
  {
    const u32x4 x = ld128(in + at + p);
    const uint64_t lo = (uint64_t)x.x | ((uint64_t)x.y << 32), hi = (uint64_t)x.z | ((uint64_t)x.w << 32);
    const uint64_t nx = ld64(in + at + p + 16u);                         // the bytes behind the chunk (S <= 2 of them are needed, + 1 for the start test)
    // byte k of the chunk against byte k + S
    static_assert(S >= 1 && S <= 4, "the bytes behind the chunk come from one 8-byte read");
    const uint64_t slo = (lo >> (8 * S)) | (hi << (64 - 8 * S));
    const uint64_t shi = (hi >> (8 * S)) | (nx << (64 - 8 * S));
    const uint64_t dl = lo ^ slo, dh = hi ^ shi;
    const uint64_t zl = ~(((dl & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | dl) & 0x8080808080808080ull;
    const uint64_t zh = ~(((dh & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | dh) & 0x8080808080808080ull;
    // 0x80 flags -> 16 bits (flag of byte i -> bit i: (8 i + 7) + 7 (7 - i) = 56 + i, no two terms meet).  Until round 4 this was
    // ((z >> 7) * 0x0102040810204081) >> 56, in which bytes 0 and 7 both land on bit 56 and carry into the rest: the probe saw a quarter
    // of the repeats (run-distributed(8), S = 1: 11.6 % instead of 48.6 %) -- the decisions held because the thresholds had been set on
    // what it reported.
    const uint32_t m = (uint32_t)((zl * 0x0002040810204081ull) >> 56) | ((uint32_t)((zh * 0x0002040810204081ull) >> 56) << 8);
    const uint32_t firstBehind = ((uint32_t)(nx & 0xFFull) == (uint32_t)((nx >> (8 * S)) & 0xFFull)) ? 1u : 0u;   // does position 16 repeat?
    const uint32_t m17 = m | (firstBehind << 16);
    e += (uint32_t)__builtin_popcount(m);
    r += (uint32_t)__builtin_popcount(m & ~(m17 >> 1) & 0xFFFFu);         // ends of repeat stretches = one per run
    cnt += 16u;
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1)
  {
    e += (uint32_t)__shfl_xor((int)e, dd, 64); r += (uint32_t)__shfl_xor((int)r, dd, 64); cnt += (uint32_t)__shfl_xor((int)cnt, dd, 64);
  }
  if (threadIdx.x == 0u) { atomicAdd(sel + 1, cnt); atomicAdd(sel + 2, e); atomicAdd(sel + 3, r); }
}
template <int S>
__global__ void k_ring_decide(uint32_t *__restrict__ sel)
{
  if (threadIdx.x != 0u) return;
  const uint64_t n = sel[1], e = sel[2], r = sel[3];
  uint32_t pick = 256u;
  if (n != 0u && r != 0u)
  {
    const uint64_t g = (n - e) / r, l = e / r + (uint64_t)S;               // mean gap, mean run (bytes)
    const uint64_t perKiB = r * 1024u / n;                                // runs per KiB
    const uint64_t frac128 = (g + l >= 128u) ? 128u : g + l;              // share of the runs whose gap has left a 128-byte ring, in 1 / 128
    const uint64_t fetches = perKiB * frac128 * ((g + 15u) / 16u) / 128u;
    if (fetches < kRingFetchesPerKiB) pick = 128u;
  }
  else if (n != 0u) pick = 128u;                                          // no runs at all: nothing is fetched either way, the waves count
  sel[0] = pick;
}

} // namespace hsrle
