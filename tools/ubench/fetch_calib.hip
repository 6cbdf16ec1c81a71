// fetch_calib.hip -- what does FETCH_SIZE report for the DECODER'S read-request mix?  
//
// The decoder's top-up (hsrle_decode.hip.h) reads every 16-byte chunk of the container exactly once: the 8 lanes that serve a row hold
// the 8 chunks of one aligned 128-byte piece of that row's stream and re-request "their" chunk 128 bytes further on whenever it has
// landed in the ring -- so ONE load instruction asks for a SUBSET of the 8 chunks of a line, and the rest of the line is asked for by
// later instructions of the same wave.  rocprofv3's FETCH_SIZE is derived from L2 -> fabric request counts (TCC_EA0_RDREQ x 64 B with
// 32-byte requests at 32; MI355X_MICROARCH.md: a 128-byte request is tallied at 64), so what it says for this mix has to be measured on
// a buffer whose size is known.  This program replays the mix over `bytes` of device memory, every byte exactly once:
//   mode 0   whole lines: all 8 lanes of a group load in the same instruction            (the guide's calibrated case: reports 1/2)
//   mode 1   half lines: chunks 0-3 in one instruction, chunks 4-7 in a later one
//   mode 2   single chunks: 8 instructions per line, one chunk each
//   mode 3   the decoder's mix: per line a pseudo-random split into 1..3 instructions (subset sizes like a ring that frees 16..128 B)
// Run it under `rocprofv3 --pmc FETCH_SIZE` (and, separately, TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum): tools/fetch_calib.sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kRowBytes = 2304;            // one compressed 4 KiB block of the 50 %-runs buffer: 18 lines
constexpr uint32_t kLines = kRowBytes / 128;

__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

template <int MODE>
__global__ __launch_bounds__(64) void k_replay(const uint8_t *__restrict__ buf, uint64_t rows, uint32_t *__restrict__ sink)
{
  const uint32_t lane = threadIdx.x, c = lane & 7u, g = lane >> 3;
  u32x4 acc = { 0, 0, 0, 0 };
  // like the decoder: in instruction q the 8 lanes of group g serve row q * 8 + g of this workgroup's 64 rows
  for (uint32_t line = 0; line < kLines; line++)
  {
    constexpr int PASSES = MODE == 0 ? 1 : (MODE == 1 ? 2 : (MODE == 2 ? 8 : 3));
#pragma unroll 1
    for (int pass = 0; pass < PASSES; pass++)
    {
#pragma unroll
      for (uint32_t q = 0; q < 8; q++)
      {
        const uint64_t row = (uint64_t)blockIdx.x * 64u + q * 8u + g;
        bool mine;
        if (MODE == 0) mine = true;
        else if (MODE == 1) mine = (c >> 2) == (uint32_t)pass;
        else if (MODE == 2) mine = c == (uint32_t)pass;
        else
        {
          // two cut points per line -> up to three contiguous chunk groups, requested by three different instructions
          const uint32_t h = mix((uint32_t)row * 31u + line);
          const uint32_t a = h & 7u, b = (h >> 3) & 7u;
          const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
          const uint32_t part = c < lo ? 0u : (c < hi ? 1u : 2u);
          mine = part == (uint32_t)pass;
        }
        if (mine && row < rows)
        {
          const u32x4 v = *(const u32x4 *)(buf + row * kRowBytes + line * 128u + c * 16u);
          acc ^= v;
        }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;   // keeps the loads alive
}

int main(int argc, char **argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const uint64_t bytes = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 4096ull) << 20;
  const uint64_t rows = bytes / kRowBytes;
  uint8_t *buf; uint32_t *sink;
  if (hipMalloc(&buf, rows * kRowBytes + 256) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(buf, 0x5A, rows * kRowBytes + 256);
  hipMemset(sink, 0, 64);
  const uint32_t grid = (uint32_t)((rows + 63) / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++)
  {
    hipEventRecord(e0);
    switch (mode)
    {
    case 0: hipLaunchKernelGGL(k_replay<0>, dim3(grid), dim3(64), 0, 0, buf, rows, sink); break;
    case 1: hipLaunchKernelGGL(k_replay<1>, dim3(grid), dim3(64), 0, 0, buf, rows, sink); break;
    case 2: hipLaunchKernelGGL(k_replay<2>, dim3(grid), dim3(64), 0, 0, buf, rows, sink); break;
    default: hipLaunchKernelGGL(k_replay<3>, dim3(grid), dim3(64), 0, 0, buf, rows, sink); break;
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("mode %d bytes %llu rows %llu  %.3f ms  %.1f GB/s\n", mode, (unsigned long long)(rows * kRowBytes), (unsigned long long)rows, ms, rows * kRowBytes / (ms * 1e-3) / 1e9);
  return 0;
}
