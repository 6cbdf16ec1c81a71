// Microbenchmark: does an LDS b128 access cost less when only some lanes are active (divergent code)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int RS, int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, unsigned long long mask)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * RS + 64];
  const uint32_t lane = threadIdx.x;
  uint8_t *row = lds + lane * RS;
  for (int i = lane; i < (64 * RS) / 4; i += 64) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t rnd = lane * 2654435761u + 12345u; u32x4 acc = { 0, 0, 0, 0 };
  if ((mask >> lane) & 1ull)
  {
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        rnd = rnd * 1664525u + 1013904223u;
        uint32_t off = ((rnd >> 8) % (RS - 32)) & ~3u;
        if (OP == 0) { acc += *(const u32x4 *)__builtin_assume_aligned(row + off, 16); }
        else { u32x4 v = { off, off, off, off }; *(u32x4 *)__builtin_assume_aligned(row + off, 16) = v; }
      }
    }
  }
  __syncthreads();
  if (lane == 0) out[blockIdx.x] = acc.x + acc.y + acc.z + acc.w + lds[5];
}
template <int RS, int OP>
void run(uint32_t *d, unsigned long long mask, const char *what)
{
  const int iters = 256; int grid = 256 * 8; float ms; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RS, OP>), dim3(grid), dim3(64), 0, 0, d, iters, mask); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<RS, OP>), dim3(grid), dim3(64), 0, 0, d, iters, mask); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  double instrs = (double)iters * 8 * grid;
  printf("%s b128 %-28s: %6.1f CU-cycles per wave-instr\n", OP ? "write" : "read ", what, ms * 1e-3 * 2.4e9 * 256 / instrs);
}
int main()
{
  uint32_t *d; hipMalloc(&d, 1 << 20);
  struct { unsigned long long m; const char *n; } ms[] = { { ~0ull, "64 lanes" }, { 0xFFFFFFFFull, "lanes 0-31" }, { 0xFFFFull, "lanes 0-15" }, { 0xFFull, "lanes 0-7" },
    { 0x0101010101010101ull, "every 8th lane (8)" }, { 0x1111111111111111ull, "every 4th lane (16)" }, { 1ull, "1 lane" } };
  for (auto &m : ms) { run<292, 0>(d, m.m, m.n); run<148, 1>(d, m.m, m.n); }
  return 0;
}
