// Microbenchmark: cost of per-lane 16-byte LDS accesses at row stride RS with aligned / unaligned / random offsets.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int RS, int MODE, int OP, int W>   // MODE 0: off=16k aligned, 1: off = 16k + lane%16 (unaligned), 2: per-lane pseudo-random byte offsets; OP 0 read 1 write 2 read+write
__global__ __launch_bounds__(64 * W) void k(uint32_t *out, int iters)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * W * RS + 64];
  const uint32_t lane = threadIdx.x;
  uint8_t *row = lds + lane * RS;
  for (int i = lane; i < (64 * W * RS) / 4; i += 64 * W) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t rnd = lane * 2654435761u + 12345u;
  u32x4 acc = { 0, 0, 0, 0 };
  long long t0 = clock64();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      uint32_t off;
      if (MODE == 0) off = ((it * 8 + u) * 16) % (RS - 32);
      else if (MODE == 1) off = (((it * 8 + u) * 16) % (RS - 48)) + (lane % 16);
      else { rnd = rnd * 1664525u + 1013904223u; off = (rnd >> 8) % (RS - 32); }
      if (OP == 0 || OP == 2) { u32x4 v; __builtin_memcpy(&v, row + off, 16); acc += v; }
      if (OP == 1) { u32x4 v = { off, off, off, off }; __builtin_memcpy(row + off, &v, 16); }
      if (OP == 2) { u32x4 v = acc; __builtin_memcpy(row + ((off + 37) % (RS - 32)), &v, 16); }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (lane == 0) { out[blockIdx.x * 2] = (uint32_t)(t1 - t0); out[blockIdx.x * 2 + 1] = acc.x + acc.y + acc.z + acc.w + lds[5]; }
}
template <int RS, int MODE, int OP, int W>
void run(const char *name, uint32_t *d, int blocksPerCU)
{
  const int iters = 256; int grid = 256 * blocksPerCU;
  hipLaunchKernelGGL((k<RS, MODE, OP, W>), dim3(grid), dim3(64 * W), 0, 0, d, iters); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL((k<RS, MODE, OP, W>), dim3(grid), dim3(64 * W), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  uint32_t h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  double instrs = (double)iters * 8 * (OP == 2 ? 2 : 1);
  printf("%-44s waves/CU %2d: %7.1f cycles per LDS instr per wave | chip %7.1f TB/s\n", name, blocksPerCU * W, h[0] / instrs, (double)grid * W * instrs * 1024 / ms / 1e9);
}
int main()
{
  uint32_t *d; hipMalloc(&d, 1 << 20);
  for (int bpc : { 1, 4, 8 })
  {
    run<272, 0, 0, 1>("read  b128 aligned   RS=272", d, bpc);
    run<272, 1, 0, 1>("read  b128 unaligned RS=272 (lane%16)", d, bpc);
    run<272, 2, 0, 1>("read  b128 random    RS=272", d, bpc);
    run<144, 0, 1, 1>("write b128 aligned   RS=144", d, bpc);
    run<144, 1, 1, 1>("write b128 unaligned RS=144 (lane%16)", d, bpc);
    run<144, 2, 1, 1>("write b128 random    RS=144", d, bpc);
    run<272, 2, 2, 1>("read+write random    RS=272", d, bpc);
  }
  return 0;
}
