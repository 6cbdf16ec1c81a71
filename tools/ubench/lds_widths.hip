// Microbenchmark: per-lane LDS accesses of width 1/4/8/16 bytes, aligned-random vs byte-random offsets, rows at stride RS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int W> struct V; template <> struct V<1> { typedef uint8_t t; }; template <> struct V<4> { typedef uint32_t t; };
template <> struct V<8> { typedef uint64_t t; }; template <> struct V<16> { typedef u32x4 t; };
template <int RS, int WIDTH, int ALIGNED, int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * RS + 64];
  const uint32_t lane = threadIdx.x;
  uint8_t *row = lds + lane * RS;
  for (int i = lane; i < (64 * RS) / 4; i += 64) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t rnd = lane * 2654435761u + 12345u;
  uint32_t acc = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      rnd = rnd * 1664525u + 1013904223u;
      uint32_t off = (rnd >> 8) % (RS - 32);
      if (ALIGNED) off &= ~(uint32_t)(WIDTH - 1);
      typename V<WIDTH>::t v;
      if (OP == 0) { __builtin_memcpy(&v, row + off, WIDTH); uint32_t x; __builtin_memcpy(&x, &v, WIDTH < 4 ? WIDTH : 4); acc += x; }
      else { __builtin_memset(&v, (int)off, WIDTH); __builtin_memcpy(row + off, &v, WIDTH); }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (lane == 0) { out[blockIdx.x * 2] = (uint32_t)(t1 - t0); out[blockIdx.x * 2 + 1] = acc + lds[5]; }
}
template <int RS, int WIDTH, int ALIGNED, int OP>
void run(uint32_t *d)
{
  const int iters = 256; int grid = 256 * 8;
  hipLaunchKernelGGL((k<RS, WIDTH, ALIGNED, OP>), dim3(grid), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL((k<RS, WIDTH, ALIGNED, OP>), dim3(grid), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  double instrs = (double)iters * 8 * grid;
  double cyc = ms * 1e-3 * 2.4e9 * 256 / instrs;  // LDS-pipe cycles per wave-instruction per CU (assuming 2.4 GHz)
  printf("%s width %2d %s RS=%d : %6.1f CU-cycles per wave-instr | %6.1f B/clk/CU | chip %5.1f TB/s\n", OP ? "write" : "read ", WIDTH, ALIGNED ? "aligned-random" : "byte-random   ", RS, cyc, 64.0 * WIDTH / cyc, instrs * 64 * WIDTH / ms / 1e9);
}
int main()
{
  uint32_t *d; hipMalloc(&d, 1 << 20);
  run<272, 16, 1, 0>(d); run<272, 16, 0, 0>(d); run<272, 8, 1, 0>(d); run<272, 8, 0, 0>(d); run<272, 4, 1, 0>(d); run<272, 4, 0, 0>(d); run<272, 1, 0, 0>(d);
  run<144, 16, 1, 1>(d); run<144, 16, 0, 1>(d); run<144, 8, 1, 1>(d); run<144, 8, 0, 1>(d); run<144, 4, 1, 1>(d); run<144, 4, 0, 1>(d); run<144, 1, 0, 1>(d);
  run<276, 16, 1, 0>(d); run<148, 16, 1, 1>(d); run<264, 8, 1, 0>(d); run<136, 8, 1, 1>(d);
  return 0;
}
