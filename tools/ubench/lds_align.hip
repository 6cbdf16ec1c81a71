// Microbenchmark (definitive): per-lane LDS access cost by width and ALIGNMENT GRANULARITY of random per-lane offsets.
// All loaded components are consumed (no narrowing by the compiler).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int RS, int WIDTH, int GRAN, int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * RS + 64];
  const uint32_t lane = threadIdx.x;
  uint8_t *row = lds + lane * RS;
  for (int i = lane; i < (64 * RS) / 4; i += 64) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t rnd = lane * 2654435761u + 12345u, acc = 0;
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      rnd = rnd * 1664525u + 1013904223u;
      const uint32_t off = ((rnd >> 8) % (RS - 32)) & ~(uint32_t)(GRAN - 1);
      if (WIDTH == 16) { if (OP == 0) { u32x4 v = *(const u32x4 *)__builtin_assume_aligned(row + off, 16); acc += v.x ^ v.y ^ v.z ^ v.w; } else { u32x4 v = { off, rnd, off, rnd }; *(u32x4 *)__builtin_assume_aligned(row + off, 16) = v; } }
      if (WIDTH == 8) { if (OP == 0) { u32x2 v = *(const u32x2 *)__builtin_assume_aligned(row + off, 8); acc += v.x ^ v.y; } else { u32x2 v = { off, rnd }; *(u32x2 *)__builtin_assume_aligned(row + off, 8) = v; } }
      if (WIDTH == 4) { if (OP == 0) { acc += *(const uint32_t *)__builtin_assume_aligned(row + off, 4); } else { *(uint32_t *)__builtin_assume_aligned(row + off, 4) = off; } }
    }
  }
  __syncthreads();
  if (lane == 0) out[blockIdx.x] = acc + lds[5];
}
template <int RS, int WIDTH, int GRAN, int OP>
void run(uint32_t *d)
{
  const int iters = 256; int grid = 256 * 8; float ms; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RS, WIDTH, GRAN, OP>), dim3(grid), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<RS, WIDTH, GRAN, OP>), dim3(grid), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  double instrs = (double)iters * 8 * grid, cyc = ms * 1e-3 * 2.4e9 * 256 / instrs;
  printf("%s b%-3d offsets multiple of %2d, row stride %3d : %6.1f CU-cycles per wave-instr  (%5.1f B/clk/CU)\n", OP ? "write" : "read ", WIDTH * 8, GRAN, RS, cyc, 64.0 * WIDTH / cyc);
}
int main()
{
  uint32_t *d; hipMalloc(&d, 1 << 20);
  run<272, 16, 16, 0>(d); run<272, 16, 8, 0>(d); run<272, 16, 4, 0>(d); run<292, 16, 16, 0>(d); run<292, 16, 4, 0>(d);
  run<272, 8, 8, 0>(d); run<272, 8, 4, 0>(d); run<272, 4, 4, 0>(d); run<292, 4, 4, 0>(d);
  run<144, 16, 16, 1>(d); run<144, 16, 8, 1>(d); run<144, 16, 4, 1>(d); run<148, 16, 16, 1>(d); run<148, 16, 4, 1>(d);
  run<144, 8, 8, 1>(d); run<144, 8, 4, 1>(d); run<144, 4, 4, 1>(d); run<148, 4, 4, 1>(d);
  return 0;
}
