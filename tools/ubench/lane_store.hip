// Microbenchmark: per-lane sequential over-writing 16-byte stores (one lane = one B-byte block), dword aligned vs byte
// unaligned, random advance per store (desynchronises the lanes like a real decoder does).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE> // 0: advance 16 aligned; 1: random advance 4..16 step 4 (dword aligned); 2: random advance 1..16 bytes (byte aligned); 3: like 2 but store address rounded down to dword
__global__ __launch_bounds__(64) void k(uint8_t *out, uint32_t B, uint32_t nBlocks)
{
  uint32_t b = blockIdx.x * 64 + threadIdx.x; if (b >= nBlocks) return;
  uint8_t *o = out + (uint64_t)b * B; u32x4 v = { b, b, b, b };
  uint32_t rnd = b * 2654435761u + 777u, k = 0;
  while (k + 32 <= B)
  {
    rnd = rnd * 1664525u + 1013904223u;
    uint32_t adv = MODE == 0 ? 16u : (MODE == 1 ? 4u * (1u + ((rnd >> 10) & 3u)) : 1u + ((rnd >> 10) & 15u));
    v.x += k;
    if (MODE == 3) __builtin_memcpy(o + (k & ~3u), &v, 16); else __builtin_memcpy(o + k, &v, 16);
    k += adv;
  }
  for (; k < B; k++) o[k] = (uint8_t)k;
}
int main()
{
  const uint64_t U = 4ull << 30; uint8_t *b; hipMalloc(&b, U + 4096); hipMemset(b, 2, U);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t B : { 1024u, 2048u, 4096u, 4224u, 16384u })
    for (int mode = 0; mode < 4; mode++)
    {
      uint32_t nb = (uint32_t)(U / B); float ms;
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb);
      };
      launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      printf("B=%5u mode %d : %7.3f ms  %7.1f GB/s of output\n", B, mode, ms, (double)nb * B / ms / 1e6);
    }
  return 0;
}
