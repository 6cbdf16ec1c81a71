// Microbenchmark: the decoder's flush store pattern alone (64 rows x T bytes per round, rows B bytes apart) vs a linear store.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int T, int NT>
__global__ __launch_bounds__(64) void k_rows(uint8_t *out, uint32_t B, uint32_t nBlocks)
{
  constexpr int CPR = T / 16, RPI = 64 / CPR;
  const uint32_t lane = threadIdx.x, wgFirst = blockIdx.x * 64;
  u32x4 v = { lane, lane, lane, lane };
  for (uint32_t base = 0; base < B; base += T)
  {
#pragma unroll
    for (int q = 0; q < CPR; q++)
    {
      const uint32_t r = wgFirst + q * RPI + lane / CPR;
      if (r < nBlocks) { u32x4 *p = (u32x4 *)(out + (uint64_t)r * B + base + (lane % CPR) * 16u); if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
    }
    v.x += base;
  }
}
__global__ __launch_bounds__(64) void k_linear(uint8_t *out, uint32_t B, uint32_t nBlocks)
{
  const uint32_t lane = threadIdx.x; u32x4 v = { lane, lane, lane, lane };
  uint8_t *o = out + (uint64_t)blockIdx.x * 64 * B;
  for (uint32_t k = 0; k < 64 * B; k += 1024) { __builtin_memcpy(o + k + lane * 16, &v, 16); v.x += k; }
}
int main()
{
  const uint64_t U = 8ull << 30; uint8_t *b; hipMalloc(&b, U + (1 << 20)); hipMemset(b, 2, U);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t B : { 1024u, 4096u, 4224u, 16384u })
    for (int mode = 0; mode < 5; mode++)
    {
      uint32_t nb = (uint32_t)(U / B); float ms; dim3 g((nb + 63) / 64);
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL((k_rows<128, 0>), g, dim3(64), 0, 0, b, B, nb);
        if (mode == 1) hipLaunchKernelGGL((k_rows<64, 0>), g, dim3(64), 0, 0, b, B, nb);
        if (mode == 3) hipLaunchKernelGGL((k_rows<128, 1>), g, dim3(64), 0, 0, b, B, nb);
        if (mode == 4) hipLaunchKernelGGL((k_rows<64, 1>), g, dim3(64), 0, 0, b, B, nb);
        if (mode == 2) hipLaunchKernelGGL(k_linear, g, dim3(64), 0, 0, b, B, nb);
      };
      launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      printf("B=%5u %-22s: %7.3f ms  %7.1f GB/s\n", B, mode == 0 ? "rows T=128" : (mode == 1 ? "rows T=64" : (mode == 2 ? "linear 1 KiB/instr" : (mode == 3 ? "rows T=128 nontemporal" : "rows T=64 nontemporal"))), ms, (double)nb * B / ms / 1e6);
    }
  return 0;
}
