// Microbenchmark: what does the memory system do with one-lane-per-block access patterns?
//   w_al   : every lane writes its own B-byte block sequentially with aligned 16-byte stores
//   w_un   : same, but each store is shifted by a per-lane odd offset (unaligned, overlapping "over-write" style)
//   r_al   : every lane reads its own region sequentially with 16-byte loads (stride = B/2 between lanes)
//   copy   : float4 coalesced copy (reference ceiling)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void w_al(uint8_t *out, uint32_t B, uint32_t nBlocks)
{
  uint32_t b = blockIdx.x * 64 + threadIdx.x; if (b >= nBlocks) return;
  uint8_t *o = out + (uint64_t)b * B; u32x4 v = { b, b, b, b };
  for (uint32_t k = 0; k < B; k += 16) { v.x += k; __builtin_memcpy(o + k, &v, 16); }
}
__global__ __launch_bounds__(64) void w_un(uint8_t *out, uint32_t B, uint32_t nBlocks)
{
  uint32_t b = blockIdx.x * 64 + threadIdx.x; if (b >= nBlocks) return;
  uint8_t *o = out + (uint64_t)b * B; u32x4 v = { b, b, b, b };
  uint32_t k = 0, step = 5 + (b % 11);
  // variable advance 5..15 bytes per store with full 16-byte stores (over-write pattern), then exact tail
  for (; k + 16 <= B; k += step) { v.x += k; __builtin_memcpy(o + k, &v, 16); }
  for (; k < B; k++) o[k] = (uint8_t)k;
}
__global__ __launch_bounds__(64) void r_al(const uint8_t *in, uint32_t stride, uint32_t len, uint32_t nBlocks, uint32_t *sink)
{
  uint32_t b = blockIdx.x * 64 + threadIdx.x; if (b >= nBlocks) return;
  const uint8_t *p = in + (uint64_t)b * stride; uint32_t acc = 0;
  for (uint32_t k = 0; k + 16 <= len; k += 16) { u32x4 v; __builtin_memcpy(&v, p + k, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678) sink[0] = acc;
}
__global__ void copy4(const u32x4 *in, u32x4 *out, uint64_t n)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main()
{
  const uint64_t U = 4ull << 30; uint8_t *a, *b; uint32_t *sink;
  CK(hipMalloc(&a, U)); CK(hipMalloc(&b, U)); CK(hipMalloc(&sink, 64)); CK(hipMemset(a, 1, U)); CK(hipMemset(b, 2, U));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, double bytes, auto fn) { fn(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 3; i++) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6); };
  timeit("copy4 (read+write)", 2.0 * U, [&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, U / 16); });
  for (uint32_t B : { 1024u, 4096u, 16384u })
  {
    uint32_t nb = (uint32_t)(U / B); char nm[64];
    snprintf(nm, 64, "w_al B=%u", B); timeit(nm, (double)U, [&] { hipLaunchKernelGGL(w_al, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb); });
    snprintf(nm, 64, "w_un B=%u", B); timeit(nm, (double)U, [&] { hipLaunchKernelGGL(w_un, dim3((nb + 63) / 64), dim3(64), 0, 0, b, B, nb); });
    snprintf(nm, 64, "r_al stride=%u", B); timeit(nm, (double)U, [&] { hipLaunchKernelGGL(r_al, dim3((nb + 63) / 64), dim3(64), 0, 0, a, B, B, nb, sink); });
  }
  return 0;
}
