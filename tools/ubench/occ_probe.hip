// occ_probe: waves (64-thread workgroups) per CU as a function of the LDS bytes per workgroup -> LDS allocation granularity
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(uint32_t *p) { extern __shared__ uint32_t s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[63 - threadIdx.x]; }
int main()
{
  int last = -1;
  for (int lds = 8192; lds <= 20480; lds += 64)
  {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, lds) != hipSuccess) { printf("query failed\n"); return 1; }
    if (n != last) { printf("lds %d -> %d workgroups/CU\n", lds, n); last = n; }
  }
  return 0;
}
