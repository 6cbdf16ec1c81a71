// host_copy.hip -- what the host-pointer drop-in path can hope for on this box: pageable and pinned hipMemcpy, hipHostRegister, parallel memcpy
// into pinned staging, and a staged pipeline (threads copy piece k + 1 while the DMA engine moves piece k).  1 GiB each way.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_memcpy(void *d, const void *s, size_t n, int threads)
{
  std::vector<std::thread> t;
  const size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int i = 0; i < threads; i++)
  {
    const size_t a = (size_t)i * per, b = a + per < n ? a + per : n;
    if (a < b) t.emplace_back([=] { memcpy((char *)d + a, (const char *)s + a, b - a); });
  }
  for (auto &x : t) x.join();
}
int main()
{
  const size_t N = 1ull << 30;
  char *pageable = (char *)malloc(N), *pageable2 = (char *)malloc(N), *pinned = nullptr, *dev = nullptr;
  memset(pageable, 1, N); memset(pageable2, 2, N);
  hipHostMalloc((void **)&pinned, N, hipHostMallocDefault); memset(pinned, 3, N);
  hipMalloc((void **)&dev, N);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  auto gbs = [&](double t) { return N / t / 1e9; };
  for (int rep = 0; rep < 2; rep++)
  {
    double t = now(); hipMemcpy(dev, pageable, N, hipMemcpyHostToDevice); printf("pageable H2D %.1f GB/s\n", gbs(now() - t));
    t = now(); hipMemcpy(pageable2, dev, N, hipMemcpyDeviceToHost); printf("pageable D2H %.1f GB/s\n", gbs(now() - t));
    t = now(); hipMemcpy(dev, pinned, N, hipMemcpyHostToDevice); printf("pinned   H2D %.1f GB/s\n", gbs(now() - t));
    t = now(); hipMemcpy(pinned, dev, N, hipMemcpyDeviceToHost); printf("pinned   D2H %.1f GB/s\n", gbs(now() - t));
  }
  { double t = now(); hipError_t e = hipHostRegister(pageable, N, hipHostRegisterDefault); double t1 = now(); printf("hipHostRegister 1 GiB: %.1f ms (%s)\n", (t1 - t) * 1e3, hipGetErrorString(e));
    if (e == hipSuccess) { t = now(); hipMemcpy(dev, pageable, N, hipMemcpyHostToDevice); printf("registered H2D %.1f GB/s\n", gbs(now() - t)); t = now(); hipHostUnregister(pageable); printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3); } }
  for (int th : { 1, 2, 4, 8, 16, 32 }) { double t = now(); par_memcpy(pinned, pageable, N, th); printf("memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", th, gbs(now() - t)); }
  // staged pipeline H2D: pieces of P bytes through two pinned buffers; T threads copy, the DMA engine moves the piece before
  for (int th : { 4, 8, 16 }) for (size_t P : { (size_t)8 << 20, (size_t)32 << 20 })
  {
    double t = now();
    hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    size_t k = 0;
    for (size_t at = 0; at < N; at += P, k++)
    {
      char *stage = pinned + (k & 1) * P;
      if (k >= 2) hipEventSynchronize(ev[k & 1]);
      par_memcpy(stage, pageable + at, P, th);
      hipMemcpyAsync(dev + at, stage, P, hipMemcpyHostToDevice, s0);
      hipEventRecord(ev[k & 1], s0);
    }
    hipStreamSynchronize(s0);
    printf("staged H2D, %2d threads, %3zu MiB pieces: %.1f GB/s\n", th, P >> 20, gbs(now() - t));
    t = now(); k = 0;
    // staged D2H: DMA piece k into stage, threads copy piece k - 1 out
    size_t prevAt = 0; bool havePrev = false;
    for (size_t at = 0; at < N; at += P, k++)
    {
      char *stage = pinned + (k & 1) * P;
      hipMemcpyAsync(stage, dev + at, P, hipMemcpyDeviceToHost, s0);
      hipEventRecord(ev[k & 1], s0);
      if (havePrev) { hipEventSynchronize(ev[(k - 1) & 1]); par_memcpy(pageable2 + prevAt, pinned + ((k - 1) & 1) * P, P, th); }
      prevAt = at; havePrev = true;
    }
    hipEventSynchronize(ev[(k - 1) & 1]); par_memcpy(pageable2 + prevAt, pinned + ((k - 1) & 1) * P, N - prevAt, th);
    printf("staged D2H, %2d threads, %3zu MiB pieces: %.1f GB/s\n", th, P >> 20, gbs(now() - t));
  }
  // both directions at once (pinned): is the link full duplex for us?
  { char *dev2 = nullptr, *pinned2 = nullptr; hipMalloc((void **)&dev2, N); hipHostMalloc((void **)&pinned2, N, hipHostMallocDefault);
    double t = now(); hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s0); hipMemcpyAsync(pinned2, dev2, N, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    printf("pinned H2D + D2H together: %.1f GB/s each\n", gbs(now() - t)); }
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
