/*
 * ref_shim.c -- helper linked into oracle/_ref/libhsrle_ref.so (TEST INFRASTRUCTURE ONLY).
 *
 * The reference picks its SIMD body from global feature flags (src/simd_platform.c:28-112);
 * `hsrlekit --max-simd X` clears some of them after detection (src/main.c:172-313).  This shim
 * does the same from a test so that both tail flavours of rle8_packed_multi_compress (SSE2 body
 * vs AVX2 body) can be minted.  It contains no reference code: it only
 * declares the extern flags it needs.
 */
#include <stdbool.h>

extern bool _CpuFeaturesDetected;
extern bool sse3Supported, ssse3Supported, sse41Supported, sse42Supported, avxSupported, avx2Supported, fma3Supported, avx512FSupported;
void _DetectCPUFeatures();

/* level: 0 = as detected, 1 = cap at SSE2, 2 = cap at AVX (no AVX2), 3 = cap at AVX2 (no AVX-512F) */
void hsrle_ref_set_max_simd(int level)
{
  _CpuFeaturesDetected = false;
  _DetectCPUFeatures();

  if (level == 1)
  {
    sse3Supported = ssse3Supported = sse41Supported = sse42Supported = false;
    avxSupported = avx2Supported = fma3Supported = avx512FSupported = false;
  }
  else if (level == 2)
  {
    avx2Supported = fma3Supported = avx512FSupported = false;
  }
  else if (level == 3)
  {
    avx512FSupported = false;
  }
}

int hsrle_ref_has_avx2(void)
{
  _DetectCPUFeatures();
  return avx2Supported ? 1 : 0;
}
