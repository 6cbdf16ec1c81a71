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

/* Run one of the reference's decompress functions (passed by address) over a sequence of block streams: what bench.py's
 * cpu_baseline leg times as kind "reference".  The reference decoders write up to 128 bytes past a block's end
 * (rle_decompress_additional_size); blocks are decoded in order so the next block repairs that, and the caller gives the
 * output buffer 256 bytes of slack for the last one. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
typedef uint32_t (*hsrle_ref_codec_fn)(const uint8_t *, uint32_t, uint8_t *, uint32_t);

uint64_t hsrle_ref_decode_blocks(hsrle_ref_codec_fn fn, const uint8_t *payload, const uint64_t *offsets, uint64_t nBlocks, uint32_t blockSize,
                                 uint8_t *pOut, uint64_t outSize)
{
  uint64_t produced = 0;
  for (uint64_t b = 0; b < nBlocks; b++)
  {
    const uint64_t at = b * blockSize;
    const uint32_t room = (uint32_t)((outSize - at) < blockSize ? (outSize - at) : blockSize);
    const uint32_t got = fn(payload + offsets[b], (uint32_t)(offsets[b + 1] - offsets[b]), pOut + at, room);
    if (got == 0) return 0;
    produced += got;
  }
  return produced;
}

uint64_t hsrle_ref_encode_blocks(hsrle_ref_codec_fn fn, const uint8_t *pIn, uint64_t inSize, uint32_t blockSize, uint8_t *pOut, uint32_t stride, uint32_t *pSizes)
{
  const uint64_t nBlocks = (inSize + blockSize - 1) / blockSize;
  for (uint64_t b = 0; b < nBlocks; b++)
  {
    const uint64_t off = b * blockSize;
    const uint32_t len = (uint32_t)((inSize - off) < blockSize ? (inSize - off) : blockSize);
    const uint32_t c = fn(pIn + off, len, pOut + b * stride, stride);
    if (c == 0) return 0;
    pSizes[b] = c;
  }
  return nBlocks;
}

/* Block by block from a guard-padded private copy of each block (the reference encoders of 2 .. 16 byte symbols read up to
 * 2 * S - 1 bytes past the end of their input and may match there; the pad is filled so that they never do -- what the manifests were minted
 * with and what the GPU library implements).  One thread, no hashing: the CPU encode baseline of bench.py for the wide codecs. */
uint64_t hsrle_ref_encode_blocks_guarded(hsrle_ref_codec_fn fn, const uint8_t *pIn, uint64_t inSize, uint32_t blockSize, uint8_t *pOut, uint32_t stride, uint32_t *pSizes)
{
  const uint64_t nBlocks = (inSize + blockSize - 1) / blockSize;
  uint8_t *tmp = (uint8_t *)malloc((size_t)blockSize + 128);
  if (!tmp) return 0;
  uint64_t done = 0;
  for (uint64_t b = 0; b < nBlocks; b++, done++)
  {
    const uint64_t off = b * blockSize;
    const uint32_t len = (uint32_t)((inSize - off) < blockSize ? (inSize - off) : blockSize);
    memcpy(tmp, pIn + off, len);
    for (uint32_t k = len; k < len + 64; k++)
    {
      static const int d[7] = { 1, 2, 3, 4, 6, 8, 16 };
      uint8_t v = 0;
      for (;;)
      {
        int clash = 0;
        for (int q = 0; q < 7; q++) if (k >= (uint32_t)d[q] && tmp[k - d[q]] == v) clash = 1;
        if (!clash) break;
        v++;
      }
      tmp[k] = v;
    }
    const uint32_t c = fn(tmp, len, pOut + b * stride, stride);
    if (c == 0) break;
    pSizes[b] = c;
  }
  free(tmp);
  return done == nBlocks ? nBlocks : 0;
}

/* The same over `nThreads` POSIX threads (one contiguous block range each): the "all host cores" context figure of bench.py.
 * Thread t writes its range at pOut + firstBlock_t * blockSize + 256 * t, i.e. every range has its own 256 bytes of slack
 * (neighbouring ranges would otherwise scribble over each other's first bytes); pOut needs outSize + 256 * (nThreads + 1) bytes.
 * Returns the number of bytes produced. */
#include <pthread.h>

typedef struct
{
  hsrle_ref_codec_fn fn;
  const uint8_t *payload;
  const uint64_t *offsets;
  uint64_t first, count;
  uint32_t blockSize;
  uint8_t *out;
  uint64_t produced;
} hsrle_ref_mt_job;

static void *hsrle_ref_mt_worker(void *p)
{
  hsrle_ref_mt_job *j = (hsrle_ref_mt_job *)p;
  j->produced = hsrle_ref_decode_blocks(j->fn, j->payload, j->offsets + j->first, j->count, j->blockSize, j->out, j->count * (uint64_t)j->blockSize);
  return 0;
}

uint64_t hsrle_ref_decode_blocks_mt(hsrle_ref_codec_fn fn, const uint8_t *payload, const uint64_t *offsets, uint64_t nBlocks, uint32_t blockSize,
                                    uint8_t *pOut, int nThreads)
{
  if (nThreads < 1) nThreads = 1;
  if (nThreads > 1024) nThreads = 1024;
  pthread_t th[1024];
  hsrle_ref_mt_job jobs[1024];
  uint64_t total = 0;

  for (int t = 0; t < nThreads; t++)
  {
    const uint64_t b0 = nBlocks * (uint64_t)t / (uint64_t)nThreads, b1 = nBlocks * (uint64_t)(t + 1) / (uint64_t)nThreads;
    jobs[t] = (hsrle_ref_mt_job){ fn, payload, offsets, b0, b1 - b0, blockSize, pOut + b0 * blockSize + 256u * (uint64_t)t, 0 };
    if (pthread_create(&th[t], 0, hsrle_ref_mt_worker, &jobs[t]) != 0)
    {
      hsrle_ref_mt_worker(&jobs[t]);
      th[t] = 0;
    }
  }
  for (int t = 0; t < nThreads; t++)
  {
    if (th[t]) pthread_join(th[t], 0);
    total += jobs[t].produced;
  }
  return total;
}

/* ---- big-config manifests (tests/golden/make_big_manifest.py): encode every block of a buffer with one of the reference's compress
 *      functions over nThreads POSIX threads, the block streams back to back per thread range (compacted by the caller through the
 *      sizes), and hash every stream (oracle/hsrle_hash.h).  guard != 0: every block is encoded from a private copy that is followed by
 *      the guard pad of SURVEY.md 8c (pad bytes differ from the bytes 1,2,3,4,6,8,16 positions in front of them), because the reference
 *      encoders of the symbols wider than 8 bit read up to 2 S - 1 bytes past inSize and this repository's semantics is "bytes beyond
 *      the end never match". ---- */
#include "hsrle_hash.h"
#include <stdlib.h>

typedef struct
{
  hsrle_ref_codec_fn fn;
  const uint8_t *in;
  uint64_t inSize, first, count;
  uint32_t blockSize, stride;
  int guard;
  uint8_t *slots;
  uint32_t *sizes;
  uint64_t *hashes;
  int failed;
} hsrle_ref_enc_job;

static void *hsrle_ref_enc_worker(void *p)
{
  hsrle_ref_enc_job *j = (hsrle_ref_enc_job *)p;
  uint8_t *tmp = j->guard ? (uint8_t *)malloc((size_t)j->blockSize + 128) : 0;
  for (uint64_t b = j->first; b < j->first + j->count; b++)
  {
    const uint64_t off = b * j->blockSize;
    const uint32_t len = (uint32_t)((j->inSize - off) < j->blockSize ? (j->inSize - off) : j->blockSize);
    const uint8_t *src = j->in + off;
    if (tmp)
    {
      memcpy(tmp, src, len);
      for (uint32_t k = len; k < len + 64; k++)
      {
        static const int d[7] = { 1, 2, 3, 4, 6, 8, 16 };
        uint8_t v = 0;
        for (;;)
        {
          int clash = 0;
          for (int q = 0; q < 7; q++) if (k >= (uint32_t)d[q] && tmp[k - d[q]] == v) clash = 1;
          if (!clash) break;
          v++;
        }
        tmp[k] = v;
      }
      src = tmp;
    }
    uint8_t *dst = j->slots + (b - j->first) * (uint64_t)j->stride;
    const uint32_t c = j->fn(src, len, dst, j->stride);
    if (c == 0) { j->failed = 1; break; }
    j->sizes[b] = c;
    j->hashes[b] = hsrle_hash64(dst, c);
  }
  free(tmp);
  return 0;
}

/* slots: nBlocks * stride bytes; thread t's blocks start at slot index firstBlock_t (i.e. block b's stream is at slots + b * stride) */
uint64_t hsrle_ref_encode_hash_blocks_mt(hsrle_ref_codec_fn fn, const uint8_t *pIn, uint64_t inSize, uint32_t blockSize, int guard, uint8_t *pSlots, uint32_t stride,
                                         uint32_t *pSizes, uint64_t *pHashes, int nThreads)
{
  const uint64_t nBlocks = (inSize + blockSize - 1) / blockSize;
  if (nThreads < 1) nThreads = 1;
  if (nThreads > 256) nThreads = 256;
  pthread_t th[256];
  hsrle_ref_enc_job jobs[256];
  for (int t = 0; t < nThreads; t++)
  {
    const uint64_t b0 = nBlocks * (uint64_t)t / (uint64_t)nThreads, b1 = nBlocks * (uint64_t)(t + 1) / (uint64_t)nThreads;
    jobs[t] = (hsrle_ref_enc_job){ fn, pIn, inSize, b0, b1 - b0, blockSize, stride, guard, pSlots + b0 * (uint64_t)stride, pSizes, pHashes, 0 };
    if (pthread_create(&th[t], 0, hsrle_ref_enc_worker, &jobs[t]) != 0) { hsrle_ref_enc_worker(&jobs[t]); th[t] = 0; }
  }
  int failed = 0;
  for (int t = 0; t < nThreads; t++) { if (th[t]) pthread_join(th[t], 0); failed |= jobs[t].failed; }
  return failed ? 0 : nBlocks;
}

/* payload = the block streams back to back; returns its size */
uint64_t hsrle_ref_compact(const uint8_t *pSlots, uint32_t stride, const uint32_t *pSizes, uint64_t nBlocks, uint8_t *pOut)
{
  uint64_t at = 0;
  for (uint64_t b = 0; b < nBlocks; b++) { memcpy(pOut + at, pSlots + b * (uint64_t)stride, pSizes[b]); at += pSizes[b]; }
  return at;
}

uint64_t hsrle_ref_hash64(const uint8_t *p, uint64_t len) { return hsrle_hash64(p, len); }
void hsrle_ref_rollups(const uint64_t *hashes, uint64_t n, uint64_t group, uint64_t *out)
{
  for (uint64_t g = 0; g * group < n; g++) out[g] = hsrle_rollup(hashes + g * group, (n - g * group) < group ? (n - g * group) : group);
}
