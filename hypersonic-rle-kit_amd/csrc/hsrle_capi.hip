// hsrle_capi.hip -- the C ABI of libhsrle_hip.so (include/hsrle.h): drop-in rle.h entry points, the device-resident
// block container API, container assembly kernels (size scan + slot compaction) and the synthetic workload generator.
//
// Host code here is plain C-style C++ behind `extern "C"`; no torch types, no CPU codec.  If no HIP device is usable
// every entry point fails loudly (0 / HSRLE_ERR_DEVICE) -- there is no fallback path.
#include "../../include/hsrle.h"

#include "hsrle_common.hip.h"
#include "hsrle_launch.h"
#include "hsrle_index.hip.h"
#include "hsrle_expand.hip.h"
#include "hsrle_mono_encode.hip.h"
#ifdef HSRLE_EXPERIMENTS
#include "experiments/hsrle_encode8w.hip.h"
#else
namespace hsrle { constexpr uint64_t kTicketBytes = 256; }   // (scratch behind the staging slots: the ring choice of the 1 / 2 byte encoders lives there)
#endif
#include "hsrle_rle8m.hip.h"

#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <string.h>

namespace hsrle {

// ------------------------------------------------------------------------------------------------------------------
// codec table (names follow the reference: src/rle.h, src/codec_funcs.h:270-410)

static const char *const kCodecNames[kCodecCount] = {
  "rle8_multi", "rle8_packed_multi", "rle8_3symlut", "rle8_7symlut", "rle8_single", "rle8_packed_single",
  "rle16_sym", "rle16_sym_packed", "rle16_3symlut_sym", "rle16_7symlut_sym", "rle16_byte", "rle16_byte_packed", "rle16_3symlut_byte", "rle16_7symlut_byte",
  "rle24_sym", "rle24_sym_packed", "rle24_3symlut_sym", "rle24_7symlut_sym", "rle24_byte", "rle24_byte_packed", "rle24_3symlut_byte", "rle24_7symlut_byte",
  "rle32_sym", "rle32_sym_packed", "rle32_3symlut_sym", "rle32_7symlut_sym", "rle32_byte", "rle32_byte_packed", "rle32_3symlut_byte", "rle32_7symlut_byte",
  "rle48_sym", "rle48_sym_packed", "rle48_3symlut_sym", "rle48_7symlut_sym", "rle48_byte", "rle48_byte_packed", "rle48_3symlut_byte", "rle48_7symlut_byte",
  "rle64_sym", "rle64_sym_packed", "rle64_3symlut_sym", "rle64_7symlut_sym", "rle64_byte", "rle64_byte_packed", "rle64_3symlut_byte", "rle64_7symlut_byte",
  "rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed",
  // Short family (reference: src/rle.h:202-348, src/codec_funcs.h:283-388)
  "rle8_multi_short", "rle8_1symlut_short", "rle8_3symlut_short", "rle8_7symlut_short",
  "rle16_sym_short", "rle16_1symlut_sym_short", "rle16_3symlut_sym_short", "rle16_7symlut_sym_short", "rle16_byte_short", "rle16_1symlut_byte_short", "rle16_3symlut_byte_short", "rle16_7symlut_byte_short",
  "rle24_sym_short", "rle24_1symlut_sym_short", "rle24_3symlut_sym_short", "rle24_7symlut_sym_short", "rle24_byte_short", "rle24_1symlut_byte_short", "rle24_3symlut_byte_short", "rle24_7symlut_byte_short",
  "rle32_sym_short", "rle32_1symlut_sym_short", "rle32_3symlut_sym_short", "rle32_7symlut_sym_short", "rle32_byte_short", "rle32_1symlut_byte_short", "rle32_3symlut_byte_short", "rle32_7symlut_byte_short",
  "rle48_sym_short", "rle48_1symlut_sym_short", "rle48_3symlut_sym_short", "rle48_7symlut_sym_short", "rle48_byte_short", "rle48_1symlut_byte_short", "rle48_3symlut_byte_short", "rle48_7symlut_byte_short",
  "rle64_sym_short", "rle64_1symlut_sym_short", "rle64_3symlut_sym_short", "rle64_7symlut_sym_short", "rle64_byte_short", "rle64_1symlut_byte_short", "rle64_3symlut_byte_short", "rle64_7symlut_byte_short",
  // Greedy encoders (reference: src/rle.h:398-416); the decode side is the Short decoder of the same grammar
  "rle16_1symlut_byte_short_greedy", "rle16_3symlut_byte_short_greedy", "rle16_7symlut_byte_short_greedy",
  "rle24_1symlut_byte_short_greedy", "rle24_3symlut_byte_short_greedy", "rle24_7symlut_byte_short_greedy",
  "rle32_1symlut_byte_short_greedy", "rle32_3symlut_byte_short_greedy", "rle32_7symlut_byte_short_greedy",
  "rle48_1symlut_byte_short_greedy", "rle48_3symlut_byte_short_greedy", "rle48_7symlut_byte_short_greedy",
  "rle64_1symlut_byte_short_greedy", "rle64_3symlut_byte_short_greedy", "rle64_7symlut_byte_short_greedy",
  "rle8_single_short",
};

static inline bool codec_is_lut(int c) { return c == 2 || c == 3 || (c >= 6 && c < 46 && (((c - 6) & 3) >= 2)) || c >= kShortBase8; }   // 8-byte stream header
static inline int codec_symbol_bytes(int c)
{
  static const int widths[5] = { 2, 3, 4, 6, 8 };
  if (c < 6) return 1;
  if (c < 46) return widths[(c - 6) >> 3];
  if (c < 50) return 16;
  if (c < kShortBaseW) return 1;
  if (c < kGreedyBase) return widths[(c - kShortBaseW) >> 3];
  if (c < kSingleShort) return widths[(c - kGreedyBase) / 3];
  return 1;
}
static inline uint32_t codec_header_size(int c) { return (c < 6 && !codec_is_lut(c)) ? 9u : 8u; }

static DecodeLaunch g_dec[kCodecCount];
static EncodeLaunch g_enc[kCodecCount];
static IndexLaunch g_idx[kCodecCount];
static SubBlockLaunch g_sub[kCodecCount];
static MonoEncodeLaunch g_menc[kCodecCount];
static WaveEncodeLaunch g_wenc[kCodecCount];
static PpwLaunch g_ppwL[kCodecCount];               // ... and those of hsrle_encodeLp.hip.h (hsrle_encodeLpw.hip.h)
static PpwLaunch g_ppwS[kCodecCount];               // ... and the codecs of hsrle_encodeSp.hip.h (hsrle_encodeSpw.hip.h), by codec id
static PpwLaunch g_ppw[2];                         // ... for units of any length (hsrle_encode8pw.hip.h): rle8_multi, rle8_packed_multi
static PpLaunch g_pp[kCodecCount];                  // position-parallel encoders (hsrle_encode8p.hip.h)
static std::once_flag g_tableOnce;

static void init_tables()
{
  std::call_once(g_tableOnce, [] {
    register_w8(g_dec, g_enc, g_idx, g_sub, g_menc, g_wenc);
    register_pp8(g_pp);
    register_pp8w(g_ppw);
    register_ppSw(g_ppwS);
    register_ppLw(g_ppwL);
    register_pp8s(g_pp);
    register_pp128(g_pp);
    register_ppL(g_pp);
    register_ppS(g_pp);
    register_w16(g_dec, g_enc, g_idx, g_sub, g_menc);
    register_w24(g_dec, g_enc, g_idx, g_sub, g_menc);
    register_w32(g_dec, g_enc, g_idx, g_sub, g_menc);
    register_w48(g_dec, g_enc, g_idx, g_sub, g_menc);
    register_w64(g_dec, g_enc, g_idx, g_sub, g_menc);
    register_w128(g_dec, g_enc, g_idx, g_sub, g_menc);
  });
}

static inline uint32_t bounds32(uint32_t n) { return (n > (1u << 30)) ? 0u : n + (16 + 4 + 1 + 4 + 1 + 64) * 2 + (3 * 4) + 1; }
static inline uint32_t slot_stride(uint32_t B) { return (bounds32(B) + 15u) & ~15u; }
static inline uint64_t block_count(uint64_t U, uint32_t B) { return (U + B - 1) / B; }
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

static uint32_t env_u32(const char *name, uint32_t dflt) { return knob_u32(name, dflt); }   // (developer knobs: -DHSRLE_EXPERIMENTS builds only, hsrle_launch.h)
static uint32_t pow2_floor(uint64_t v) { uint32_t r = 1; while ((uint64_t)r * 2u <= v && r < 0x80000000u) r *= 2u; return r; }

static bool valid_block_size(uint32_t B) { return B >= HSRLE_MIN_BLOCK_SIZE && B <= HSRLE_MAX_BLOCK_SIZE && (B % 128u) == 0; }

// ------------------------------------------------------------------------------------------------------------------
// container assembly kernels

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems; // 2048 elements per workgroup

template <int THREADS = kScanThreads>
__device__ __forceinline__ uint64_t wg_exclusive_scan_u64(uint64_t v, uint64_t *total)
{
  // wave scan by shuffles, then a scan of the wave totals through LDS
  __shared__ uint64_t waveTotals[THREADS / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint64_t y = __shfl_up(x, d, 64);
    if ((int)lane >= d) x += y;
  }
  if (lane == 63u) waveTotals[wave] = x;
  __syncthreads();
  uint64_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; w++)
  {
    const uint64_t t = waveTotals[w];
    if ((uint32_t)w < wave) base += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return base + x - v;
}

// sums[wg] = sum of in[wg * 2048 .. +2048)
template <typename TIN>
__global__ __launch_bounds__(kScanThreads) void k_tile_sums(const TIN *__restrict__ in, uint64_t n, uint64_t *__restrict__ sums)
{
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
  {
    const uint64_t idx = base + (uint64_t)k * kScanThreads + threadIdx.x;
    if (idx < n) acc += (uint64_t)in[idx];
  }
  uint64_t total;
  wg_exclusive_scan_u64(acc, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// out[i] = tileBase[wg] + exclusive prefix of in within the tile; out[n] = grand total when writeTotal.
// `in` and `out` may alias (in-place scan of a sums level): every thread reads its items before it writes them.
// `carry` (may be null): a device value added to every result -- the chunked compression scans chunk after chunk, each starting at the
// total of the chunks in front of it (which is the out[n] the previous chunk's scan wrote: carry may alias out[0]).
template <typename TIN>
__global__ __launch_bounds__(kScanThreads) void k_tile_scan(const TIN *in, uint64_t n, const uint64_t *tileBase, uint64_t *out, int writeTotal, const uint64_t *carry = nullptr)
{
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t v[kScanItems];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
  {
    v[k] = (base + k < n) ? (uint64_t)in[base + k] : 0;
    acc += v[k];
  }
  uint64_t total;
  const uint64_t carried = carry ? *carry : 0;
  uint64_t run = wg_exclusive_scan_u64(acc, &total) + (tileBase ? tileBase[blockIdx.x] : 0) + carried;
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
  {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (writeTotal && n > 0 && base <= n - 1 && n - 1 < base + kScanItems) // the thread that owns the last element
    out[n] = run;
}

// one wave per block: copy the slot stream to its place in the payload (destination-aligned 16-byte stores)
__global__ __launch_bounds__(256) void k_compact(const uint8_t *__restrict__ slots, uint32_t slotStride, const uint64_t *__restrict__ offsets,
                                                 uint8_t *__restrict__ payload, uint32_t nBlocks)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t b = xcd_tile(blockIdx.x, gridDim.x) * 4u + (threadIdx.x >> 6);   // XCD-aware tile order (hsrle_common.hip.h)
  if (b >= nBlocks)
    return;

  const uint64_t off = offsets[b];
  const uint32_t size = (uint32_t)(offsets[b + 1] - off);
  const uint8_t *src = slots + (uint64_t)b * slotStride;
  uint8_t *dst = payload + off;

  uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
  if (head > size) head = size;
  if (lane < head) dst[lane] = src[lane];

  const uint32_t body = (size - head) & ~15u;
  for (uint32_t k = lane * 16u; k < body; k += 64u * 16u)
    st128(dst + head + k, ld128(src + head + k));

  const uint32_t tail = size - head - body;
  if (lane < tail) dst[head + body + lane] = src[head + body + lane];
}

struct ContainerHeader
{
  char magic[8];
  uint32_t version, codec;
  uint64_t uncompressedSize;
  uint32_t blockSize, blockCount;
  uint64_t payloadSize, totalSize;
  uint8_t reserved[16];
};
static_assert(sizeof(ContainerHeader) == HSRLE_CONTAINER_HEADER_SIZE, "container header is 64 bytes");

__device__ __forceinline__ void finish_container(uint8_t *__restrict__ container, uint32_t codec, uint64_t U, uint32_t B, uint32_t nBlocks, uint64_t payloadSize);

__global__ void k_finish_container(uint8_t *__restrict__ container, uint32_t codec, uint64_t U, uint32_t B, uint32_t nBlocks)
{
  // offsets[nBlocks] was written by the scan; fill the header and the zero tail pad
  const uint64_t *offsets = (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE);
  finish_container(container, codec, U, B, nBlocks, offsets[nBlocks]);
}

// Small containers (up to kScanSmallMax blocks): the size scan in ONE launch -- every workgroup first adds up all sizes in front of its tile
// itself (at most 128 KB of coalesced reads from L2: cheaper than a launch), then scans its tile; the last one writes the container's header
// and tail pad.  One launch where k_tile_sums + 2 x k_tile_scan + k_finish_container were four, ~5 us each on a call of 150 (BASELINE
// config 3).  (A version with one workgroup of 1024 threads took 17 us: its loads and stores were strided by thread.)
constexpr uint32_t kScanSmallMax = 262144u;   // (128 workgroups, the last of which adds up 1 MiB of sizes: still cheaper than two more launches)
__global__ __launch_bounds__(kScanThreads) void k_scan_small_finish(const uint32_t *__restrict__ sizes, uint32_t n, uint64_t *__restrict__ out, uint8_t *__restrict__ container, uint32_t codec,
                                                                    uint64_t U, uint32_t B)
{
  const uint32_t tileFirst = blockIdx.x * (uint32_t)kScanTile;
  // sum of sizes[0, tileFirst): 16 bytes per thread and load, four loads in flight (tileFirst is a multiple of 2048)
  uint64_t before = 0;
  for (uint32_t i = threadIdx.x * 4u; i < tileFirst; i += 4u * 4u * kScanThreads)
  {
    u32x4 q[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      const uint32_t at = i + j * 4u * kScanThreads;
      q[j] = *(const u32x4 *)(sizes + (at < tileFirst ? at : 0u));
      if (at >= tileFirst) q[j] = u32x4{ 0, 0, 0, 0 };
    }
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) before += (uint64_t)q[j].x + q[j].y + q[j].z + q[j].w;
  }
  uint64_t beforeAll;
  (void)wg_exclusive_scan_u64(before, &beforeAll);

  const uint32_t base = tileFirst + threadIdx.x * (uint32_t)kScanItems;
  static_assert(kScanItems == 8, "two 16-byte loads per thread");
  const uint32_t lastVec = (n - 1u) >> 2;                                // (the size table is padded to 256 bytes: the vector that holds size n - 1 is readable)
  const uint32_t v0 = base >> 2, v1 = v0 + 1u;
  const u32x4 qa = *(const u32x4 *)(sizes + 4u * (v0 < lastVec ? v0 : lastVec)), qb = *(const u32x4 *)(sizes + 4u * (v1 < lastVec ? v1 : lastVec));
  const uint32_t x[8] = { qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w };
  uint32_t v[8];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k] = (base + (uint32_t)k < n) ? x[k] : 0u; acc += v[k]; }
  uint64_t total;
  uint64_t run = wg_exclusive_scan_u64(acc, &total) + beforeAll;
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    if (base + (uint32_t)k < n) out[base + (uint32_t)k] = run;
    run += v[k];
  }
  if (blockIdx.x == gridDim.x - 1u)
  {
    if (threadIdx.x == 0) out[n] = beforeAll + total;
    if (container != nullptr) finish_container(container, codec, U, B, n, beforeAll + total);   // (nullptr: a scan only -- scan_sizes)
  }
}

__device__ __forceinline__ void finish_container(uint8_t *__restrict__ container, uint32_t codec, uint64_t U, uint32_t B, uint32_t nBlocks, uint64_t payloadSize)
{
  const uint64_t payloadStart = HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)nBlocks + 1ull);

  if (threadIdx.x == 0)
  {
    ContainerHeader h;
    const char m[8] = { 'H', 'S', 'R', 'L', 'E', 'K', 'I', 'T' };
    for (int k = 0; k < 8; k++) h.magic[k] = m[k];
    h.version = 1;
    h.codec = codec;
    h.uncompressedSize = U;
    h.blockSize = B;
    h.blockCount = nBlocks;
    h.payloadSize = payloadSize;
    h.totalSize = payloadStart + payloadSize + HSRLE_CONTAINER_TAIL_PAD;
    for (int k = 0; k < 16; k++) h.reserved[k] = 0;
    *(ContainerHeader *)container = h;
  }

  if (threadIdx.x < HSRLE_CONTAINER_TAIL_PAD)
    container[payloadStart + payloadSize + threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 64 bit hash of every block stream of a container (include/hsrle.h: hsrle_hash_blocks_dev_async): what the big-config manifests pin
// (tests/golden/big/, minted from the compiled reference) -- every block of an 8 GiB container is compared, not a sample.

__device__ __forceinline__ uint64_t rotl64(uint64_t v, int sh) { return (v << sh) | (v >> (64 - sh)); }

__global__ __launch_bounds__(256) void k_hash_blocks(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, uint64_t payloadBytes, uint32_t firstBlock,
                                                     uint32_t blockCount, uint64_t *__restrict__ out)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= blockCount) return;
  const uint64_t off0 = offsets[firstBlock + i], off1 = offsets[firstBlock + i + 1u];
  if (off0 > off1 || off1 > payloadBytes) { out[i] = 0; return; }
  const uint8_t *p = payload + off0;
  const uint64_t len = off1 - off0;
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (len * 0xD6E8FEB86659FD93ull);
  uint64_t k = 0;
  for (; k + 8 <= len; k += 8)
    h = rotl64(h ^ ld64(p + k), 27) * 0x9E3779B97F4A7C15ull + 0x165667B19E3779F9ull;
  if (k < len)
  {
    uint64_t w = 0;
    for (uint32_t j = 0; k + j < len; j++) w |= (uint64_t)p[k + j] << (8u * j);
    h = rotl64(h ^ w, 27) * 0x9E3779B97F4A7C15ull + 0x165667B19E3779F9ull;
  }
  out[i] = h ^ (h >> 31);
}

// ------------------------------------------------------------------------------------------------------------------
// synthetic workloads.  One lane generates one 64 KiB chunk; chunks are independent so the same
// bytes can be produced on the CPU (oracle/hsrle_synth.c, tests/hsrle_testlib.py:synth_chunk_py) for any slice.

constexpr uint32_t kSynthChunk = 65536u;

__device__ __forceinline__ uint64_t splitmix64(uint64_t &state)
{
  state += 0x9E3779B97F4A7C15ull;
  uint64_t z = state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(64) void k_synth(int kind, int S, uint64_t seed, uint8_t *__restrict__ out, uint64_t size)
{
  const uint64_t chunk = (uint64_t)blockIdx.x * 64u + threadIdx.x;
  const uint64_t start = chunk * kSynthChunk;
  if (start >= size)
    return;

  const uint32_t len = (uint32_t)((size - start) < kSynthChunk ? (size - start) : kSynthChunk);
  uint8_t *o = out + start;
  uint64_t st = seed * 0x9E3779B97F4A7C15ull + chunk * 0xD1B54A32D192ED03ull + (uint64_t)kind;
  uint32_t at = 0;

  if (kind == HSRLE_SYNTH_RUNS)
  {
    while (at < len)
    {
      uint64_t r = splitmix64(st);
      uint32_t L = 1u + (uint32_t)(r % 63u);
      for (uint32_t k = 0; k < L; k += 8)
      {
        const uint64_t v = splitmix64(st);
        for (uint32_t j = 0; j < 8 && k + j < L; j++)
          if (at + k + j < len) o[at + k + j] = (uint8_t)(v >> (8 * j));
      }
      at += L;

      r = splitmix64(st);
      const uint32_t R = 2u + (uint32_t)(r % 62u);
      uint8_t sym[16];
      for (int k = 0; k < S; k += 8)
      {
        const uint64_t v = splitmix64(st);
        for (int j = 0; j < 8 && k + j < S; j++) sym[k + j] = (uint8_t)(v >> (8 * j));
      }
      for (uint32_t k = 0; k < R * (uint32_t)S && at + k < len; k++)
        o[at + k] = sym[k % (uint32_t)S];
      at += R * (uint32_t)S;
    }
  }
  else
  {
    const uint8_t vals[6] = { 0x01, 0x02, 0x03, 0xFF, 0xFE, 0x04 };
    while (at < len)
    {
      uint64_t r = splitmix64(st);
      const uint32_t Z = (((r >> 32) & 3u) == 0u) ? 40u + (uint32_t)(r % 120u) : 10u + (uint32_t)(r % 16u);
      for (uint32_t k = 0; k < Z && at + k < len; k++) o[at + k] = 0;
      at += Z;
      r = splitmix64(st);
      const uint32_t Bn = 1u + (uint32_t)(r % 9u);
      for (uint32_t k = 0; k < Bn && at + k < len; k++) o[at + k] = vals[(r >> (8 + 4 * k)) % 6u];
      at += Bn;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// device state

// Per DEVICE (hipGetDevice() of the calling thread): the staging buffers of the host-pointer drop-in functions.  A drop-in call holds
// the device's mutex from its first copy to its last, so concurrent drop-in calls on one device run one after the other (the reference's
// own callers are single threaded: src/main.c, src/rle_fuzz.c); calls on different devices do not meet.  The device-pointer API keeps NO
// library-owned state between calls: what it needs beyond the caller's buffers is allocated stream-ordered (hipMallocAsync on the
// caller's stream) for the duration of the call.
constexpr int kMaxDevices = 64;
struct DeviceState
{
  std::mutex mu;
  void *ws = nullptr;    // rle8m drop-in: compression workspace
  uint64_t wsSize = 0;
  hipStream_t aux = nullptr;   // second stream of the chunked compression
  hipMemPool_t pool = nullptr; // the library's OWN stream-ordered pool on this device (scratch_alloc): the application's default pool is never touched
  bool poolTried = false;
  void *monoIn = nullptr, *monoOut = nullptr, *monoAux = nullptr, *monoWs = nullptr; // staging of the drop-in (host pointer) path
  uint64_t monoInSize = 0, monoOutSize = 0, monoWsSize = 0;
};

static DeviceState g_devs[kMaxDevices];
static std::once_flag g_deviceOnce;
static int g_deviceCount = 0;

static bool device_ok()
{
  std::call_once(g_deviceOnce, [] {
    int n = 0;
    g_deviceCount = (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? n : 0;
  });
  return g_deviceCount > 0;
}

static DeviceState &this_device()
{
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return g_devs[d];
}

// Scratch for the duration of one call on `st`: stream-ordered, so concurrent calls on other streams (or threads) never share it.  It
// comes from a pool the LIBRARY owns (one per device, created on first use): freed scratch stays in that pool up to the retention set
// with hsrle_scratch_retention() (default 2 GiB: the small per-call scratch of decode / info calls is never given back and re-mapped;
// the 9 GB workspace of an 8 GiB compression without a caller's workspace is) and hsrle_trim() hands all of it back.  Round 2 raised the
// release threshold of the device's DEFAULT pool instead, which kept gigabytes away from the host application's own allocator.
static std::atomic<uint64_t> g_scratchRetention{ 2ull << 30 };
static hipMemPool_t device_pool()
{
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!D.poolTried)
  {
    D.poolTried = true;
    int d = 0;
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = (hipGetDevice(&d) == hipSuccess) ? d : 0;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess)
    {
      uint64_t keep = g_scratchRetention.load();
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      D.pool = pool;
    }
    else
      (void)hipGetLastError();
  }
  return D.pool;
}
static void *scratch_alloc(uint64_t bytes, hipStream_t st)
{
  void *p = nullptr;
  hipMemPool_t pool = device_pool();
  if (pool != nullptr && hipMallocFromPoolAsync(&p, bytes, pool, st) == hipSuccess) return p;
  (void)hipGetLastError();
  if (hipMallocAsync(&p, bytes, st) == hipSuccess) return p;           // (no pool of our own on this runtime: the default pool, untouched)
  (void)hipGetLastError();
  return nullptr;
}
static void scratch_free(void *p, hipStream_t st) { if (p) (void)hipFreeAsync(p, st); }

static bool grow(void **p, uint64_t *have, uint64_t need)
{
  if (*have >= need && *p)
    return true;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  if (hipMalloc(p, need) != hipSuccess)
    return false;
  *have = need;
  return true;
}

// workspace layout: [slots][sizes u32][level-1 sums u64][level-2 sums u64][level-3 sums u64]
//
// Optionally (encode_chunk_blocks) inputs are compressed CHUNK by chunk: the compaction of chunk k (bandwidth bound, no LDS) runs on a
// second stream while the encode kernel (latency bound, LDS limited to 9 waves per CU) works on chunk k + 1, and the staging slots are
// two chunks' worth instead of the whole input's (8 GiB: 2.3 GB instead of 9 GB).  The scan of a chunk's stream sizes starts at the
// running total the previous chunk left in the offset table.
static uint32_t encode_chunk_blocks()
{
  // OFF by default: measured on the 8 GiB headline buffer the overlap does not pay -- 1 130 GiB/s in one piece against 846 / 964 / 1 015 /
  // 1 063 GiB/s with chunks of 131 072 / 262 144 / 524 288 / 1 048 576 blocks (the compaction's traffic slows the encode kernel by more
  // than its own 1.8 ms, and every chunk boundary drains the GPU).  HSRLE_ENCODE_CHUNK_BLOCKS=n turns it on for callers that would rather
  // have the smaller workspace (two chunks of staging slots instead of the whole input's).
  static const uint32_t v = [] { const uint32_t e = env_u32("HSRLE_ENCODE_CHUNK_BLOCKS", 0); return e ? e : 0xFFFFFFF0u; }();
  return v;
}

// Split encode (hsrle_mono_encode.hip.h): containers with too few blocks for one lane each are encoded chunk by chunk -- the blocks cut behind
// runs every encoder state stores, like a monolithic stream, with the block starts as forced cuts.  kSplitPieces pieces (= cut finder lanes) per block.
constexpr uint32_t kSplitEncodeBelow = 131072u;   // blocks: from here on one lane per block fills the device
constexpr uint32_t kSplitPieces = 4u, kSplitPiecesMax = 8u;             // (the workspace is planned for the larger number: split_pieces())
static bool split_encode_applies(int codec, uint64_t nBlocks, uint32_t B);
static bool split_codec_small(int codec);

struct Workspace
{
  uint64_t nBlocks, chunk, nChunks, t1, t2, t3;
  // split encode: cut finder results per piece, chunk table, sizes / offsets per chunk, lists (8 words per chunk), staging slots
  uint64_t spCutPos, spCutSym, spFlags, spIdx, spStarts, spSyms, spSlotOff, spSizes, spChunkOff, spFirst, spCtrl, spGuess, spListOut, spSlots, spL1, spL2, spL3, spPieces, spMaxChunks, spPick, spJobs, spJobCap;
  uint64_t offSlots, offSizes, offL1, offL2, offL3, total;
};

// splitSmall: reserve the split-encode regions for blocks of 1 .. 4 KiB too (the codecs without a run list encoder -- 8 bit Single, 128 bit --
// take the split encode there: split_codec_small(); hsrle_compress_workspace_size_codec)
static Workspace plan_workspace(uint64_t U, uint32_t B, bool splitSmall = false, bool noSplit = false)
{
  Workspace w;
  w.nBlocks = block_count(U, B);
  w.chunk = encode_chunk_blocks();
  if (w.nBlocks < w.chunk + w.chunk / 2u) w.chunk = w.nBlocks;            // no short last chunk: up to 1.5 chunks go in one piece
  w.nChunks = (w.nBlocks + w.chunk - 1) / w.chunk;
  const uint64_t slotBlocks = w.nChunks > 1 ? 2u * w.chunk : w.nBlocks;
  w.t1 = (w.chunk + kScanTile - 1) / kScanTile;
  w.t2 = (w.t1 + kScanTile - 1) / kScanTile;
  w.t3 = (w.t2 + kScanTile - 1) / kScanTile;
  uint64_t at = 0;
  w.offSlots = at; at += align_up(slotBlocks * (uint64_t)slot_stride(B) + kTicketBytes, 256);    // (+ the wave encoder's counters: hsrle_encode8w.hip.h)
  w.offSizes = at; at += align_up(w.nBlocks * 4ull, 256);
  w.offL1 = at; at += align_up((w.t1 + 1) * 8ull, 256);
  w.offL2 = at; at += align_up((w.t2 + 1) * 8ull, 256);
  w.offL3 = at; at += align_up((w.t3 + 1) * 8ull, 256);
  // (split encode regions: sized for every codec, but only where the split path can run -- small containers of blocks above 4 KiB; up to
  //  4 KiB the run list encoders take every codec that has a chunk mode (split_encode_applies), and reserving the regions there cost
  //  3x the input for nothing: ADVICE r3.  Experiment builds can send those containers to the split path too, HSRLE_RUNLIST=2.)
  w.spPieces = w.nBlocks * kSplitPiecesMax; w.spMaxChunks = w.spPieces + w.nBlocks;
  w.spCutPos = w.spCutSym = w.spFlags = w.spIdx = w.spStarts = w.spSyms = w.spSlotOff = w.spSizes = w.spChunkOff = w.spFirst = w.spCtrl = w.spGuess = w.spListOut = w.spSlots = w.spL1 = w.spL2 = w.spL3 = 0;
  w.spPick = 0; w.spJobs = 0; w.spJobCap = 0;
  // (noSplit: a codec whose blocks above 4 KiB go to the windowed position-parallel encoders never takes the split encode -- its regions, 2.3 x the input, are not reserved)
  if (!noSplit && w.nBlocks < kSplitEncodeBelow && B >= 1024u && B <= (1u << 20) && (kExperiments || B > 4096u || splitSmall))
  {
    const uint64_t np = w.spPieces, nc = w.spMaxChunks;
    const uint64_t s1 = (nc + 2 + kScanTile - 1) / kScanTile, s2 = (s1 + kScanTile - 1) / kScanTile, s3 = (s2 + kScanTile - 1) / kScanTile;
    w.spCutPos = at; at += align_up(8ull * np, 256);
    w.spCutSym = at; at += align_up(8ull * np, 256);
    w.spFlags = at; at += align_up(4ull * np, 256);
    w.spIdx = at; at += align_up(8ull * (np + 1), 256);
    w.spStarts = at; at += align_up(8ull * (nc + 2), 256);
    w.spSyms = at; at += align_up(8ull * (nc + 1), 256);
    w.spSlotOff = at; at += align_up(8ull * (nc + 1), 256);
    w.spSizes = at; at += align_up(4ull * (nc + 1), 256);
    w.spChunkOff = at; at += align_up(8ull * (nc + 2), 256);
    w.spCtrl = at; at += 256;                                            // (the table of first chunks follows the control words directly: the ring encoders find it at ctrl + 64)
    w.spFirst = at; at += align_up(4ull * (w.nBlocks + 2), 256);
    w.spGuess = at; at += align_up(64ull * (nc + 1), 256);
    w.spListOut = at; at += align_up(64ull * (nc + 1), 256);
    w.spL1 = at; at += align_up(8ull * (s1 + 1), 256);
    w.spL2 = at; at += align_up(8ull * (s2 + 1), 256);
    w.spL3 = at; at += align_up(8ull * (s3 + 1), 256);
    w.spPick = at; at += align_up(w.nBlocks + 16ull, 256);               // 8 bit Single: a symbol per block
    w.spJobCap = (U >> 10) + nc + 16ull;                                  // ... and its copy jobs (literal stretches of >= 1 KiB go to whole waves: k_copy_jobs)
    w.spJobs = at; at += align_up(24ull * w.spJobCap, 256);
    w.spSlots = at; at += align_up(U + (U >> 7) + 256ull * (nc + 2) + 4096ull, 256);
  }
  w.total = at;
  return w;
}

// exclusive scan of `n` values (u32 at level 0) into out[0..n] (out[n] = total) using the pre-planned sum levels; `carry` as in k_tile_scan
static hipError_t scan_sizes(const uint32_t *sizes, uint64_t n, uint64_t *out, uint8_t *ws, const Workspace &w, hipStream_t st, const uint64_t *carry = nullptr)
{
  uint64_t *l1 = (uint64_t *)(ws + w.offL1), *l2 = (uint64_t *)(ws + w.offL2), *l3 = (uint64_t *)(ws + w.offL3);
  const uint64_t t1 = (n + kScanTile - 1) / kScanTile, t2 = (t1 + kScanTile - 1) / kScanTile, t3 = (t2 + kScanTile - 1) / kScanTile;
  if (carry == nullptr && n != 0 && n <= kScanSmallMax && (((uintptr_t)sizes) & 15u) == 0u)
  {
    // small tables (the split encode's flags and chunk sizes, small containers): one launch (k_scan_small_finish without the finish)
    hipLaunchKernelGGL(k_scan_small_finish, dim3((uint32_t)t1), dim3(kScanThreads), 0, st, sizes, (uint32_t)n, out, (uint8_t *)nullptr, 0u, 0ull, 0u);
    return hipGetLastError();
  }

  if (t1 > 1)
  {
    hipLaunchKernelGGL(k_tile_sums<uint32_t>, dim3((uint32_t)t1), dim3(kScanThreads), 0, st, sizes, n, l1);
    if (t2 > 1)
    {
      hipLaunchKernelGGL(k_tile_sums<uint64_t>, dim3((uint32_t)t2), dim3(kScanThreads), 0, st, l1, t1, l2);
      if (t3 > 1)
        return hipErrorInvalidValue; // > 2048^3 blocks: not representable anyway
      hipLaunchKernelGGL(k_tile_scan<uint64_t>, dim3(1), dim3(kScanThreads), 0, st, l2, t2, (const uint64_t *)nullptr, l3, 0, (const uint64_t *)nullptr);
      hipLaunchKernelGGL(k_tile_scan<uint64_t>, dim3((uint32_t)t2), dim3(kScanThreads), 0, st, l1, t1, l3, l1, 0, (const uint64_t *)nullptr);
    }
    else
    {
      hipLaunchKernelGGL(k_tile_scan<uint64_t>, dim3(1), dim3(kScanThreads), 0, st, l1, t1, (const uint64_t *)nullptr, l1, 0, (const uint64_t *)nullptr);
    }
    hipLaunchKernelGGL(k_tile_scan<uint32_t>, dim3((uint32_t)t1), dim3(kScanThreads), 0, st, sizes, n, l1, out, 1, carry);
  }
  else
  {
    hipLaunchKernelGGL(k_tile_scan<uint32_t>, dim3(1), dim3(kScanThreads), 0, st, sizes, n, (const uint64_t *)nullptr, out, 1, carry);
  }

  return hipGetLastError();
}

// the device's second stream (compaction behind the encode kernel); created once, shared by concurrent calls (their compactions then
// simply queue up behind each other)
static hipStream_t aux_stream()
{
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!D.aux && hipStreamCreateWithFlags(&D.aux, hipStreamNonBlocking) != hipSuccess) D.aux = nullptr;
  return D.aux;
}

static int compress_split(int codec, const uint8_t *dIn, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *ws, const Workspace &w, uint64_t *offsets, uint8_t *payload, hipStream_t st);

// the position-parallel encoder takes the containers of its codecs whose blocks it can hold (HSRLE_PP=1 / 2 in experiment builds: always / never)
constexpr uint32_t kPpMinBlocks = 1u;       // (faster than the ring and the run list encoders from 1 MiB to 8 GiB: experiments/r05, call 20)
static bool pp_applies(int codec, uint32_t nBlocks, uint32_t B)
{
  static const uint32_t force = knob_u32("HSRLE_PP", 0u);
  if (codec < 0 || codec >= kCodecCount || !g_pp[codec] || B > kPpMaxBlock || force == 2u) return false;
  // (the slot area holds the records: pp_scratch_bytes() is less than nBlocks staging slots for every block size)
  return force == 1u || nBlocks >= kPpMinBlocks;
}

// ... and, window by window, their containers of blocks above 4 KiB (hsrle_encode8pw.hip.h), whatever their size: 8 MiB of 8 KiB / 64 KiB blocks 19 / 51 us against
// 211 / 1 187 us of the split encode, 8 GiB 5.0 / 5.0 ms against 7.4 / 7.6 of the ring encoders + k_compact (1 MiB blocks: 5.5 against 20.7; experiments/r06, calls 22, 25)
#ifndef HSRLE_PPW_MIN_BLOCKS
#define HSRLE_PPW_MIN_BLOCKS 1u
#endif
constexpr uint32_t kPpwMinBlocks = HSRLE_PPW_MIN_BLOCKS;   // (A/B builds: 0xFFFFFFFF = never, also for the monolithic streams)
// words of a window's state record: the 8 bit pair 8, the codecs of hsrle_encodeSp.hip.h 16, those of hsrle_encodeLp.hip.h (the list travels with it) 32
static uint32_t ppw_state_words(int codec) { return codec <= 1 ? kPpwStateWords : (g_ppwL[codec] ? kPpwLStateWords : kPpwSStateWords); }
static PpwLaunch ppw_launcher(int codec) { return (codec < 0 || codec >= kCodecCount) ? nullptr : ((codec <= 1) ? g_ppw[codec] : (g_ppwL[codec] ? g_ppwL[codec] : g_ppwS[codec])); }
// (the plain / Packed codecs write 8 or 32 bit fields whatever the block size; the LUT / Short forms choose their field widths -- and the reference its penalties, with
//  thresholds of 0xFFFFF: rleX_Xsl.h:130, rleX_Xsl_short.h:178 -- by the values: below 1 MiB per block no count or range gets there and "every run is stored" holds)
constexpr uint32_t kPpwListMaxBlock = (1u << 20) - 128u;
static bool ppw_applies(int codec, uint32_t nBlocks, uint32_t B)
{
  static const uint32_t force = knob_u32("HSRLE_PP", 0u);
  if (!ppw_launcher(codec) || B <= kPpMaxBlock || force == 2u) return false;
  const bool anyBlock = codec <= 1 || (codec >= 6 && codec < 46 && ((codec - 6) & 2) == 0);                   // 8 bit multi, plain / Packed of 2 .. 8 byte symbols
  if (!anyBlock && B > kPpwListMaxBlock) return false;
  if ((uint64_t)nBlocks * ((B + kPpwWindow - 1u) / kPpwWindow) > 0xFFFFFFF0ull) return false;            // (windows are counted in 32 bits: 16 TiB)
  return force == 1u || nBlocks >= kPpwMinBlocks;
}

static int compress_async(int codec, const void *dIn, uint64_t U, void *dOut, uint64_t cap, uint32_t B, void *dWs, uint64_t wsSize, hipStream_t st, bool noSplit = false)
{
  if (codec < 0 || codec >= kCodecCount || dIn == nullptr || dOut == nullptr || U == 0)
    return HSRLE_ERR_ARGUMENT;
  if (B == 0) B = HSRLE_DEFAULT_BLOCK_SIZE;
  if (!valid_block_size(B))
    return HSRLE_ERR_ARGUMENT;
  if (cap < hsrle_container_bound(U, B))
    return HSRLE_ERR_CAPACITY;
  if (block_count(U, B) > 0xFFFFFFF0ull)
    return HSRLE_ERR_ARGUMENT;
  if (!device_ok())
    return HSRLE_ERR_DEVICE;

  init_tables();
  if (!g_enc[codec])
    return HSRLE_ERR_UNSUPPORTED;

  // (8 bit Single / 128 bit, small containers of 1 .. 4 KiB blocks: the split encode needs regions the general workspace does not reserve --
  //  the library's own scratch has them, a caller's workspace if it was sized by hsrle_compress_workspace_size_codec)
  Workspace w = plan_workspace(U, B, false, ppw_applies(codec, (uint32_t)block_count(U, B), B));
  if (!noSplit && split_codec_small(codec) && B <= 4096u && !pp_applies(codec, (uint32_t)w.nBlocks, B))
  {
    // (only where the split encode will really run: its regions are 2 - 3 x the input -- ADVICE r4)
    const Workspace w2 = plan_workspace(U, B, true);
    if (w2.spSlots != 0 && split_encode_applies(codec, w2.nBlocks, B) && (dWs == nullptr || wsSize >= w2.total)) w = w2;
  }

  void *own = nullptr;
  if (dWs == nullptr)
  {
    own = scratch_alloc(w.total, st);                    // freed (stream-ordered) behind the last kernel below
    if (!own)
      return HSRLE_ERR_DEVICE;
    dWs = own;
  }
  else if (wsSize < w.total)
    return HSRLE_ERR_CAPACITY;

  uint8_t *ws = (uint8_t *)dWs;
  uint8_t *container = (uint8_t *)dOut;
  const uint32_t nBlocks = (uint32_t)w.nBlocks;
  uint64_t *offsets = (uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE);
  uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)nBlocks + 1ull);
  const uint32_t stride = slot_stride(B);
  uint32_t *sizes = (uint32_t *)(ws + w.offSizes);
  int rc = HSRLE_OK;
  bool finished = false;                                                // (small containers: the size scan's launch wrote header and tail pad)

  // HSRLE_ENCODE_WAVE=1: rle8_multi / rle8_packed_multi with blocks of at most 4 KiB by ONE WAVE PER BLOCK (hsrle_encode8w.hip.h: position
  // parallel run detection, offsets by decoupled look-back, the payload written once, no staging slots and no compaction pass).  Bit-exact
  // (every block of the 8 GiB buffer against the reference manifest), but OFF by default: at 4 KiB per wave the kernel is bound by the
  // latency chain of one block, not by traffic -- 8 GiB: 13.5 ms against 7.4 ms of the lane-per-block kernel + compaction (input load + masks
  // 2.1 ms, run list 0.7, decisions 2.1, packets 2.6, look-back 1.4 .. 5.9 ms).
  hipStream_t aux = w.nChunks > 1 ? aux_stream() : nullptr;
#ifdef HSRLE_EXPERIMENTS
  static const bool waveEncode = env_u32("HSRLE_ENCODE_WAVE", 0) != 0u;
  if (waveEncode && g_wenc[codec] && B <= kWaveEncodeMaxBlock)
  {
    // (the slot area of the lane-per-block path is not needed: the tile words and the ticket counters live at its start)
    uint32_t *ticket = (uint32_t *)(ws + w.offSlots);
    unsigned long long *tiles = (unsigned long long *)(ws + w.offSlots + kTicketBytes);
    WaveEncodeArgs wa{ (const uint8_t *)dIn, U, B, nBlocks, offsets, payload, tiles, ticket };
    if (w.offSizes - w.offSlots < kTicketBytes + 8ull * nBlocks) rc = HSRLE_ERR_CAPACITY;
    else if (zero_async(ticket, kTicketBytes + 8ull * nBlocks, st) != hipSuccess || g_wenc[codec](wa, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
  }
  else
#endif
  if (w.nChunks <= 1 && pp_applies(codec, nBlocks, B))
  {
    // position-parallel encoder (hsrle_encode8p.hip.h): sizes + one record per stored run, scan, then every stream written once to its final place --
    // no staging slots, no compaction.  The records live in the (otherwise unused) slot area.
    PpArgs pa{ (const uint8_t *)dIn, U, B, nBlocks, sizes, offsets, payload, ws + w.offSlots };
    if (g_pp[codec](pa, 0, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    else if (nBlocks <= kScanSmallMax && (((uintptr_t)sizes) & 15u) == 0u)   // (the one-launch scan reads the sizes 16 bytes at a time: a caller's workspace may sit anywhere)
    {
      hipLaunchKernelGGL(k_scan_small_finish, dim3((nBlocks + kScanTile - 1u) / kScanTile), dim3(kScanThreads), 0, st, (const uint32_t *)sizes, nBlocks, offsets, container, (uint32_t)codec, U, B);
      finished = true;
    }
    else if (scan_sizes(sizes, nBlocks, offsets, ws, w, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    if (rc == HSRLE_OK && g_pp[codec](pa, 1, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
  }
  else if (w.nChunks <= 1 && ppw_applies(codec, nBlocks, B))
  {
    // the same encoder, a block walked in windows of 4 KiB: a wave per block leaves sizes, window states and records; after the scan a wave per window writes
    // the packets of the runs that end in it.  States and records live in the slot area (1 056 bytes per window).
    PpwArgs pa{};
    pa.in = (const uint8_t *)dIn; pa.U = U; pa.B = B; pa.nUnits = nBlocks; pa.sizes = sizes; pa.offsets = offsets; pa.payload = payload;
    pa.nWindows = nBlocks * ((B + kPpwWindow - 1u) / kPpwWindow);
    pa.states = (uint32_t *)(ws + w.offSlots);
    pa.recs = (uint32_t *)(ws + w.offSlots + align_up(4ull * ppw_state_words(codec) * pa.nWindows, 256));
    const PpwLaunch launch = ppw_launcher(codec);
    if (launch(pa, 0, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    else if (nBlocks <= kScanSmallMax && (((uintptr_t)sizes) & 15u) == 0u)
    {
      hipLaunchKernelGGL(k_scan_small_finish, dim3((nBlocks + kScanTile - 1u) / kScanTile), dim3(kScanThreads), 0, st, (const uint32_t *)sizes, nBlocks, offsets, container, (uint32_t)codec, U, B);
      finished = true;
    }
    else if (scan_sizes(sizes, nBlocks, offsets, ws, w, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    if (rc == HSRLE_OK && launch(pa, 1, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
  }
  else if (w.nChunks <= 1 && w.spSlots != 0 && split_encode_applies(codec, w.nBlocks, B))
  {
    rc = compress_split(codec, (const uint8_t *)dIn, U, B, nBlocks, ws, w, offsets, payload, st);
    finished = true;                                                    // (k_split_finish wrote header and tail pad)
  }
  else if (w.nChunks > 1 && aux == nullptr)
    rc = HSRLE_ERR_DEVICE;
  else if (w.nChunks <= 1)
  {
    EncodeArgs ea{ (const uint8_t *)dIn, U, B, nBlocks, ws + w.offSlots, stride, sizes };
    ea.ringSel = (uint32_t *)(ws + w.offSlots + align_up((uint64_t)nBlocks * stride, 256));   // (in the wave encoder's counter area behind the slots: unused on this path)
    if (g_enc[codec](ea, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    else if (nBlocks <= kScanSmallMax && (((uintptr_t)sizes) & 15u) == 0u)   // (the one-launch scan reads the sizes 16 bytes at a time: a caller's workspace may sit anywhere)
    {
      hipLaunchKernelGGL(k_scan_small_finish, dim3((nBlocks + kScanTile - 1u) / kScanTile), dim3(kScanThreads), 0, st, (const uint32_t *)sizes, nBlocks, offsets, container, (uint32_t)codec, U, B);
      finished = true;
    }
    else if (scan_sizes(sizes, nBlocks, offsets, ws, w, st) != hipSuccess)
      rc = HSRLE_ERR_DEVICE;
    if (rc == HSRLE_OK)
      hipLaunchKernelGGL(k_compact, dim3((nBlocks + 3u) / 4u), dim3(256), 0, st, ea.slots, stride, (const uint64_t *)offsets, payload, nBlocks);
  }
  else
  {
    // chunk k: encode + size scan on the caller's stream, compaction on the second stream; the slot buffer of chunk k is free again when
    // the compaction of chunk k - 2 is done
    hipEvent_t scanned = nullptr, compacted[2] = { nullptr, nullptr };
    bool ok = hipEventCreateWithFlags(&scanned, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&compacted[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&compacted[1], hipEventDisableTiming) == hipSuccess;
    for (uint64_t k = 0; k < w.nChunks && ok; k++)
    {
      const uint64_t first = k * w.chunk;
      const uint32_t count = (uint32_t)((w.nBlocks - first) < w.chunk ? (w.nBlocks - first) : w.chunk);
      uint8_t *slots = ws + w.offSlots + (k & 1u) * w.chunk * (uint64_t)stride;
      if (k >= 2) ok = hipStreamWaitEvent(st, compacted[k & 1u], 0) == hipSuccess;
      EncodeArgs ea{ (const uint8_t *)dIn + first * B, U - first * B, B, count, slots, stride, sizes + first };
      ok = ok && g_enc[codec](ea, st) == hipSuccess && scan_sizes(sizes + first, count, offsets + first, ws, w, st, k ? offsets + first : nullptr) == hipSuccess &&
           hipEventRecord(scanned, st) == hipSuccess && hipStreamWaitEvent(aux, scanned, 0) == hipSuccess;
      if (!ok) break;
      hipLaunchKernelGGL(k_compact, dim3((count + 3u) / 4u), dim3(256), 0, aux, (const uint8_t *)slots, stride, (const uint64_t *)(offsets + first), payload, count);
      ok = hipEventRecord(compacted[k & 1u], aux) == hipSuccess;
    }
    // the caller's stream continues when every compaction is done
    ok = ok && hipStreamWaitEvent(st, compacted[0], 0) == hipSuccess && hipStreamWaitEvent(st, compacted[1], 0) == hipSuccess;
    if (scanned) (void)hipEventDestroy(scanned);
    if (compacted[0]) (void)hipEventDestroy(compacted[0]);
    if (compacted[1]) (void)hipEventDestroy(compacted[1]);
    if (!ok) rc = HSRLE_ERR_DEVICE;
  }
  if (rc == HSRLE_OK && !finished)
  {
    hipLaunchKernelGGL(k_finish_container, dim3(1), dim3(64), 0, st, container, (uint32_t)codec, U, B, nBlocks);
    if (hipGetLastError() != hipSuccess) rc = HSRLE_ERR_DEVICE;
  }
  scratch_free(own, st);
  return rc;
}

static int check_info(const ContainerHeader &h, uint64_t containerSize, hsrle_container_info_t *info)
{
  static const char m[8] = { 'H', 'S', 'R', 'L', 'E', 'K', 'I', 'T' };
  if (memcmp(h.magic, m, 8) != 0 || h.version != 1 || h.codec >= (uint32_t)kCodecCount)
    return HSRLE_ERR_FORMAT;
  if (!valid_block_size(h.blockSize) || h.uncompressedSize == 0)
    return HSRLE_ERR_FORMAT;
  if (h.blockCount != block_count(h.uncompressedSize, h.blockSize))
    return HSRLE_ERR_FORMAT;
  const uint64_t payloadStart = HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)h.blockCount + 1ull);
  if (h.totalSize != payloadStart + h.payloadSize + HSRLE_CONTAINER_TAIL_PAD || h.totalSize > containerSize)
    return HSRLE_ERR_FORMAT;
  info->version = h.version;
  info->codec = h.codec;
  info->uncompressedSize = h.uncompressedSize;
  info->blockSize = h.blockSize;
  info->blockCount = h.blockCount;
  info->payloadSize = h.payloadSize;
  info->totalSize = h.totalSize;
  return HSRLE_OK;
}

static int decompress_blocks_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t first, uint32_t count, void *dOut, uint64_t cap,
                                   uint32_t *dStatus, hipStream_t st)
{
  if (dContainer == nullptr || info == nullptr || dOut == nullptr)
    return HSRLE_ERR_ARGUMENT;
  if (info->codec >= (uint32_t)kCodecCount || !valid_block_size(info->blockSize) || info->blockCount != block_count(info->uncompressedSize, info->blockSize))
    return HSRLE_ERR_FORMAT;
  if ((uint64_t)first + count > info->blockCount)
    return HSRLE_ERR_ARGUMENT;
  if (cap < info->uncompressedSize)
    return HSRLE_ERR_CAPACITY;
  if (!device_ok())
    return HSRLE_ERR_DEVICE;
  if (count == 0)
    return HSRLE_OK;

  init_tables();
  if (!g_dec[info->codec])
    return HSRLE_ERR_UNSUPPORTED;

  const uint8_t *container = (const uint8_t *)dContainer;
  const uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)info->blockCount + 1ull);
  DecodeArgs da{ payload, (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE), payload + info->payloadSize + HSRLE_CONTAINER_TAIL_PAD,
                 (uint8_t *)dOut, info->uncompressedSize, info->blockSize, first, count, dStatus };
  return g_dec[info->codec](da, st) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------------------------------
// split decode of a container: every block is cut into sub-blocks of SB output bytes.  One lane per block walks the block's packets
// (k_container_records) and leaves the decoder state at every SB bytes; the block kernel then runs one lane per SUB-block.  For
// containers with too few blocks to fill the GPU with one lane per block (an 88 MB frame in 4 KiB blocks has 21 600): the block size,
// and with it the compression ratio, stays what it is.

constexpr uint32_t kPacketListBelow = 81920u;     // blocks: up to here the packet list decode is the library's choice for small containers
static uint32_t split_sub_block(const hsrle_container_info_t *info, uint32_t want)
{
  const uint32_t B = info->blockSize;
  // the packet-list decode (hsrle_index.hip.h: k_container_packets, hsrle_expand.hip.h): the library's choice for blocks of up to 16 KiB --
  // the 88 MB frame: 102 us against 180 with records at every 1 KiB and 322 with one lane per block.  It wins up to ~65 536 blocks (256 MiB
  // in 4 KiB blocks: 183 / 186 us video-shaped rle64_3symlut_byte / run-distributed rle8_packed against 329 / 212 plain) and loses at 131 072
  // (408 / 591 against 361 / 274): from 81 920 blocks on one lane per block has the waves it needs.  Records were never the fastest of the
  // three at any size (experiments/r04/call24.sh); they remain for blocks above 16 KiB, whose offsets do not fit the list entries.
  const bool listFits = B <= kPacketListMaxBlock && B >= 256u;
  if (want == HSRLE_SPLIT_PACKET_LIST)
    return listFits ? HSRLE_SPLIT_PACKET_LIST : B;
  if (want == 0u && listFits)
    return info->blockCount <= kPacketListBelow ? HSRLE_SPLIT_PACKET_LIST : B;
  if (want == 0u)
  {
    want = pow2_floor(info->uncompressedSize >> 16);                 // ~2^16 .. 2^17 lanes (measured on the 88 MB frame: 1 KiB sub-blocks 187 us, 512 / 256 bytes 206)
    want = want < 256u ? 256u : want;
  }
  if (want >= B || (want % 128u) != 0u || (B % want) != 0u)
    return B;                                                           // no split
  return want;
}

static int decompress_split_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t first, uint32_t count, void *dOut, uint64_t cap, uint32_t *dStatus,
                                  void *dWs, uint64_t wsSize, uint32_t subBlock, hipStream_t st)
{
  if (dContainer == nullptr || info == nullptr || dOut == nullptr)
    return HSRLE_ERR_ARGUMENT;
  if (info->codec >= (uint32_t)kCodecCount || !valid_block_size(info->blockSize) || info->blockCount != block_count(info->uncompressedSize, info->blockSize))
    return HSRLE_ERR_FORMAT;
  const uint32_t SB = split_sub_block(info, subBlock);
  if (SB == info->blockSize)
    return decompress_blocks_async(dContainer, info, first, count, dOut, cap, dStatus, st);
  if ((uint64_t)first + count > info->blockCount)
    return HSRLE_ERR_ARGUMENT;
  if (cap < info->uncompressedSize)
    return HSRLE_ERR_CAPACITY;
  if (!device_ok())
    return HSRLE_ERR_DEVICE;
  if (count == 0)
    return HSRLE_OK;
  init_tables();
  if (!g_dec[info->codec] || !g_sub[info->codec])
    return HSRLE_ERR_UNSUPPORTED;

  if (SB == HSRLE_SPLIT_PACKET_LIST)
  {
    const uint64_t countsBytes = packet_list_counts_bytes(count);
    if (dWs == nullptr || ((uintptr_t)dWs & 15u) != 0u || wsSize < countsBytes + (uint64_t)count * packet_list_cap(info->blockSize) * 8ull)
      return HSRLE_ERR_CAPACITY;
    const uint8_t *container = (const uint8_t *)dContainer;
    const uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)info->blockCount + 1ull);
    DecodeArgs da{ payload, (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE), payload + info->payloadSize + HSRLE_CONTAINER_TAIL_PAD,
                   (uint8_t *)dOut, info->uncompressedSize, info->blockSize, first, count, dStatus };
    if (g_sub[info->codec](da, kSubPacketList, (uint32_t *)dWs, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    return launch_expand_packets(codec_symbol_bytes((int)info->codec), da, (const uint64_t *)((const uint8_t *)dWs + countsBytes), (const uint32_t *)dWs, st) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
  }

  const uint32_t per = info->blockSize / SB;
  const uint64_t subFirst = (uint64_t)first * per;
  const uint64_t subAll = (info->uncompressedSize + SB - 1u) / SB;
  uint64_t subEnd = ((uint64_t)first + count) * per;
  if (subEnd > subAll) subEnd = subAll;
  if (subEnd - subFirst > 0xFFFFFFF0ull || subEnd > 0xFFFFFFF0ull)
    return HSRLE_ERR_ARGUMENT;
  if (dWs == nullptr || wsSize < (subEnd - subFirst) * 4ull * kEntryRecDwords)
    return HSRLE_ERR_CAPACITY;

  const uint8_t *container = (const uint8_t *)dContainer;
  const uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)info->blockCount + 1ull);
  DecodeArgs da{ payload, (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE), payload + info->payloadSize + HSRLE_CONTAINER_TAIL_PAD,
                 (uint8_t *)dOut, info->uncompressedSize, info->blockSize, first, count, dStatus };
  // records are indexed by the global sub-block number: hand the kernel the address record 0 would have
  uint32_t *rec0 = (uint32_t *)((uintptr_t)dWs - (uintptr_t)(subFirst * 4ull * kEntryRecDwords));
  if (g_sub[info->codec](da, SB, rec0, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  da.B = SB; da.firstBlock = (uint32_t)subFirst; da.blockCount = (uint32_t)(subEnd - subFirst);
  da.entries = (const uint32_t *)dWs; da.entryBase = (uint32_t)subFirst;
  return g_dec[info->codec](da, st) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

// one WAVE per block (experiments/hsrle_decode_wave.hip.h): for containers with too few blocks to fill the chip with one lane each
constexpr uint32_t kWaveDecodeBelow = 65536u;     // blocks: above, one lane per block has the waves it needs
static int decompress_wave_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t first, uint32_t count, void *dOut, uint64_t cap, uint32_t *dStatus, hipStream_t st)
{
  if (dContainer == nullptr || info == nullptr || dOut == nullptr)
    return HSRLE_ERR_ARGUMENT;
  if (info->codec >= (uint32_t)kCodecCount || !valid_block_size(info->blockSize) || info->blockCount != block_count(info->uncompressedSize, info->blockSize))
    return HSRLE_ERR_FORMAT;
  if (info->blockSize > 16384u || !kExperiments)                        // (the wave-per-block decoder is not part of the shipped build: measured slower than the split decode)
    return HSRLE_ERR_UNSUPPORTED;
  if ((uint64_t)first + count > info->blockCount)
    return HSRLE_ERR_ARGUMENT;
  if (cap < info->uncompressedSize)
    return HSRLE_ERR_CAPACITY;
  if (!device_ok())
    return HSRLE_ERR_DEVICE;
  if (count == 0)
    return HSRLE_OK;
  init_tables();
  if (!g_sub[info->codec])
    return HSRLE_ERR_UNSUPPORTED;
  const uint8_t *container = (const uint8_t *)dContainer;
  const uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)info->blockCount + 1ull);
  DecodeArgs da{ payload, (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE), payload + info->payloadSize + HSRLE_CONTAINER_TAIL_PAD,
                 (uint8_t *)dOut, info->uncompressedSize, info->blockSize, first, count, dStatus };
  return g_sub[info->codec](da, 0u, nullptr, st) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;     // (sub-block size 0 = the wave decoder)
}

// ------------------------------------------------------------------------------------------------------------------
// monolithic stream decode: index passes (hsrle_index.hip.h) + the block kernel started from entry records

// symbol-state slots of the codec's decoder (IndexState<FAM>::KE): 0 plain / Single / 0-symbol Short, 1 Packed / 1-symbol list, 3, 7
static int codec_state_slots(int c)
{
  static const int byVariant[8] = { 0, 1, 3, 7, 0, 1, 3, 7 };
  if (c < 6) { static const int k8[6] = { 0, 1, 3, 7, 0, 1 }; return k8[c]; }
  if (c < 46) return byVariant[(c - 6) & 7];
  if (c < 50) return (c & 1) ? 1 : 0;                      // 46 sym, 47 sym_packed, 48 byte, 49 byte_packed
  if (c < kShortBaseW) return byVariant[c - kShortBase8];
  if (c < kGreedyBase) return byVariant[(c - kShortBaseW) & 7];
  if (c < kSingleShort) { static const int kg[3] = { 1, 3, 7 }; return kg[(c - kGreedyBase) % 3]; }
  return 0;
}

struct MonoPlan
{
  uint32_t G, M, B, R, KE;
  uint64_t nb;
  uint64_t offG, offE, offOlen, offT, offEntry, offOutStart, offStateIn, offFix, offList, offMark, offCtrl, offRec, offFast, offBatch, total;
  bool range7;
};


// (atomics: hsrle_mono_tuning() is a TEST knob and process-global -- a call that changes it between another thread's *_workspace_size() and
//  *_mono_dev() can make that workspace too small, which that call reports as HSRLE_ERR_CAPACITY; include/hsrle.h says so)
static std::atomic<uint32_t> g_monoTune[3] = { { env_u32("HSRLE_MONO_BLOCK", 0) }, { env_u32("HSRLE_MONO_REGION", 0) }, { env_u32("HSRLE_MONO_LOOKBACK", 0) } };

static MonoPlan plan_mono(int codec, uint32_t U, uint32_t C, uint32_t p0)
{
  MonoPlan m;
  // output bytes per decode lane: enough lanes to fill the GPU (>= 2^18 where the stream allows it), at most the container's 4 KiB
  uint32_t B = pow2_floor((uint64_t)U >> 18);
  B = B < 256u ? 256u : (B > 4096u ? 4096u : B);
  // stream bytes per index lane, and the look-back of its entry guess
  uint32_t G = pow2_floor((uint64_t)C >> 15);
  G = G < 2048u ? 2048u : (G > 8192u ? 8192u : G);
  const uint32_t tB = g_monoTune[0], tG = g_monoTune[1], tM = g_monoTune[2];   // tuning / test knobs (hsrle_mono_tuning, HSRLE_MONO_* in the environment)
  if (tB >= 128u && tB <= (1u << 20) && (tB % 128u) == 0u) B = tB;
  if (tG >= 32u && tG <= (1u << 24)) G = tG;
  m.B = B; m.G = G;
  // Look-back of the entry guess.  Formats with the 7-bit-or-4-byte range field (8 bit Packed, byte-aligned Packed) kill a walk that
  // starts at a wrong byte within a few hops (every other junk range byte claims a 4-byte literal count that points outside the
  // stream), so 1 KiB in front of a region is plenty.  The other formats' junk walks live on and only find the chain by falling onto
  // one of its packet starts (1 hop in ~40 on random literals): they start with 4 KiB, and mono_decode_dev widens the look-back when
  // too many guesses turn out wrong.
  const bool range7 = (codec == 1 || (codec >= 6 && codec < 50 && (codec == 49 || (codec < 46 && ((codec - 6) & 7) == 5))));
  // (range7 formats: regions of at most 4 KiB -- their guesses hold with a 2 KiB look-back, and the walk is one latency chain per region:
  //  the 1 GiB stream 1.47 -> 1.32 ms with 142 191 regions instead of 71 096, none guessed wrong; round 4, since the resolve pass scales)
  if (range7 && G > 4096u && !(tG >= 32u && tG <= (1u << 24))) { G = 4096u; m.G = G; }
  // (range7: 1 KiB leaves ~1 wrong guess in 7 000 on random literals, and each wrong guess costs a repair walk + a second resolve pass:
  //  2 KiB -- none in 71 096 -- where the regions are large enough to carry it: 1 GiB stream 2.28 -> 1.79 ms)
  m.M = tM ? tM : (range7 ? (G >= 4096u ? 2048u : 1024u) : 4096u);
  m.range7 = range7;
  m.R = (uint32_t)(((uint64_t)(C - p0) + G - 1u) / G);
  if (m.R == 0u) m.R = 1u;
  m.KE = (uint32_t)codec_state_slots(codec);
  m.nb = ((uint64_t)U + B - 1u) / B;
  const uint64_t ks = m.KE ? m.KE : 1u;
  uint64_t at = 0;
  m.offG = at; at += align_up(4ull * m.R, 256);
  m.offE = at; at += align_up(4ull * m.R, 256);
  m.offOlen = at; at += align_up(8ull * m.R, 256);
  m.offT = at; at += align_up(4ull * m.R * ks, 256);
  m.offEntry = at; at += align_up(4ull * m.R, 256);
  m.offOutStart = at; at += align_up(8ull * m.R, 256);
  m.offStateIn = at; at += align_up(4ull * m.R * ks, 256);
  m.offFix = at; at += align_up(4ull * m.R, 256);
  m.offList = at; at += align_up(4ull * m.R, 256);
  m.offMark = at; at += align_up(4ull * m.R, 256);        // (mark | ctrl | rec stay neighbours in this order: mono_prepare clears [offMark, offFast) in one launch)
  m.offCtrl = at; at += 256;
  m.offRec = at; at += align_up(4ull * kEntryRecDwords * m.nb, 256);
  m.offFast = at; at += 256;                                             // the parallel resolve passes: flag + carries, totals per batch of 1 024 regions
  m.offBatch = at; at += align_up(4ull * kFastBatchWords * ((uint64_t)m.R / kResolveThreads + 1ull), 256);
  m.total = at;
  return m;
}

__global__ void k_set_word(uint32_t *p, uint32_t v) { *p = v; }

static hipError_t launch_resolve(const MonoPlan &m, uint8_t *ws, uint32_t p0, uint64_t U, uint32_t roundTag, hipStream_t st)
{
  // the full batches but the last in parallel when every guess is right (hsrle_index.hip.h: k_resolve_fast_*); k_index_resolve finishes -- or, when
  // a region failed the check, does everything
  const uint32_t fastBatches = (m.R > 2u * (uint32_t)kResolveThreads) ? (m.R - 1u) / (uint32_t)kResolveThreads : 0u;
  uint32_t *fast = (uint32_t *)(ws + m.offFast), *batch = (uint32_t *)(ws + m.offBatch);
  const uint32_t *cg = (const uint32_t *)(ws + m.offG), *ce = (const uint32_t *)(ws + m.offE), *ct = (const uint32_t *)(ws + m.offT);
  const uint64_t *col = (const uint64_t *)(ws + m.offOlen);
  if (fastBatches != 0u) hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, fast, 1u);      // (a kernel, not a memset node: see mono_prepare)
#define HSRLE_RESOLVE(KE)                                                                                                                                        \
  if (fastBatches != 0u)                                                                                                                                         \
  {                                                                                                                                                              \
    hipLaunchKernelGGL(k_resolve_fast_totals<KE>, dim3(fastBatches), dim3(kResolveThreads), 0, st, cg, ce, col, ct, p0, m.G, fast, batch);                       \
    hipLaunchKernelGGL(k_resolve_fast_carries<KE>, dim3(1), dim3(kResolveThreads), 0, st, fast, batch, fastBatches);                                             \
    hipLaunchKernelGGL(k_resolve_fast_emit<KE>, dim3(fastBatches), dim3(kResolveThreads), 0, st, cg, col, ct, (const uint32_t *)fast, (const uint32_t *)batch,    \
                       (uint32_t *)(ws + m.offEntry), (uint64_t *)(ws + m.offOutStart), (uint32_t *)(ws + m.offStateIn));                                        \
  }                                                                                                                                                              \
  hipLaunchKernelGGL(k_index_resolve<KE>, dim3(1), dim3(kResolveThreads), 0, st, cg, ce, col, ct, m.R, p0, m.G, U, (uint32_t *)(ws + m.offEntry),               \
                     (uint64_t *)(ws + m.offOutStart), (uint32_t *)(ws + m.offStateIn), (uint32_t *)(ws + m.offFix), (uint32_t *)(ws + m.offList),                \
                     (uint32_t *)(ws + m.offCtrl), (uint32_t *)(ws + m.offMark), roundTag, fastBatches ? (const uint32_t *)fast : (const uint32_t *)nullptr, fastBatches)
  switch (m.KE)
  {
  case 0: HSRLE_RESOLVE(0); break;
  case 1: HSRLE_RESOLVE(1); break;
  case 3: HSRLE_RESOLVE(3); break;
  default: HSRLE_RESOLVE(7); break;
  }
#undef HSRLE_RESOLVE
  return hipGetLastError();
}

// what the first bytes of a stream say (reference: rle8_extreme_cpu.h:704-712, :759-760, rleX_extreme_cpu.h:84-91, rleX_Xsl.h:1850-1858)
struct MonoHeader
{
  uint32_t U, C, p0, single, singleSym;
  int codec;   // the id whose kernels decode it (Single mode streams of ids 0 / 1 -> ids 4 / 5)
};

static bool mono_header(int codec, const uint8_t *h16, uint32_t inSize, uint32_t outSize, MonoHeader *mh)
{
  if (codec < 0 || codec >= kCodecCount)
    return false;
  const uint32_t hs = codec_header_size(codec);
  if (inSize < hs)
    return false;
  memcpy(&mh->U, h16, 4);
  memcpy(&mh->C, h16 + 4, 4);
  if (mh->U > outSize || mh->C > inSize)
    return false;
  if (hs == 9 && h16[8] > 1) // unknown mode (rle8_extreme_cpu.h:759-760)
    return false;
  mh->single = 0; mh->singleSym = 0; mh->p0 = hs; mh->codec = codec;
  if (hs == 9 && h16[8] == 1)
  {
    // rle8_decompress / rle8_packed_decompress switch on the mode byte (rle8_extreme_cpu.h:702-764): Single mode -> the general kernel
    if (codec == HSRLE_RLE8_MULTI || codec == HSRLE_RLE8_PACKED_MULTI) mh->codec = codec + 4;
    mh->single = 1; mh->singleSym = h16[9]; mh->p0 = 10;
  }
  else if (codec == kSingleShort) { mh->singleSym = h16[8]; mh->p0 = 9; }   // rleX_Xsl_short.h:1211-1216
  if (mh->U == 0 || mh->C < mh->p0 + 2u || mh->C > 0x7FFFFF00u)
    return false;
  return true;
}

// ---- monolithic decode.  The passes in stream order: walk (every region from a guessed entry) -> resolve (chains the regions, checks the
//      guesses; verdict in ctrl[0..3]) -> records (decoder state at every B output bytes) -> decode.  Since round 5 the records pass is GATED on
//      the verdict on the device and the whole sequence is enqueued without the host in between: a stream whose guesses all hold (the normal
//      case) costs ONE host read at the end instead of two round trips (and none at all through hsrle_decompress_mono_dev_async, which a
//      HIP graph can capture); a stream that needs repair finds zero records, its decode lanes end at once, and the host-driven repair loop
//      takes over where the resolve pass stopped.
__global__ __launch_bounds__(256) void k_mono_clear(u32x4 *__restrict__ p, uint64_t n16)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n16) p[i] = u32x4{ 0, 0, 0, 0 };
}

struct MonoRun
{
  IndexArgs ia;
  uint32_t *ctrl;
  DecodeArgs da;
};

static int mono_prepare(const MonoHeader &mh, const uint8_t *dStream, uint8_t *dOut, uint8_t *ws, const MonoPlan &m, MonoRun *run, hipStream_t st)
{
  init_tables();
  if (!g_dec[mh.codec] || !g_idx[mh.codec])
    return HSRLE_ERR_UNSUPPORTED;
  uint32_t *ctrl = (uint32_t *)(ws + m.offCtrl);
  // mark | ctrl | records are neighbours in the workspace (plan_mono), every piece a multiple of 256 bytes: ONE clearing launch.  (Not
  // hipMemsetAsync, whose node misbehaves in a captured and replayed HIP graph -- hsrle_common.hip.h zero_async; seen again here: junk in ctrl[4..15].)
  {
    const uint64_t bytes = m.offFast - m.offMark;
    hipLaunchKernelGGL(k_mono_clear, dim3((uint32_t)((bytes / 16u + 255u) / 256u)), dim3(256), 0, st, (u32x4 *)(ws + m.offMark), bytes / 16u);
    if (hipGetLastError() != hipSuccess)
      return HSRLE_ERR_DEVICE;
  }

  IndexArgs ia{};
  ia.stream = dStream; ia.C = mh.C; ia.p0 = mh.p0; ia.G = m.G; ia.M = m.M; ia.R = m.R; ia.single = mh.single; ia.singleSym = mh.singleSym;
  ia.list = nullptr; ia.listCount = 0; ia.fix = (const uint32_t *)(ws + m.offFix);
  ia.g = (uint32_t *)(ws + m.offG); ia.e = (uint32_t *)(ws + m.offE); ia.olen = (uint64_t *)(ws + m.offOlen); ia.t = (uint32_t *)(ws + m.offT);
  ia.entry = (const uint32_t *)(ws + m.offEntry); ia.outStart = (const uint64_t *)(ws + m.offOutStart); ia.stateIn = (const uint32_t *)(ws + m.offStateIn);
  ia.U = mh.U; ia.B = m.B; ia.rec = (uint32_t *)(ws + m.offRec);
  ia.mark = (uint32_t *)(ws + m.offMark); ia.roundTag = 0;
  { static const uint32_t ext = env_u32("HSRLE_MONO_REPAIR_EXTEND", 48); ia.extMax = ext; }
  run->ia = ia;
  run->ctrl = ctrl;
  run->da = DecodeArgs{ dStream, nullptr, dStream + mh.C + HSRLE_CONTAINER_TAIL_PAD, dOut, mh.U, m.B, 0u, (uint32_t)m.nb, ctrl + 8 };
  run->da.entries = (const uint32_t *)(ws + m.offRec);
  run->da.entryBase = 0;
  return HSRLE_OK;
}

// walk of every region, resolve round 1, gated records, decode: nothing here waits for the host
static int mono_enqueue_first_try(const MonoHeader &mh, uint8_t *ws, const MonoPlan &m, MonoRun &run, hipStream_t st)
{
  if (g_idx[mh.codec](run.ia, 0, st) != hipSuccess || launch_resolve(m, ws, mh.p0, mh.U, 1u, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  run.ia.gate = run.ctrl;
  const hipError_t e = g_idx[mh.codec](run.ia, 1, st);
  run.ia.gate = nullptr;
  run.da.gate = run.ctrl;                                                // (the decode too: nothing to decode from records that were not written)
  const hipError_t e2 = e != hipSuccess ? e : g_dec[mh.codec](run.da, st);
  run.da.gate = nullptr;
  if (e2 != hipSuccess)
    return HSRLE_ERR_DEVICE;
  return HSRLE_OK;
}

// ctrl[0] regions whose guess failed, ctrl[1] malformed stream, ctrl[8] the decode kernel's error bits -> one word for the caller
__global__ void k_mono_status(const uint32_t *__restrict__ ctrl, uint32_t *__restrict__ status)
{
  *status = ctrl[1] != 0u ? (uint32_t)HSRLE_MONO_MALFORMED : (ctrl[0] != 0u ? (uint32_t)HSRLE_MONO_NEEDS_REPAIR : (ctrl[8] != 0u ? (uint32_t)HSRLE_MONO_MALFORMED : (uint32_t)HSRLE_MONO_DONE));
}

// dStream: 128-byte aligned, readable up to C + 64.  stats (optional): [0] regions, [1] repair rounds, [2] regions walked again.
// Synchronises the stream (once when every guess holds; the repair loop reads the resolve pass's verdict per round).
// Returns HSRLE_OK / HSRLE_ERR_FORMAT / HSRLE_ERR_DEVICE.
static int mono_decode_dev(const MonoHeader &mh, const uint8_t *dStream, uint8_t *dOut, uint8_t *ws, const MonoPlan &m, uint32_t *stats, hipStream_t st)
{
  MonoRun run;
  const int prc = mono_prepare(mh, dStream, dOut, ws, m, &run, st);
  if (prc != HSRLE_OK)
    return prc;
  IndexArgs &ia = run.ia;
  uint32_t *const ctrl = run.ctrl;
  if (!m.range7 && m.R >= 16384u && g_monoTune[2] == 0u)                  // (small streams: the pilot's launch + read costs more than a widened second try)
  {
    // formats whose junk walks do not die: does the short look-back find the chain on THIS stream?  A pilot over the first 128 regions
    // tells (data with little entropy synchronises within bytes, random literals need ~16 KiB): each wrong guess costs a repair later
    uint32_t pg[128], pe[128];
    IndexArgs pilot = ia;
    pilot.R = 128u;
    if (g_idx[mh.codec](pilot, 0, st) != hipSuccess || hipMemcpyAsync(pg, ia.g, sizeof(pg), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(pe, ia.e, sizeof(pe), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    uint32_t agree = 0;
    for (uint32_t r = 1; r < 128u; r++) agree += (pe[r - 1] == pg[r]) ? 1u : 0u;
    if (agree < 120u) ia.M = 16384u;
  }
  uint32_t rounds = 0, rewalked = 0, roundTag = 1;
  uint32_t verdict[12] = { 0 };                                            // [0..3] the resolve pass's verdict, [8] the decode kernel's status
  if (mono_enqueue_first_try(mh, ws, m, run, st) != HSRLE_OK ||
      hipMemcpyAsync(verdict, ctrl, 36, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (verdict[0] == 0u)
  {
    if (stats) { stats[0] = m.R; stats[1] = 0; stats[2] = 0; stats[3] = ia.M; }
    return (verdict[1] != 0u || verdict[8] != 0u) ? HSRLE_ERR_FORMAT : HSRLE_OK;
  }
  for (;;)
  {
    if (rounds++ > m.R)                                    // every round proves at least one more region: cannot happen
      return HSRLE_ERR_DEVICE;
    rewalked += verdict[0];
    if (rounds == 1u && verdict[0] > 8u && verdict[0] > m.R / 32u && ia.M < 65536u && g_monoTune[2] == 0u)
    {
      // the guesses of this stream do not find the chain within the look-back (wrong guesses come in streaks, and a streak is repaired
      // one region per round): guess again, everywhere, from four times as far back
      ia.M *= 4u;
      ia.list = nullptr; ia.listCount = 0;
      rounds = 0;
    }
    else { ia.list = (const uint32_t *)(ws + m.offList); ia.listCount = verdict[0]; ia.roundTag = roundTag; }
    if (g_idx[mh.codec](ia, 0, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    roundTag++;
    if (launch_resolve(m, ws, mh.p0, mh.U, roundTag, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    if (hipMemcpyAsync(verdict, ctrl, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    if (verdict[0] == 0u)
      break;
  }
  if (stats) { stats[0] = m.R; stats[1] = rounds; stats[2] = rewalked; stats[3] = ia.M; }
  if (verdict[1] != 0u)
    return HSRLE_ERR_FORMAT;

  // (the first try's decode lanes found zero records and left their error bits in the status word)
  ia.list = nullptr; ia.listCount = 0;
  hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, ctrl + 8, 0u);
  if (g_idx[mh.codec](ia, 1, st) != hipSuccess || g_dec[mh.codec](run.da, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  uint32_t status = 1;
  if (hipMemcpyAsync(&status, ctrl + 8, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  return status == 0u ? HSRLE_OK : HSRLE_ERR_FORMAT;
}

// ------------------------------------------------------------------------------------------------------------------
// monolithic stream encode by many lanes (hsrle_mono_encode.hip.h): cut behind long runs, block kernels in MONO mode, compaction

// the run length every state of the codec's encoder stores (SURVEY.md A.2 LONG / the Short family's SMINL); 0 = the codec is not cut
// (the Greedy encoders, rle8_single_short).  *pS / *pAligned: symbol bytes and sym-alignment of the codec.
static uint32_t mono_cut_long(int codec, int *pS = nullptr, int *pAligned = nullptr, int *pListK = nullptr)
{
  int S = 1, al = 0, K = 0;
  uint32_t longc = 0;
  static const int shortK[4] = { 0, 1, 3, 7 };
  if (codec == kSingleShort) longc = 27u;                        // rle8_single_short: runs of THE symbol of SMINL = 11 bytes are always stored; + 16, the body counts a run from where its search found it
  else if (codec == 4) longc = 8u;                              // 8 bit Single: runs of THE symbol with count >= LONG (rle8_extreme_cpu.h:10-11, :21-23)
  else if (codec == 5) longc = 10u;
  else if (codec == HSRLE_RLE8_MULTI) longc = 6u;               // rle8_extreme_cpu.h:974: count >= 6 whatever the range
  else if (codec == HSRLE_RLE8_PACKED_MULTI) longc = 11u;       // :978 (body) and :122 (tail)
  else if (codec == 2 || codec == 3) { longc = 11u; K = codec == 2 ? 3 : 7; }   // rle8 3 / 7 symbol LUT: rleX_Xsl.h:132 with S = 1
  else if (codec == HSRLE_RLE8_MULTI_SHORT) longc = 13u;        // rleX_Xsl_short.h: always stored from S + 12 on (0-symbol codec)
  else if (codec > kShortBase8 && codec < kShortBase8 + 4) { longc = 12u; K = shortK[codec - kShortBase8]; }   // ... from S + 11 on with a list
  else if (codec >= 46 && codec < 50) { S = 16; al = codec < 48 ? 1 : 0; longc = (codec == 47) ? 26u : 27u; }   // 128 bit (rle128_extreme_cpu.h:10-11; sym-aligned Packed: the hybrid of A.5 q10)
  else
  {
    static const int widths[5] = { 2, 3, 4, 6, 8 };
    if (codec >= 6 && codec < 46)
    {
      const int k = (codec - 6) & 7;
      S = widths[(codec - 6) >> 3];
      al = k < 4 ? 1 : 0;
      if (k == 0 || k == 4) longc = (uint32_t)S + 11u;                                  // plain: rleX_extreme_cpu.h:10-11
      else if (k == 1) longc = (uint32_t)S + 10u;                                       // sym-aligned Packed: the hybrid of A.5 q10
      else if (k == 5) longc = (uint32_t)S + 11u;                                       // byte-aligned Packed
      else { longc = (uint32_t)S + 10u; K = (k & 1) ? 7 : 3; }                          // LUT: rleX_Xsl.h:132
    }
    else if (codec >= kShortBaseW && codec < kGreedyBase)
    {
      const int k = (codec - kShortBaseW) & 7;
      S = widths[(codec - kShortBaseW) >> 3];
      al = k < 4 ? 1 : 0;
      if (k == 0 || k == 4) longc = (uint32_t)S + 12u;                                  // 0-symbol Short codecs
      else { longc = (uint32_t)S + 11u; K = shortK[k & 3]; }                            // 1 / 3 / 7 symbol Short codecs
    }
    else if (codec >= kGreedyBase && codec < kSingleShort)
    {
      // Greedy encoders (rleX_Xsl_short.h:746-1000): a run of SMINL = S + 11 bytes is stored whatever the state -- but the scan may enter a
      // periodic stretch up to ~2 S bytes late (through a prefix of a listed symbol), so a stretch is a cut from S + 11 + 3 S bytes on
      static const int kg[3] = { 1, 3, 7 };
      S = widths[(codec - kGreedyBase) / 3];
      al = 0;
      K = kg[(codec - kGreedyBase) % 3];
      longc = 4u * (uint32_t)S + 11u;
    }
  }
  if (pS) *pS = S;
  if (pAligned) *pAligned = al;
  if (pListK) *pListK = K;
  return longc;
}

// pieces per block (= cut finder lanes; a block has at most pieces + 1 chunks).  8 pieces were measured for the list codecs that settle their
// lists inside the encode kernel (88 MB frame / 64 MiB runs, us: rle64_3symlut_byte 4: 310 / 378, 8: 341 / 485; rle32_7symlut_sym 4: 411 / 355,
// 8: 409 / 328; rle16_3symlut_byte 4: 434 / 315, 8: 497 / 327): the kernel's time follows its total trips, not the trips per wave, so
// shorter chunks buy nothing; the per-lane chunk encoders (Single, 128 bit, Greedy with a one-symbol list), 4 / 8 pieces: rle8_single 285 / 290 and 379 / 365,
// rle128_sym 380 / 374 and 282 / 324, rle64_1symlut greedy 732 / 689 and 775 / 1029 (experiments/r04/call61.sh).  The workspace is planned for kSplitPiecesMax so that experiment builds can try (HSRLE_SPLIT_PIECES).
static uint32_t split_pieces(uint32_t B)
{
  const uint32_t k = knob_u32("HSRLE_SPLIT_PIECES", kSplitPieces);
  return (k == kSplitPiecesMax && B % (kSplitPiecesMax * 128u) == 0u) ? k : kSplitPieces;
}

// (round 6: the codecs with a position-parallel encoder have no run list instantiation any more -- hsrle_inst_generic.inc: pp_covers; callers run behind init_tables())
static bool run_list_codec(int codec) { return (codec <= 3 || (codec >= 6 && codec <= 45) || (codec >= kShortBase8 && codec < kGreedyBase)) && !g_pp[codec]; }

// the codecs that have a many-lane chunk encoder but no run list encoder: small containers of 1 .. 4 KiB blocks take the split encode IF the
// caller's workspace has its regions (hsrle_compress_workspace_size_codec; the library's own scratch always has)
static bool greedy_one_symbol_list(int codec) { return codec >= kGreedyBase && codec < kSingleShort && (codec - kGreedyBase) % 3 == 0; }
static bool split_codec_small(int codec) { return codec == 4 || codec == 5 || codec == kSingleShort || (codec >= 46 && codec < 50) || greedy_one_symbol_list(codec); }

static bool split_encode_applies(int codec, uint64_t nBlocks, uint32_t B)
{
  // the codecs whose ring encoders have the chunk mode, and (round 4) 8 bit Single and 128 bit with the per-lane chunk encoders of the monolithic path
  // rle8_multi / rle8_packed_multi / rle8_{3,7}symlut (ids 0 .. 3), the plain / Packed / LUT codecs of 2 .. 8 byte symbols (ids 6 .. 45) and their Short family (ids 50 .. 93): the run list encoders take these whole
  if (run_list_codec(codec) && run_list_applies(nBlocks, B, 1024u, knob_u32("HSRLE_RUNLIST", 0u))) return false;
  if (nBlocks >= kSplitEncodeBelow || B < 1024u || B > (1u << 20) || (B % (kSplitPieces * 128u)) != 0u) return false;
  // Greedy: the lists of 3 / 7 symbols decide which runs the scan stores, so a chunk's list is only known when the chunk in front of it is final --
  // measured (88 MB frame, rle32_7symlut: two full passes + three repair rounds 3.7 ms against 3.1 ms with one lane per block); a list of ONE symbol
  // behind a stored run is that run's symbol, and the first guess is right
  if (codec >= kGreedyBase && codec < kSingleShort && !greedy_one_symbol_list(codec)) return false;
  if ((codec == 4 || codec == 5 || codec == kSingleShort) && B > 32768u) return false;   // (the per-block symbol pick holds a block in LDS: hsrle_encode8s.hip.h)
  init_tables();
  return mono_cut_long(codec) != 0u && g_menc[codec] != nullptr;
}

// Split encode of a container (no host synchronisation: graph capturable like the plain path): cuts inside the blocks -> chunk table (block
// starts are forced cuts) -> the ring encoders' chunk mode with B given (headers at block starts, the block's end as the end of the input) ->
// for the list codecs: every chunk from the default list, then the proof and kSplitPieces repair rounds (a chunk's list is wrong only if
// a chunk in front of it IN ITS BLOCK stored a symbol the default list does not lead with; a round makes one more chunk per block right) ->
// size scan over the chunks, placement, offset table + compressedLength fields.
static int compress_split(int codec, const uint8_t *dIn, uint64_t U, uint32_t B, uint32_t nBlocks, uint8_t *ws, const Workspace &w, uint64_t *offsets, uint8_t *payload, hipStream_t st)
{
  int S = 1, aligned = 0, listK = 0;
  const uint32_t longc = mono_cut_long(codec, &S, &aligned, &listK);
  const uint32_t ppb = split_pieces(B);
  const uint32_t G = B / ppb, pieces = nBlocks * ppb, maxChunks = pieces + nBlocks;
  uint64_t *cutPos = (uint64_t *)(ws + w.spCutPos), *cutSym = (uint64_t *)(ws + w.spCutSym), *idx = (uint64_t *)(ws + w.spIdx), *starts = (uint64_t *)(ws + w.spStarts);
  uint64_t *syms = (uint64_t *)(ws + w.spSyms), *slotOff = (uint64_t *)(ws + w.spSlotOff), *chunkOff = (uint64_t *)(ws + w.spChunkOff);
  uint64_t *guess = (uint64_t *)(ws + w.spGuess), *listOut = (uint64_t *)(ws + w.spListOut);
  uint32_t *flags = (uint32_t *)(ws + w.spFlags), *sizes = (uint32_t *)(ws + w.spSizes), *firstChunk = (uint32_t *)(ws + w.spFirst), *ctrl = (uint32_t *)(ws + w.spCtrl);
  Workspace sw{};
  sw.offL1 = w.spL1; sw.offL2 = w.spL2; sw.offL3 = w.spL3;
  static_assert(9u + kSplitPiecesMax - 1u <= 16u, "the list-verify rounds count into ctrl[9 + round]: below the verdict flag ctrl[24]");
  if (zero2_async(ctrl, 128, sizes, 4ull * (maxChunks + 1ull), st) != hipSuccess)   // (graph capturable: not hipMemsetAsync, see zero_async; 256 bytes are reserved)
    return HSRLE_ERR_DEVICE;
  const dim3 cgrid((pieces + 63u) / 64u);
  const bool single = codec == 4 || codec == 5 || codec == kSingleShort;
  uint32_t *const pickTable = (uint32_t *)(ws + w.spPick);             // (8 bit Single: a byte per block)
  if (single)
  {
    // every block's symbol first (k_single_pick through the codec's launcher): only its runs are cuts, and the chunk encoder needs it
    EncodeArgs pa{ dIn, U, B, nBlocks, nullptr, 0u, nullptr };
    MonoEncodeArgs pm{ nullptr, nullptr, nullptr, 0u };
    pm.pick = pickTable; pm.phase = 1u;
    pm.cutPos = cutPos; pm.cutSym = cutSym; pm.cutFlags = flags; pm.cutG = G; pm.cutLong = longc;   // (... and the blocks' cuts, by the same waves: the block is in LDS there)
    if (g_menc[codec](pa, pm, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
  }
#define HSRLE_CUTS(SS) \
  if (aligned) hipLaunchKernelGGL((k_mono_cutsS<SS, 1>), cgrid, dim3(64), 0, st, dIn, U, G, pieces, longc, cutPos, cutSym, flags, B); \
  else hipLaunchKernelGGL((k_mono_cutsS<SS, 0>), cgrid, dim3(64), 0, st, dIn, U, G, pieces, longc, cutPos, cutSym, flags, B)
  switch (S)
  {
  case 1: if (!single) hipLaunchKernelGGL(k_mono_cuts8, cgrid, dim3(64), 0, st, dIn, U, G, pieces, longc, cutPos, cutSym, flags, (const uint32_t *)nullptr, B); break;
  case 16: HSRLE_CUTS(16); break;
  case 2: HSRLE_CUTS(2); break;
  case 3: HSRLE_CUTS(3); break;
  case 4: HSRLE_CUTS(4); break;
  case 6: HSRLE_CUTS(6); break;
  default: HSRLE_CUTS(8); break;
  }
#undef HSRLE_CUTS
  if (scan_sizes(flags, pieces, idx, ws, sw, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (listK != 0)
    hipLaunchKernelGGL(k_mono_list_default, dim3((maxChunks + 255u) / 256u), dim3(256), 0, st, maxChunks, (uint32_t)listK, (uint32_t)S, guess);
  // list codecs of 2 .. 8 byte symbols settle their lists INSIDE the encode kernel: a wave takes the chunks of kSplitGroup whole blocks (k_encodeS_blocks)
  const uint32_t kSplitGroup = 64u / (ppb + 1u);                         // blocks per wave: x (ppb + 1) chunks at most <= a wave's 64 lanes
  const bool inKernelLists = listK != 0 && S > 1 && codec < kGreedyBase;   // (Greedy: one lane per chunk, lists between launches like the 8 bit codecs)
  hipLaunchKernelGGL(k_split_scatter, dim3((pieces + 255u) / 256u), dim3(256), 0, st, (const uint64_t *)cutPos, (const uint64_t *)cutSym, (const uint32_t *)flags, (const uint64_t *)idx, pieces,
                     ppb, nBlocks, U, B, starts, listK ? guess : syms, slotOff, firstChunk, ctrl, (uint32_t)listK | ((codec >= kGreedyBase && listK == 1) ? kSplitGuessCutSym : 0u), inKernelLists ? kSplitGroup : 0u);

  EncodeArgs ea{ dIn, U, B, inKernelLists ? ((nBlocks + kSplitGroup - 1u) / kSplitGroup) * 64u : maxChunks, ws + w.spSlots, 0u, sizes };
  ea.ringSel = ctrl;                                                     // chunk mode with B: ctrl[0] = the number of chunks, [1] = blocks per wave (0: chunks in a row), [2] = blocks
  MonoEncodeArgs ma{ starts, listK ? guess : syms, slotOff, 2u * (B / 64u) + 64u };
  ma.listOut = listK ? listOut : nullptr;
  ma.pick = pickTable;                                                 // (8 bit Single: the blocks' symbols)
  if (single) { ma.jobs = (uint64_t *)(ws + w.spJobs); ma.jobCount = ctrl + 12; ma.jobCap = (uint32_t)(w.spJobCap < 0xFFFFFFFFull ? w.spJobCap : 0xFFFFFFFFull); }   // (ctrl[12]: zeroed above)
  if (g_menc[codec](ea, ma, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (listK != 0 && !inKernelLists)
  {
    // (8 bit symbols: between launches.)  What every chunk does to a list is known from the first pass: the lists in front of all chunks at once, then the chunks whose list is
    // not the default one again (run-distributed rle64_3symlut_byte: every chunk).  Then the proof, with repair: a re-encoded chunk may store
    // other runs than it did from the default list and leave another list behind; after the pass above the first two chunks of every block are
    // final, every verify round makes one more so, and a block has at most kSplitPieces + 1 chunks.  The rounds tell the encoder which
    // counter says whether they have anything to do (an empty round is two launches that return at once).
    const uint32_t base = ma.steps;
    hipLaunchKernelGGL(k_split_list_guess, dim3((maxChunks + 255u) / 256u), dim3(256), 0, st, guess, (const uint64_t *)listOut, (const uint64_t *)starts, (const uint32_t *)ctrl, B, (uint32_t)listK, (uint32_t)S,
                       ctrl + 8);
    ma.steps = base | (1u << 16);
    if (g_menc[codec](ea, ma, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    for (uint32_t round = 0; round + 1u < ppb; round++)
    {
      hipLaunchKernelGGL(k_split_list_verify, dim3((maxChunks + 255u) / 256u), dim3(256), 0, st, guess, (const uint64_t *)listOut, (const uint64_t *)starts, (const uint32_t *)ctrl, B, (uint32_t)listK,
                         (uint32_t)S, ctrl + 9 + round);
      ma.steps = base | ((2u + round) << 16);
      if (g_menc[codec](ea, ma, st) != hipSuccess)                       // (chunks whose list was right are switched off: word 7 of their guess)
        return HSRLE_ERR_DEVICE;
    }
  }
  const bool laneChunks = single || S == 16 || codec >= kGreedyBase;     // per-lane chunk encoders: a chunk that missed its boundary run says so with size 0
  if (laneChunks)
    hipLaunchKernelGGL(k_split_check, dim3((nBlocks + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)firstChunk, (const uint64_t *)starts, (const uint32_t *)sizes, nBlocks, ctrl + 24);   // (ctrl[24]: zeroed above; outside the rounds' counters ctrl[9 ..] and the Single encoders' ctrl[12] -- ADVICE r5)
  if (scan_sizes(sizes, maxChunks, chunkOff, ws, sw, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  launch_compact_var(G <= 2048u, (const uint8_t *)(ws + w.spSlots), (const uint64_t *)slotOff, (const uint64_t *)chunkOff, payload, maxChunks, st);
  // (... and the container's header and tail pad: the caller does not launch k_finish_container behind a split encode)
  uint8_t *const container = (uint8_t *)offsets - HSRLE_CONTAINER_HEADER_SIZE;
  const uint32_t codecId = (uint32_t)codec;
  auto finish = [=] __device__(uint64_t payloadSize) { finish_container(container, codecId, U, B, nBlocks, payloadSize); };
  hipLaunchKernelGGL((k_split_finish<decltype(finish)>), dim3((nBlocks + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)firstChunk, (const uint64_t *)chunkOff, nBlocks, offsets, payload, finish);
  if (laneChunks)
    hipLaunchKernelGGL(k_split_verdict, dim3(1), dim3(64), 0, st, (const uint32_t *)(ctrl + 24), container);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

constexpr uint32_t kMonoListRounds = 64u;     // repair rounds of the guessed move-to-front lists before the caller falls back to one lane
__global__ void k_copy_word(const uint32_t *__restrict__ from, uint32_t *__restrict__ to) { *to = *from; }

static thread_local uint32_t g_monoEncLast[4] = { 0, 0, 0, 0 };   // this thread's last list-codec encode: extra rounds, chunks encoded again in rounds 1, 2, chunks the proof rejected (hsrle_mono_encode_stats)

struct MonoEncPlan
{
  uint32_t G, pieces;
  bool windowed;                                       // rle8_multi / rle8_packed_multi: chunks of any length by the windowed position-parallel encoder (hsrle_encode8pw.hip.h)
  uint64_t offCutPos, offCutSym, offFlags, offIdx, offStarts, offSyms, offSlotOff, offSizes, offOffsets, offL1, offL2, offL3, offCtrl, offSlots, total;
  uint64_t offGuess, offListOut, offRoll1, offRoll2;   // codecs with a move-to-front list: 8 words per chunk / per 64 / per 4096 chunks
  uint64_t offPick;                                    // 8 bit Single: the symbol pick's sums (k_single_pick_mono)
  uint64_t offJobs; uint32_t jobCap;                   // 8 bit Single: literal stretches noted by the chunk encoders for k_copy_jobs
};

static bool mono_windowed(int codec) { return (codec == 0 || codec == 1) && g_ppw[codec] != nullptr && kPpwMinBlocks != 0xFFFFFFFFu && knob_u32("HSRLE_PP", 0u) != 2u; }

static MonoEncPlan plan_mono_encode(uint32_t U, int codec, bool lists = true)
{
  MonoEncPlan m;
  init_tables();
  m.windowed = mono_windowed(codec);
  // ~131 072 pieces (= lanes) keep the device busy: 1 GiB: 8 KiB pieces 780 GiB/s, 4 KiB 720; 256 MiB: 2 KiB 523, 4 KiB 321; 88 MB: 1 KiB 286, 2 KiB 224
  uint32_t G = 1024u;
  while (G < 8192u && ((uint64_t)U + G - 1u) / G > 131072ull) G *= 2u;
  // windowed: a chunk is a WAVE's work and every chunk ends with a partial window, so the pieces are several windows long as soon as that leaves
  // ~6 000 of them (1 GiB: 4 / 8 / 16 / 32 / 64 / 128 KiB pieces 1.72 / 1.23 / 1.09 / 1.00 / 0.95 / 0.97 ms; 88 MB: 0.25 / 0.22 / 0.20 / 0.21 / 0.22 / 0.26)
  if (m.windowed) { G = 8192u; while (G < 65536u && (uint64_t)U / G > 6000ull) G *= 2u; }
  if (g_monoTune[1] >= 32u && g_monoTune[1] <= (1u << 24)) G = g_monoTune[1];
  m.G = G;
  m.pieces = (uint32_t)(((uint64_t)U + G - 1u) / G);
  const uint64_t n = m.pieces;
  const uint64_t t1 = (n + 2 + kScanTile - 1) / kScanTile, t2 = (t1 + kScanTile - 1) / kScanTile, t3 = (t2 + kScanTile - 1) / kScanTile;
  uint64_t at = 0;
  m.offCutPos = at; at += align_up(8ull * n, 256);
  m.offCutSym = at; at += align_up(8ull * n, 256);
  m.offFlags = at; at += align_up(4ull * n, 256);
  m.offIdx = at; at += align_up(8ull * (n + 1), 256);
  m.offStarts = at; at += align_up(8ull * (n + 2), 256);
  m.offSyms = at; at += align_up(8ull * (n + 1), 256);
  m.offSlotOff = at; at += align_up(8ull * (n + 1), 256);
  m.offSizes = at; at += align_up(4ull * (n + 1), 256);
  m.offOffsets = at; at += align_up(8ull * (n + 2), 256);
  m.offL1 = at; at += align_up(8ull * (t1 + 1), 256);
  m.offL2 = at; at += align_up(8ull * (t2 + 1), 256);
  m.offL3 = at; at += align_up(8ull * (t3 + 1), 256);
  m.offCtrl = at; at += 256;
  m.offPick = at; at += 4096;
  m.jobCap = U / 1024u + 16u;                                            // (every stretch of >= kCopyJobMin bytes there can be)
  m.offJobs = at; at += align_up(24ull * m.jobCap, 256);
  m.offGuess = m.offListOut = m.offRoll1 = m.offRoll2 = at;
  if (lists)
  {
    m.offGuess = at; at += align_up(64ull * (n + 1), 256);
    m.offListOut = at; at += align_up(64ull * (n + 1), 256);
    m.offRoll1 = at; at += align_up(64ull * ((n + 1) / 64 + 1), 256);
    m.offRoll2 = at; at += align_up(64ull * ((n + 1) / 4096 + 1), 256);
  }
  // (windowed: no staging slots -- the window states and records live there: 32 + 1 024 bytes per window, at most U / 4 096 + chunks windows)
  const uint64_t windowsMax = ((uint64_t)U >> 12) + n + 1ull;
  const uint64_t slotBytes = (uint64_t)U + ((uint64_t)U >> 7) + 256ull * (n + 2) + 4096ull, windowBytes = align_up(4ull * kPpwStateWords * windowsMax, 256) + 4ull * kPpwStride * windowsMax + 512ull;
  m.offSlots = at; at += align_up(m.windowed && windowBytes > slotBytes ? windowBytes : slotBytes, 256);
  m.total = at;
  return m;
}

// dOut: capacity >= rle_compress_bounds(U).  Synchronises the stream twice (chunk count, stream size) -- the windowed encoders once, at the end.
static int mono_encode_dev(int codec, const uint8_t *dIn, uint32_t U, uint8_t *dOut, uint8_t *ws, const MonoEncPlan &m, uint32_t *pSize, uint32_t *pChunks, hipStream_t st)
{
  init_tables();
  int S = 1, aligned = 0, listK = 0;
  const uint32_t longc = mono_cut_long(codec, &S, &aligned, &listK);
  if (!g_menc[codec] || longc == 0u)
    return HSRLE_ERR_UNSUPPORTED;
  const bool single = codec == 4 || codec == 5 || codec == kSingleShort;
  const uint32_t hs = codec_header_size(codec) + (single ? 1u : 0u);   // (Single: the symbol byte follows the header)
  uint64_t *cutPos = (uint64_t *)(ws + m.offCutPos), *idx = (uint64_t *)(ws + m.offIdx), *starts = (uint64_t *)(ws + m.offStarts), *slotOff = (uint64_t *)(ws + m.offSlotOff);
  uint64_t *offsets = (uint64_t *)(ws + m.offOffsets);
  uint64_t *cutSym = (uint64_t *)(ws + m.offCutSym), *syms = (uint64_t *)(ws + m.offSyms);
  uint32_t *flags = (uint32_t *)(ws + m.offFlags), *sizes = (uint32_t *)(ws + m.offSizes), *ctrl = (uint32_t *)(ws + m.offCtrl);
  Workspace w{};
  w.offL1 = m.offL1; w.offL2 = m.offL2; w.offL3 = m.offL3;

  if (zero_async(ctrl, 64, st) != hipSuccess)                            // (a kernel, not hipMemsetAsync: the windowed flow below can be captured in a HIP graph, see zero_async)
    return HSRLE_ERR_DEVICE;
  if (single)
  {
    // the stream's ONE symbol first (rle8_extreme_cpu.c:53-153 over the whole input): sums per piece, then the estimator's end game and the argmax -> ctrl[8]
    uint32_t *table = (uint32_t *)(ws + m.offPick);
    const uint32_t pp = (U + kPickPiece - 1u) / kPickPiece;
    if (hipMemsetAsync(table, 0, 2064, st) != hipSuccess || hipMemsetAsync(table + 514, 0xFF, 8, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    hipLaunchKernelGGL(k_single_pick_mono, dim3(pp < 4096u ? pp : 4096u), dim3(64), 0, st, dIn, U, pp, table);
    hipLaunchKernelGGL(k_single_pick_final, dim3(1), dim3(64), 0, st, dIn, U, table, ctrl + 8);
  }
  const dim3 cgrid((m.pieces + 63u) / 64u);
#define HSRLE_CUTS(SS) \
  if (aligned) hipLaunchKernelGGL((k_mono_cutsS<SS, 1>), cgrid, dim3(64), 0, st, dIn, (uint64_t)U, m.G, m.pieces, longc, cutPos, cutSym, flags); \
  else hipLaunchKernelGGL((k_mono_cutsS<SS, 0>), cgrid, dim3(64), 0, st, dIn, (uint64_t)U, m.G, m.pieces, longc, cutPos, cutSym, flags)
  switch (S)
  {
  case 1: hipLaunchKernelGGL(k_mono_cuts8, cgrid, dim3(64), 0, st, dIn, (uint64_t)U, m.G, m.pieces, longc, cutPos, cutSym, flags, single ? (const uint32_t *)(ctrl + 8) : (const uint32_t *)nullptr); break;
  case 2: HSRLE_CUTS(2); break;
  case 3: HSRLE_CUTS(3); break;
  case 4: HSRLE_CUTS(4); break;
  case 6: HSRLE_CUTS(6); break;
  case 16: HSRLE_CUTS(16); break;
  default: HSRLE_CUTS(8); break;
  }
#undef HSRLE_CUTS
  if (scan_sizes(flags, m.pieces, idx, ws, w, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_mono_scatter, dim3((m.pieces + 255u) / 256u), dim3(256), 0, st, (const uint64_t *)cutPos, (const uint64_t *)cutSym, (const uint32_t *)flags, (const uint64_t *)idx,
                     m.pieces, (uint64_t)U, starts, syms, slotOff, ctrl);
  // rle8_multi / rle8_packed_multi: the windowed position-parallel encoder takes chunks of any length (hsrle_encode8pw.hip.h) -- no step bound, no staging slots
  const uint64_t windowsMax = ((uint64_t)U >> 12) + m.pieces + 1ull;
  const bool windowed = m.windowed && mono_windowed(codec);
  if (windowed)
  {
    // every piece may be a chunk: a wave per possible chunk (those behind the last one write a zero size), a wave per possible window -- nothing is read back
    // before the end
    PpwArgs pa{};
    pa.in = dIn; pa.U = U; pa.B = 0u; pa.nUnits = m.pieces + 1u; pa.starts = starts; pa.syms = syms; pa.count = ctrl; pa.sizes = sizes; pa.offsets = offsets; pa.payload = dOut + hs;
    pa.nWindows = (uint32_t)windowsMax;
    pa.states = (uint32_t *)(ws + m.offSlots);
    pa.recs = (uint32_t *)(ws + m.offSlots + align_up(4ull * kPpwStateWords * windowsMax, 256));
    if (g_ppw[codec](pa, 0, st) != hipSuccess || scan_sizes(sizes, pa.nUnits, offsets, ws, w, st) != hipSuccess || g_ppw[codec](pa, 1, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    hipLaunchKernelGGL(k_mono_finish, dim3(1), dim3(64), 0, st, dOut, U, hs, (const uint64_t *)offsets, (const uint32_t *)ctrl, ctrl, 0u);
    if (pSize == nullptr)
    {
      // hsrle_compress_mono_dev_async: the size stays on the device (the stream's own header holds it; pChunks, if given, is a DEVICE word that receives it too)
      if (pChunks) hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, st, (const uint32_t *)(ctrl + 2), pChunks);
      return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
    }
    uint32_t back[4] = { 0, 0, 0, 0 };                                    // chunks, -, stream size, error
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(back, ctrl, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    if (back[0] == 0u || back[0] > m.pieces + 1u || back[3] != 0u || back[2] == 0u)
      return HSRLE_ERR_DEVICE;
    if (pChunks) *pChunks = back[0];
    *pSize = back[2];
    return HSRLE_OK;
  }
  hipLaunchKernelGGL(k_mono_longest, dim3((m.pieces + 1u + 255u) / 256u), dim3(256), 0, st, (const uint64_t *)starts, ctrl);
  uint32_t head[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  if (hipMemcpyAsync(head, ctrl, 40, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (single && head[9] != 0u)
    return HSRLE_ERR_UNSUPPORTED;                                        // (a run of more than 16 MiB: the pick gave up -- callers fall back to one lane)
  const uint32_t chunks = head[0], longest = head[1];
  if (chunks == 0u || chunks > m.pieces + 1u)
    return HSRLE_ERR_DEVICE;
  if (pChunks) *pChunks = chunks;

  EncodeArgs ea{ dIn, (uint64_t)U, 0u, chunks, ws + m.offSlots, 0u, sizes };
  MonoEncodeArgs ma{ starts, syms, slotOff, 2u * (longest / 64u) + 64u };
  ma.pick = ctrl + 8;
  if (single) { ma.jobs = (uint64_t *)(ws + m.offJobs); ma.jobCount = ctrl + 12; ma.jobCap = m.jobCap; }   // (ctrl[12] was zeroed with the rest)
  if (listK == 0)
  {
    if (g_menc[codec](ea, ma, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
  }
  else
  {
    // move-to-front list: dry pass -> guessed lists -> encode -> verify, repeat for the chunks whose guess was wrong (hsrle_mono_encode.hip.h)
    uint64_t *guess = (uint64_t *)(ws + m.offGuess), *listOut = (uint64_t *)(ws + m.offListOut), *roll1 = (uint64_t *)(ws + m.offRoll1), *roll2 = (uint64_t *)(ws + m.offRoll2);
    const uint32_t n1 = (chunks + 63u) / 64u, n2 = (n1 + 63u) / 64u;
    ma.syms = guess; ma.listOut = listOut;
    hipLaunchKernelGGL(k_mono_list_default, dim3((chunks + 255u) / 256u), dim3(256), 0, st, chunks, (uint32_t)listK, (uint32_t)S, guess);
    ma.dry = 1u;
    if (g_menc[codec](ea, ma, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    ma.dry = 0u;
    uint32_t rounds = 0;
    g_monoEncLast[0] = g_monoEncLast[1] = g_monoEncLast[2] = g_monoEncLast[3] = 0u;
    for (;; rounds++)
    {
      // lists from what the chunks did in the last pass; chunks whose list changed (first time: all) are encoded from it
      if (rounds > kMonoListRounds)
        return HSRLE_ERR_UNSUPPORTED;                                    // (callers fall back to one lane)
      if (hipMemsetAsync(ctrl + 4, 0, 4, st) != hipSuccess)
        return HSRLE_ERR_DEVICE;
      hipLaunchKernelGGL(k_mono_list_tiles, dim3((n1 + 63u) / 64u), dim3(64), 0, st, (const uint64_t *)listOut, chunks, (uint32_t)listK, roll1);
      hipLaunchKernelGGL(k_mono_list_tiles, dim3((n2 + 63u) / 64u), dim3(64), 0, st, (const uint64_t *)roll1, n1, (uint32_t)listK, roll2);
      hipLaunchKernelGGL(k_mono_list_guess, dim3((chunks + 63u) / 64u), dim3(64), 0, st, (const uint64_t *)listOut, (const uint64_t *)roll1, (const uint64_t *)roll2, chunks,
                         (uint32_t)listK, (uint32_t)S, guess, rounds == 0u ? 1u : 0u, ctrl + 4);
      if (rounds > 0u)
      {
        uint32_t todo = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&todo, ctrl + 4, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
          return HSRLE_ERR_DEVICE;
        g_monoEncLast[0] = rounds - 1u;
        if (rounds <= 2u) g_monoEncLast[rounds] = todo;
        if (todo == 0u)
          break;
      }
      if (g_menc[codec](ea, ma, st) != hipSuccess)
        return HSRLE_ERR_DEVICE;
    }
    for (;; rounds++)
    {
      // the proof (and, should the fixed point above not be one, the repair)
      if (hipMemsetAsync(ctrl + 4, 0, 4, st) != hipSuccess)
        return HSRLE_ERR_DEVICE;
      hipLaunchKernelGGL(k_mono_list_verify, dim3((chunks + 255u) / 256u), dim3(256), 0, st, guess, (const uint64_t *)listOut, chunks, (uint32_t)listK, ctrl + 4);
      uint32_t bad = 0;
      if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad, ctrl + 4, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return HSRLE_ERR_DEVICE;
      g_monoEncLast[3] += bad;
      if (bad == 0u)
        break;
      if (rounds > kMonoListRounds)
        return HSRLE_ERR_UNSUPPORTED;
      if (g_menc[codec](ea, ma, st) != hipSuccess)
        return HSRLE_ERR_DEVICE;
    }
  }
  if (scan_sizes(sizes, chunks, offsets, ws, w, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  launch_compact_var(false, (const uint8_t *)(ws + m.offSlots), (const uint64_t *)slotOff, (const uint64_t *)offsets, dOut + hs, chunks, st);
  if (single || codec >= kGreedyBase)
    hipLaunchKernelGGL(k_mono_zero_sizes, dim3((chunks + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)sizes, chunks, ctrl + 5);
  hipLaunchKernelGGL(k_mono_finish, dim3(1), dim3(64), 0, st, dOut, U, hs, (const uint64_t *)offsets, (const uint32_t *)ctrl, ctrl, codec == kSingleShort ? 1u : 0u);
  uint32_t tail[4] = { 0, 0, 0, 0 };
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(tail, ctrl + 2, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (tail[3] != 0u)
    return HSRLE_ERR_UNSUPPORTED;                                        // (a Single chunk that did not end on its boundary run: one lane, by the caller)
  if (tail[1] != 0u || tail[0] == 0u)
    return HSRLE_ERR_DEVICE;
  *pSize = tail[0];
  return HSRLE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// drop-in (host pointer, monolithic stream) path.  Decode: index + block kernel (above).  Encode: one block spanning the whole input.

static uint32_t mono_compress(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  // argument checks of the reference (rle8_extreme_cpu.h:88-89, rleX_extreme_cpu.h:49-50, rleX_Xsl.h:271-272)
  if (pIn == nullptr || inSize == 0 || pOut == nullptr || outSize < bounds32(inSize))
    return 0;
  if (inSize > (1u << 30)) // A.5 q8: sizes above 1 GiB cannot be bounded; treated as unsupported
    return 0;
  if (codec < 0 || codec >= kCodecCount || !device_ok())
    return 0;

  init_tables();
  if (!g_enc[codec])
    return 0;

  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  const uint32_t stride = (bounds32(inSize) + 15u) & ~15u;
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)inSize + 64) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)stride + 64))
    return 0;

  // many lanes where the codec allows it (cuts behind long runs, hsrle_mono_encode.hip.h); else -- and for inputs of one piece -- one lane
  if (mono_cut_long(codec) != 0u && g_menc[codec])
  {
    const MonoEncPlan m = plan_mono_encode(inSize, codec);
    if (m.pieces >= 2u)
    {
      if (!grow(&D.monoWs, &D.monoWsSize, m.total))
        return 0;
      if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess)
        return 0;
      uint32_t size = 0;
      const int rc = mono_encode_dev(codec, (const uint8_t *)D.monoIn, inSize, (uint8_t *)D.monoOut, (uint8_t *)D.monoWs, m, &size, nullptr, nullptr);
      if (rc == HSRLE_OK)
      {
        if (size == 0 || size > outSize || hipMemcpy(pOut, D.monoOut, size, hipMemcpyDeviceToHost) != hipSuccess)
          return 0;
        return size;
      }
      if (rc != HSRLE_ERR_UNSUPPORTED)                                   // (UNSUPPORTED: the list guesses did not settle -- one lane, below)
        return 0;
    }
  }
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;

  if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess)
    return 0;

  EncodeArgs ea{ (const uint8_t *)D.monoIn, inSize, inSize, 1u, (uint8_t *)D.monoOut, stride, (uint32_t *)D.monoAux };
  if (g_enc[codec](ea, nullptr) != hipSuccess)
    return 0;

  uint32_t size = 0;
  if (hipMemcpy(&size, D.monoAux, 4, hipMemcpyDeviceToHost) != hipSuccess || size == 0 || size > outSize)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, size, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return size;
}

static uint32_t mono_decompress(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  // argument + header checks of the reference (rle8_extreme_cpu.h:704-712, rleX_extreme_cpu.h:84-91, rleX_Xsl.h:1850-1858)
  if (pIn == nullptr || pOut == nullptr || inSize == 0 || outSize == 0)
    return 0;
  if (codec < 0 || codec >= kCodecCount || inSize < codec_header_size(codec))
    return 0;
  uint8_t h16[16] = { 0 };
  memcpy(h16, pIn, inSize < 16u ? inSize : 16u);
  MonoHeader mh;
  if (!mono_header(codec, h16, inSize, outSize, &mh) || !device_ok())
    return 0;

  const MonoPlan m = plan_mono(mh.codec, mh.U, mh.C, mh.p0);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)mh.C + 256) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)mh.U + 64) ||
      !grow(&D.monoWs, &D.monoWsSize, m.total))
    return 0;
  if (hipMemcpy(D.monoIn, pIn, mh.C, hipMemcpyHostToDevice) != hipSuccess || hipMemset((uint8_t *)D.monoIn + mh.C, 0, 128) != hipSuccess)
    return 0;
  if (mono_decode_dev(mh, (const uint8_t *)D.monoIn, (uint8_t *)D.monoOut, (uint8_t *)D.monoWs, m, nullptr, nullptr) != HSRLE_OK)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, mh.U, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return mh.U;
}

// ---- rle8m: the reference's GPU decode path, rle8m_opencl_decompress (src/rle8_ocl.c:265-413) ----

constexpr uint32_t kRle8mWaveBelow = 131072u;   // measured on 1 GiB: 65 536 sections 570 (wave) against 321 GiB/s (lane), 262 144 sections 344-498 against 612-680

static int rle8m_decode_async(const void *dStream, uint64_t streamSize, uint32_t uncompressedSize, uint32_t sections, void *dOut, uint64_t outCapacity,
                              uint32_t *dStatus, hipStream_t st)
{
  // the caller has checked device_ok()
  if (!dStream || !dOut || streamSize < 12 || sections == 0 || uncompressedSize == 0 || outCapacity < uncompressedSize)
    return HSRLE_ERR_ARGUMENT;
  if (dStatus && zero_async(dStatus, 4, st) != hipSuccess)               // (a kernel, not hipMemsetAsync: graph capturable, see zero_async)
    return HSRLE_ERR_DEVICE;
  // few, large sections: one wave per section (one lane per section needs ~1e5 sections to fill the GPU)
  static const int forced = (int)knob_u32("HSRLE_RLE8M_DECODE", 0);   // 1 = lane, 2 = wave kernel (A/B runs)
  // ... and sections of 4 KiB and more decode faster that way whatever their number (1 GiB, 4 KiB sections: 1034 against 717 GiB/s on
  // video-shaped bytes, 618 against 629 on bytes that do not compress; 1 KiB sections: 841 / 501 against 737 / 617)
  const bool wave = forced ? forced == 2 : (sections < kRle8mWaveBelow || uncompressedSize / sections >= 4096u);
  if (wave)
    hipLaunchKernelGGL(k_rle8m_decode_wave, dim3(sections), dim3(64), 0, st, (const uint8_t *)dStream, streamSize, (uint8_t *)dOut, dStatus, uncompressedSize, sections);
  else
    hipLaunchKernelGGL(k_rle8m_decode, dim3((sections + 63u) / 64u), dim3(64), 0, st, (const uint8_t *)dStream, streamSize, (uint8_t *)dOut, dStatus, uncompressedSize, sections);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

// rle8m encode (device resident): workspace = [Rle8mTables][offsets u64 x (sections + 1)] + the slots / sizes / scan levels of plan_workspace
// (launch_le_stats below: three words per 4 KiB of input)
static inline uint64_t le_stats_ws_bytes(uint64_t n) { return 3ull * align_up(4ull * ((n + 4095ull) / 4096ull + 1ull), 256); }
struct Rle8mPlan
{
  Workspace w;
  uint64_t offTables, offOffsets, offStats, total;
  uint32_t slotStride;
};

static Rle8mPlan plan_rle8m(uint32_t n, uint32_t sections)
{
  Rle8mPlan p;
  const uint32_t ss = n / sections, lastLen = n - ss * (sections - 1u);
  p.slotStride = (uint32_t)align_up(2ull * (uint64_t)(lastLen > ss ? lastLen : ss) + 16ull, 16);   // a section grows to at most twice its size
  Workspace &w = p.w;
  w.nBlocks = sections; w.chunk = sections; w.nChunks = 1;
  w.t1 = (w.nBlocks + kScanTile - 1) / kScanTile;
  w.t2 = (w.t1 + kScanTile - 1) / kScanTile;
  w.t3 = (w.t2 + kScanTile - 1) / kScanTile;
  uint64_t at = 0;
  p.offTables = at; at += align_up(sizeof(Rle8mTables), 256);
  p.offOffsets = at; at += align_up(((uint64_t)sections + 1ull) * 8ull, 256);
  p.offStats = at; at += le_stats_ws_bytes(n);
  w.offSlots = at; at += align_up((uint64_t)sections * p.slotStride, 256);
  w.offSizes = at; at += align_up((uint64_t)sections * 4ull, 256);
  w.offL1 = at; at += align_up((w.t1 + 1) * 8ull, 256);
  w.offL2 = at; at += align_up((w.t2 + 1) * 8ull, 256);
  w.offL3 = at; at += align_up((w.t3 + 1) * 8ull, 256);
  w.total = at;
  p.total = at;
  return p;
}

// the statistics of the low-entropy encoders (rle8_low_entropy_cpu.c:264-296) over the whole input: k_rle8m_stats_wave piece by piece, the pieces' run starts
// scanned, the token boundaries of runs that cross pieces added (hsrle_rle8m.hip.h, round 6: no lane follows a run through global memory any more).
// wsStats: le_stats_ws_bytes(n) bytes; *pRunStart4 (optional): the start of the run that covers the first byte of every 4 KiB piece (k_le_cuts)
static hipError_t launch_le_stats(const uint8_t *dIn, uint32_t n, Rle8mTables *t, uint32_t maxLen, uint8_t *wsStats, hipStream_t st, uint32_t maxWaves = 32768u, const uint32_t **pRunStart4 = nullptr)
{
  const uint32_t p4 = (uint32_t)(((uint64_t)n + 4095u) / 4096u);
  const uint64_t stride = align_up(4ull * ((uint64_t)p4 + 1ull), 256);
  uint32_t *lastB4 = (uint32_t *)wsStats, *firstB4 = (uint32_t *)(wsStats + stride), *runStart4 = (uint32_t *)(wsStats + 2ull * stride);
  uint32_t grid = p4 < maxWaves ? p4 : maxWaves;
  if ((uint64_t)grid * kRle8mStatsPieces < p4) grid = (p4 + kRle8mStatsPieces - 1u) / kRle8mStatsPieces;   // (no wave gets more pieces than its packed counters hold)
  hipLaunchKernelGGL(k_rle8m_stats_wave, dim3(grid), dim3(64), 0, st, dIn, n, t, maxLen, lastB4, firstB4);
  hipLaunchKernelGGL(k_le_scan_last, dim3(1), dim3(1024), 0, st, (const uint32_t *)lastB4, p4, runStart4);
  hipLaunchKernelGGL(k_le_stats_fixup, dim3((p4 + 255u) / 256u), dim3(256), 0, st, dIn, n, p4, (const uint32_t *)lastB4, (const uint32_t *)firstB4, (const uint32_t *)runStart4, t, maxLen);
  if (pRunStart4) *pRunStart4 = runStart4;
  return hipGetLastError();
}

static uint32_t rle8m_bounds(uint32_t sections, uint32_t n) { return n + (256 / 8) + 1 + 256 + 4u * (2u + sections - 1u + 1u); }

// the caller has checked device_ok()
// maxLen / onlyMax: the four unsectioned encoders share these kernels (rle8_low_entropy[_short]_compress[_only_max_frequency]: runs are cut
// every 255 or 32 bytes, and either every symbol whose runs average >= 2 carries repeat codes or only the one that saves the most)
static int rle8m_encode_async(const void *dIn, uint32_t n, uint32_t sections, void *dOut, uint64_t outCapacity, void *dWs, uint64_t wsSize, uint32_t *dStatus, hipStream_t st,
                              uint32_t maxLen = 255u, uint32_t onlyMax = 0u)
{
  if (!dIn || !dOut || !dWs || n == 0 || sections == 0)
    return HSRLE_ERR_ARGUMENT;
  if (outCapacity < rle8m_bounds(sections, n))
    return HSRLE_ERR_CAPACITY;
  const Rle8mPlan p = plan_rle8m(n, sections);
  if (wsSize < p.total)
    return HSRLE_ERR_CAPACITY;
  uint8_t *ws = (uint8_t *)dWs;
  Rle8mTables *t = (Rle8mTables *)(ws + p.offTables);
  uint64_t *offsets = (uint64_t *)(ws + p.offOffsets);
  uint32_t *sizes = (uint32_t *)(ws + p.w.offSizes);
  if (zero_async(t, sizeof(Rle8mTables), st) != hipSuccess || (dStatus && zero_async(dStatus, 4, st) != hipSuccess))
    return HSRLE_ERR_DEVICE;
  const uint32_t grid = (sections + 63u) / 64u;
  // the statistics are over the whole input: one lane per 4 KiB piece, whatever the section count
  const uint32_t pieces = (n / 4096u > sections) ? n / 4096u : sections;
  static const uint32_t g_rle8mStatsWaves = knob_u32("HSRLE_RLE8M_STATS_WAVES", 32768u);   // (1 GiB run-distributed / video-shaped: 1 024 waves 7.5 / 6.2 ms per encode, 8 192: 3.96 / 4.17, 32 768: 3.75 / 3.98; the byte-walking kernel: 4.03 / 4.60)
  static const int statsV1 = (int)knob_u32("HSRLE_RLE8M_STATS", 0);   // 1 = the byte-walking kernel (A/B runs)
  if (statsV1 == 1)
    hipLaunchKernelGGL(k_rle8m_stats, dim3((pieces + 63u) / 64u), dim3(64), 0, st, (const uint8_t *)dIn, n, pieces, t, maxLen);
  else
  {
    if (launch_le_stats((const uint8_t *)dIn, n, t, maxLen, ws + p.offStats, st, g_rle8mStatsWaves) != hipSuccess)
      return HSRLE_ERR_DEVICE;
  }
  hipLaunchKernelGGL(k_rle8m_info, dim3(1), dim3(256), 0, st, t, sections, (uint8_t *)dOut, onlyMax);
  static const int forced = (int)knob_u32("HSRLE_RLE8M_ENCODE", 0);   // 1 = lane, 2 = wave kernel (A/B runs)
  if (forced ? forced == 2 : sections < kRle8mWaveBelow)
    hipLaunchKernelGGL(k_rle8m_encode_wave, dim3(sections), dim3(64), 0, st, (const uint8_t *)dIn, n, sections, (const Rle8mTables *)t, ws + p.w.offSlots, p.slotStride, sizes, maxLen);
  else
    hipLaunchKernelGGL(k_rle8m_encode, dim3(grid), dim3(64), 0, st, (const uint8_t *)dIn, n, sections, (const Rle8mTables *)t, ws + p.w.offSlots, p.slotStride, sizes, maxLen);
  if (scan_sizes(sizes, sections, offsets, ws, p.w, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_rle8m_place, dim3((sections + 3u) / 4u), dim3(256), 0, st, (const uint8_t *)(ws + p.w.offSlots), p.slotStride, (const uint64_t *)offsets, (const Rle8mTables *)t,
                     (uint8_t *)dOut, outCapacity, n, sections, dStatus);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

static uint32_t rle8m_mono_compress(uint32_t sections, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  // argument checks of the reference (rle8_low_entropy_cpu.c:133-134)
  if (pIn == nullptr || inSize == 0 || pOut == nullptr || sections == 0 || outSize < rle8m_bounds(sections, inSize) || !device_ok())
    return 0;
  const Rle8mPlan p = plan_rle8m(inSize, sections);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)inSize + 64) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)outSize + 64) || !grow(&D.ws, &D.wsSize, p.total))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess)
    return 0;
  if (rle8m_encode_async(D.monoIn, inSize, sections, D.monoOut, outSize, D.ws, D.wsSize, dStatus, nullptr) != HSRLE_OK)
    return 0;
  uint32_t status = 1, size = 0;
  if (hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost) != hipSuccess || status != 0)
    return 0;
  if (hipMemcpy(&size, D.monoOut, 4, hipMemcpyDeviceToHost) != hipSuccess || size == 0 || size > outSize)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, size, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return size;
}

static uint32_t rle8m_mono_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  // argument + header checks of the reference (rle8_ocl.c:267-283, rle8_low_entropy_cpu.c:195-211)
  if (pIn == nullptr || pOut == nullptr || inSize < 12 || outSize == 0)
    return 0;
  uint32_t expIn, expOut, sections;
  memcpy(&expIn, pIn, 4); memcpy(&expOut, pIn + 4, 4); memcpy(&sections, pIn + 8, 4);
  if (expOut > outSize || expIn > inSize || sections == 0 || expOut == 0 || !device_ok())
    return 0;

  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)expIn + 64) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)expOut + 64))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  if (hipMemcpy(D.monoIn, pIn, expIn, hipMemcpyHostToDevice) != hipSuccess)
    return 0;
  if (rle8m_decode_async(D.monoIn, expIn, expOut, sections, D.monoOut, expOut, dStatus, nullptr) != HSRLE_OK)
    return 0;
  uint32_t status = 1;
  if (hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost) != hipSuccess || status != 0)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, expOut, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return expOut;
}

} // namespace hsrle

// ====================================================================================================================
using namespace hsrle;

extern "C" {

int hsrle_codec_from_name(const char *name)
{
  if (!name) return -1;
  for (int k = 0; k < kCodecCount; k++)
    if (strcmp(name, kCodecNames[k]) == 0) return k;
  return -1;
}

const char *hsrle_codec_name(int codec) { return (codec >= 0 && codec < kCodecCount) ? kCodecNames[codec] : nullptr; }

const char *hsrle_status_string(int s)
{
  switch (s)
  {
  case HSRLE_OK: return "ok";
  case HSRLE_ERR_ARGUMENT: return "invalid argument";
  case HSRLE_ERR_CAPACITY: return "buffer too small";
  case HSRLE_ERR_FORMAT: return "malformed container or stream";
  case HSRLE_ERR_DEVICE: return "no usable HIP device / HIP runtime error";
  case HSRLE_ERR_UNSUPPORTED: return "codec not available in this build";
  }
  return "unknown";
}

const char *hsrle_version(void) { return "hsrle-hip 0.1 (gfx950)"; }

uint32_t hsrle_suggest_block_size(uint64_t inSize)
{
  if (inSize >= 65536ull * 4096ull) return 4096u;
  if (inSize >= 65536ull * 2048ull) return 2048u;
  return 1024u;
}

int hsrle_kernel_waves_per_cu(int codec, int decode)
{
  if (codec < 0 || codec >= kCodecCount)
    return 0;
  init_tables();
  int n = 0;
  if (decode) { DecodeArgs a{}; a.residentWorkgroups = &n; if (g_dec[codec](a, nullptr) != hipSuccess) return 0; }
  else { EncodeArgs a{}; a.residentWorkgroups = &n; if (g_enc[codec](a, nullptr) != hipSuccess) return 0; }
  return n;
}

int hsrle_scratch_retention(uint64_t bytes)
{
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  g_scratchRetention.store(bytes);
  hipMemPool_t pool = device_pool();
  if (pool == nullptr) return HSRLE_ERR_UNSUPPORTED;
  uint64_t keep = bytes;
  return hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

int hsrle_trim(void)
{
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  hipMemPool_t pool = device_pool();
  if (pool == nullptr) return HSRLE_OK;
  return hipMemPoolTrimTo(pool, 0) == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

int hsrle_experiments_enabled(void) { return kExperiments ? 1 : 0; }

int hsrle_encode_path(int codec, uint64_t uncompressedSize, uint32_t blockSize)
{
  if (codec < 0 || codec >= kCodecCount || uncompressedSize == 0 || !valid_block_size(blockSize)) return -1;
  const Workspace w = plan_workspace(uncompressedSize, blockSize);
  init_tables();
  if (w.nChunks <= 1 && (pp_applies(codec, (uint32_t)w.nBlocks, blockSize) || ppw_applies(codec, (uint32_t)w.nBlocks, blockSize))) return HSRLE_PATH_POSITION_PARALLEL;
  if (w.nChunks <= 1 && w.spSlots != 0 && split_encode_applies(codec, w.nBlocks, blockSize)) return HSRLE_PATH_SPLIT;
  if (w.nChunks <= 1 && run_list_codec(codec) && run_list_applies(w.nBlocks, blockSize, uncompressedSize, knob_u32("HSRLE_RUNLIST", 0u))) return HSRLE_PATH_RUN_LIST;
  return HSRLE_PATH_RING;
}

#ifndef HSRLE_BUILD_ID
#define HSRLE_BUILD_ID "unknown"
#endif
const char *hsrle_build_id(void) { return HSRLE_BUILD_ID; }

int hsrle_device_count(void)
{
  int n = 0;
  return (hipGetDeviceCount(&n) == hipSuccess) ? n : 0;
}

// ---- the low-entropy codec in its UNSECTIONED forms (src/rle.h:53-57, :90-93; rle8_low_entropy_cpu.c:6-124, rle8_low_entropy_short_cpu.c:16-124):
//      [u32 compressedSize][u32 uncompressedSize][info][one stream].  Many waves on the one stream: the input is cut at run boundaries, the stream
//      at arbitrary bytes whose symbol / code parity a short backward scan finds (hsrle_rle8m.hip.h, "the UNSECTIONED low-entropy streams"). ----
static uint32_t le_bounds(uint32_t n) { return n + (256 / 8) + 1 + 256 + 8u; }

// many waves for ONE stream (hsrle_rle8m.hip.h, "the UNSECTIONED low-entropy streams"): pieces of kLePiece input / stream bytes
constexpr uint32_t kLePiece = 16384u;
struct LePlan
{
  Workspace w;                   // scan levels for the piece sizes
  uint32_t pieces;
  uint64_t offTables, offTmpTables, offTmpInfo, offCuts, offStats, offSizes, offOffsets, offSlots, total;
};
static LePlan plan_le(uint64_t bytes, bool withSlots, uint32_t piece = kLePiece)
{
  LePlan p;
  p.pieces = (uint32_t)((bytes + piece - 1u) / piece);
  if (p.pieces == 0u) p.pieces = 1u;
  Workspace &w = p.w;
  w = Workspace{};
  w.nBlocks = p.pieces; w.chunk = p.pieces; w.nChunks = 1;
  w.t1 = (w.nBlocks + kScanTile - 1) / kScanTile;
  w.t2 = (w.t1 + kScanTile - 1) / kScanTile;
  w.t3 = (w.t2 + kScanTile - 1) / kScanTile;
  uint64_t at = 0;
  p.offTables = at; at += align_up(sizeof(Rle8mTables), 256);
  p.offTmpTables = at; at += align_up(sizeof(Rle8mTables), 256);   // (le_compress_with_info: the statistics pass that only feeds the cut finder)
  p.offTmpInfo = at; at += 512;
  p.offCuts = at; at += align_up(4ull * ((uint64_t)p.pieces + 1ull), 256);
  p.offStats = at; at += le_stats_ws_bytes(bytes);                      // per 4 KiB of input: last / first run start, and the start of the run that enters (launch_le_stats)
  p.offSizes = at; at += align_up(4ull * ((uint64_t)p.pieces + 1ull), 256);
  p.offOffsets = at; at += align_up(8ull * ((uint64_t)p.pieces + 2ull), 256);
  w.offL1 = at; at += align_up((w.t1 + 1) * 8ull, 256);
  w.offL2 = at; at += align_up((w.t2 + 1) * 8ull, 256);
  w.offL3 = at; at += align_up((w.t3 + 1) * 8ull, 256);
  p.offSlots = at; if (withSlots) at += align_up(2ull * bytes + 64ull, 256);
  p.total = at;
  return p;
}

// dOut: capacity >= le_bounds(n) (a stream that does not fit sets RLE8M_ERR_STREAM in *dStatus: the reference would write behind its
// caller's buffer there, rle8_low_entropy_cpu.c:476).  Only enqueues.  The stream's size is its first u32.
static int le_encode_async(const void *dIn, uint32_t n, void *dOut, uint64_t outCapacity, void *dWs, uint64_t wsSize, uint32_t *dStatus, uint32_t maxLen, uint32_t onlyMax, hipStream_t st)
{
  if (!dIn || !dOut || !dWs || n == 0)
    return HSRLE_ERR_ARGUMENT;
  if (outCapacity < le_bounds(n))
    return HSRLE_ERR_CAPACITY;
  const LePlan p = plan_le(n, true);
  if (wsSize < p.total)
    return HSRLE_ERR_CAPACITY;
  uint8_t *ws = (uint8_t *)dWs;
  Rle8mTables *t = (Rle8mTables *)(ws + p.offTables);
  uint32_t *cuts = (uint32_t *)(ws + p.offCuts), *sizes = (uint32_t *)(ws + p.offSizes);
  uint64_t *offsets = (uint64_t *)(ws + p.offOffsets);
  if (zero_async(t, sizeof(Rle8mTables), st) != hipSuccess || (dStatus && zero_async(dStatus, 8, st) != hipSuccess))
    return HSRLE_ERR_DEVICE;
  const uint32_t *runStart4 = nullptr;
  if (launch_le_stats((const uint8_t *)dIn, n, t, maxLen, ws + p.offStats, st, 32768u, &runStart4) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_rle8m_info, dim3(1), dim3(256), 0, st, t, 1u, ws + p.offTmpInfo, onlyMax);
  hipLaunchKernelGGL(k_le_move_info, dim3(1), dim3(64), 0, st, (const uint8_t *)(ws + p.offTmpInfo), (const Rle8mTables *)t, (uint8_t *)dOut);
  hipLaunchKernelGGL(k_le_cuts, dim3((p.pieces + 63u) / 64u), dim3(64), 0, st, (const uint8_t *)dIn, n, kLePiece, p.pieces, runStart4, (const Rle8mTables *)t, maxLen, cuts);
  hipLaunchKernelGGL(k_le_encode_wave, dim3(p.pieces), dim3(64), 0, st, (const uint8_t *)dIn, n, (const uint32_t *)cuts, p.pieces, (const Rle8mTables *)t, ws + p.offSlots, sizes, maxLen);
  if (scan_sizes(sizes, p.pieces, offsets, ws, p.w, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_le_place, dim3((p.pieces + 3u) / 4u), dim3(256), 0, st, (const uint8_t *)(ws + p.offSlots), (const uint32_t *)cuts, (const uint64_t *)offsets, (const Rle8mTables *)t,
                     (uint8_t *)dOut, outCapacity, n, p.pieces, dStatus);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

// dStream: the stream (nothing at or beyond streamSize is read), dataStart = 8 + 33 + listed symbols (the caller has read the header).  dStatus: two
// words ([0] error bits, [1] "a piece's parity could not be found within kLeCarryLimit bytes").  onePiece: decode with ONE wave (the fallback).
static int le_decode_async(const void *dStream, uint32_t streamSize, uint32_t dataStart, uint32_t expOut, void *dOut, uint64_t outCapacity, void *dWs, uint64_t wsSize, uint32_t *dStatus,
                           bool onePiece, hipStream_t st)
{
  if (!dStream || !dOut || !dWs || streamSize < dataStart || expOut == 0 || outCapacity < expOut || !dStatus)
    return HSRLE_ERR_ARGUMENT;
  const uint32_t body = streamSize - dataStart;
  const uint32_t G = onePiece ? (((body + 63u) / 64u) * 64u + 64u) : kLePiece;
  const LePlan p = plan_le(body, false, G);
  if (wsSize < p.total)
    return HSRLE_ERR_CAPACITY;
  uint8_t *ws = (uint8_t *)dWs;
  uint32_t *carry = (uint32_t *)(ws + p.offCuts), *sizes = (uint32_t *)(ws + p.offSizes);
  uint64_t *outStart = (uint64_t *)(ws + p.offOffsets);
  if (zero_async(dStatus, 8, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_le_carry, dim3((p.pieces + 63u) / 64u), dim3(64), 0, st, (const uint8_t *)dStream, (uint64_t)streamSize, G, p.pieces, carry, dStatus);
  hipLaunchKernelGGL(k_le_decode_wave<true>, dim3(p.pieces), dim3(64), 0, st, (const uint8_t *)dStream, (uint64_t)streamSize, (uint8_t *)dOut, dStatus, expOut, G, p.pieces,
                     (const uint32_t *)carry, (const uint64_t *)nullptr, sizes);
  if (scan_sizes(sizes, p.pieces, outStart, ws, p.w, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  hipLaunchKernelGGL(k_le_decode_wave<false>, dim3(p.pieces), dim3(64), 0, st, (const uint8_t *)dStream, (uint64_t)streamSize, (uint8_t *)dOut, dStatus, expOut, G, p.pieces,
                     (const uint32_t *)carry, (const uint64_t *)outStart, (uint32_t *)nullptr);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

static uint32_t le_mono_compress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize, uint32_t maxLen, uint32_t onlyMax)
{
  // argument checks of the reference (rle8_low_entropy_cpu.c:13-14; the Short form checks against the same bound, rle8_low_entropy_short_cpu.c:23)
  if (pIn == nullptr || inSize == 0 || pOut == nullptr || outSize < le_bounds(inSize) || !device_ok())
    return 0;
  const LePlan p = plan_le(inSize, true);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)inSize + 64) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)outSize + 64) || !grow(&D.ws, &D.wsSize, p.total))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess)
    return 0;
  if (le_encode_async(D.monoIn, inSize, D.monoOut, outSize, D.ws, D.wsSize, dStatus, maxLen, onlyMax, nullptr) != HSRLE_OK)
    return 0;
  uint32_t status = 1, size = 0;
  if (hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost) != hipSuccess || status != 0)
    return 0;
  if (hipMemcpy(&size, D.monoOut, 4, hipMemcpyDeviceToHost) != hipSuccess || size < 8u + 33u || size > outSize)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, size, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return size;
}

static uint32_t le_mono_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  // argument + header checks of the reference (rle8_low_entropy_cpu.c:98-107)
  if (pIn == nullptr || pOut == nullptr || inSize < 8u + 33u || outSize == 0 || !device_ok())
    return 0;
  uint32_t expIn, expOut;
  memcpy(&expIn, pIn, 4); memcpy(&expOut, pIn + 4, 4);
  if (expOut > outSize || expIn > inSize || expIn < 8u + 33u || expOut == 0 || expIn > 0xFFFFFF00u)
    return 0;
  uint32_t listed = pIn[8 + 32];
  if (listed == 0u) listed = 255u;
  const uint32_t dataStart = 8u + 33u + listed;
  if (dataStart > expIn)
    return 0;
  const LePlan p = plan_le(expIn - dataStart, false);
  const LePlan p1 = plan_le(expIn - dataStart, false, expIn + 128u);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)expIn + 256) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)expOut + 64) || !grow(&D.ws, &D.wsSize, p.total > p1.total ? p.total : p1.total))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  if (hipMemcpy(D.monoIn, pIn, expIn, hipMemcpyHostToDevice) != hipSuccess || hipMemset((uint8_t *)D.monoIn + expIn, 0, 128) != hipSuccess)
    return 0;
  uint32_t status[2] = { 1, 0 };
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (le_decode_async(D.monoIn, expIn, dataStart, expOut, D.monoOut, expOut, D.ws, D.wsSize, dStatus, attempt != 0, nullptr) != HSRLE_OK)
      return 0;
    if (hipMemcpy(status, dStatus, 8, hipMemcpyDeviceToHost) != hipSuccess)
      return 0;
    if (status[1] == 0u) break;                                          // (status[1]: a degenerate stretch of flagged-valued bytes -- once more with one wave)
  }
  if (status[0] != 0u || status[1] != 0u)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, expOut, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return expOut;
}

// ---- the split-phase helpers (src/rle.h:67-96): statistics, header writer / reader and the stream bodies as separate calls, tables through host structs ----

// rle8_low_entropy_get_compress_info[_only_max_frequency] (rle8_low_entropy_cpu.c:254-439): the statistics and table kernels of le_encode_async, tables back to the host
static bool le_get_info(const uint8_t *pIn, uint32_t inSize, rle8_low_entropy_compress_info_t *info, uint32_t onlyMax)
{
  if (pIn == nullptr || inSize == 0 || info == nullptr || !device_ok())
    return false;
  const LePlan p = plan_le(inSize, false);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)inSize + 64) || !grow(&D.ws, &D.wsSize, p.total))
    return false;
  uint8_t *ws = (uint8_t *)D.ws;
  Rle8mTables *t = (Rle8mTables *)(ws + p.offTables);
  if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess || zero_async(t, sizeof(Rle8mTables), nullptr) != hipSuccess)
    return false;
  if (launch_le_stats((const uint8_t *)D.monoIn, inSize, t, 255u, ws + p.offStats, nullptr) != hipSuccess)
    return false;
  hipLaunchKernelGGL(k_rle8m_info, dim3(1), dim3(256), 0, nullptr, t, 1u, ws + p.offTmpInfo, onlyMax);
  Rle8mTables ht;
  uint8_t used = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpy(&ht, t, sizeof(ht), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(&used, ws + p.offTmpInfo + 12u + 32u, 1, hipMemcpyDeviceToHost) != hipSuccess)
    return false;
  for (int i = 0; i < 256; i++) { info->rle[i] = ht.rle[i] != 0; info->symbolsByProb[i] = ht.order[i]; }
  info->symbolCount = used;                                              // (a uint8: 256 symbols in use -> 0, rle8_low_entropy_cpu.c:333)
  return true;
}

// rle8_low_entropy[_short]_compress_with_info (rle8_low_entropy_cpu.c:474-543, rle8_low_entropy_short_cpu.c:128-198): the body for the caller's tables
static uint32_t le_compress_with_info(const uint8_t *pIn, uint32_t inSize, const rle8_low_entropy_compress_info_t *info, uint8_t *pOut, uint32_t outSize, uint32_t maxLen)
{
  if (pIn == nullptr || inSize == 0 || info == nullptr || pOut == nullptr || outSize < inSize || !device_ok())
    return 0;
  Rle8mTables ht;
  memset(&ht, 0, sizeof(ht));
  for (uint32_t i = 0; i < 256u; i++)
  {
    const uint8_t flag = ((const uint8_t *)info->rle)[i] ? 1 : 0;
    ht.rle[i] = flag; ht.order[i] = info->symbolsByProb[i];
    ht.rleBits[i >> 5] |= (uint32_t)flag << (i & 31u);
  }
  ht.listed = info->symbolCount ? info->symbolCount : 255u;
  ht.headerSize = 12u + 33u + ht.listed;                                 // (as k_rle8m_info counts it: an rle8m header of one section)
  const uint32_t H = ht.headerSize - 4u;
  const LePlan p = plan_le(inSize, true);
  const uint64_t cap = 2ull * inSize + 512ull;                           // a body is at most twice its input (every byte a flagged symbol with its code)
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)inSize + 64) || !grow(&D.monoOut, &D.monoOutSize, cap + 64) || !grow(&D.ws, &D.wsSize, p.total))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  uint8_t *ws = (uint8_t *)D.ws;
  Rle8mTables *t = (Rle8mTables *)(ws + p.offTables);
  uint32_t *cuts = (uint32_t *)(ws + p.offCuts), *sizes = (uint32_t *)(ws + p.offSizes);
  uint64_t *offsets = (uint64_t *)(ws + p.offOffsets);
  if (hipMemcpy(D.monoIn, pIn, inSize, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(t, &ht, sizeof(ht), hipMemcpyHostToDevice) != hipSuccess ||
      zero_async(dStatus, 8, nullptr) != hipSuccess)
    return 0;
  {
    // (the cuts need the pieces' run starts: the statistics pass into a scratch table -- the caller's tables in `t` stay as they are)
    Rle8mTables *scratchT = (Rle8mTables *)(ws + p.offTmpTables);
    const uint32_t *runStart4 = nullptr;
    if (zero_async(scratchT, sizeof(Rle8mTables), nullptr) != hipSuccess || launch_le_stats((const uint8_t *)D.monoIn, inSize, scratchT, maxLen, ws + p.offStats, nullptr, 32768u, &runStart4) != hipSuccess)
      return 0;
    hipLaunchKernelGGL(k_le_cuts, dim3((p.pieces + 63u) / 64u), dim3(64), 0, nullptr, (const uint8_t *)D.monoIn, inSize, kLePiece, p.pieces, runStart4, (const Rle8mTables *)t, maxLen, cuts);
  }
  hipLaunchKernelGGL(k_le_encode_wave, dim3(p.pieces), dim3(64), 0, nullptr, (const uint8_t *)D.monoIn, inSize, (const uint32_t *)cuts, p.pieces, (const Rle8mTables *)t, ws + p.offSlots, sizes, maxLen);
  if (scan_sizes(sizes, p.pieces, offsets, ws, p.w, nullptr) != hipSuccess)
    return 0;
  hipLaunchKernelGGL(k_le_place, dim3((p.pieces + 3u) / 4u), dim3(256), 0, nullptr, (const uint8_t *)(ws + p.offSlots), (const uint32_t *)cuts, (const uint64_t *)offsets, (const Rle8mTables *)t,
                     (uint8_t *)D.monoOut, cap, inSize, p.pieces, dStatus);
  uint32_t status = 1, size = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost) != hipSuccess || status != 0u)
    return 0;
  if (hipMemcpy(&size, D.monoOut, 4, hipMemcpyDeviceToHost) != hipSuccess || size < H || size - H > outSize)
    return 0;
  if (size > H && hipMemcpy(pOut, (const uint8_t *)D.monoOut + H, size - H, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return size - H;
}

// rle8_low_entropy[_short]_decompress_with_info (rle8_low_entropy_cpu.c:930-1022, rle8_low_entropy_short_cpu.c:440-534): the body becomes a stream again --
// header and the tables' symbols in front of it, in device memory -- and takes the way of le_mono_decompress (the decode kernels read their tables from the stream)
static uint32_t le_decompress_with_info(const uint8_t *pIn, const uint8_t *pEnd, const rle8_low_entropy_decompress_info_t *info, uint8_t *pOut, uint32_t expOut)
{
  if (pIn == nullptr || pEnd == nullptr || pEnd < pIn || info == nullptr || pOut == nullptr || expOut == 0 || !device_ok())
    return 0;
  const uint64_t body64 = (uint64_t)(pEnd - pIn);
  constexpr uint32_t dataStart = 8u + 33u + 255u;
  if (body64 > 0xFFFFFF00ull - dataStart)
    return 0;
  const uint32_t body = (uint32_t)body64, expIn = dataStart + body;
  uint8_t head[dataStart];
  memset(head, 0, sizeof(head));
  memcpy(head, &expIn, 4); memcpy(head + 4, &expOut, 4);
  for (uint32_t i = 0; i < 256u; i++)
    if (((const uint8_t *)info->rle)[i]) head[8u + (i >> 3)] |= (uint8_t)(1u << (i & 7u));
  head[8u + 32u] = 255u;
  // the symbols in the order of their codes' counts: the inverse of symbolToCount, which must be a permutation (what read_decompress_info produces)
  bool seen[256] = { false };
  for (uint32_t sym = 0; sym < 256u; sym++)
  {
    const uint32_t c = info->symbolToCount[sym];
    if (seen[c]) return 0;
    seen[c] = true;
    if (c < 255u) head[8u + 33u + c] = (uint8_t)sym;
  }
  const LePlan p = plan_le(body, false);
  const LePlan p1 = plan_le(body, false, expIn + 128u);
  DeviceState &D = this_device();
  std::lock_guard<std::mutex> lock(D.mu);
  if (!grow(&D.monoIn, &D.monoInSize, (uint64_t)expIn + 256) || !grow(&D.monoOut, &D.monoOutSize, (uint64_t)expOut + 64) || !grow(&D.ws, &D.wsSize, p.total > p1.total ? p.total : p1.total))
    return 0;
  if (!D.monoAux && hipMalloc(&D.monoAux, 256) != hipSuccess)
    return 0;
  uint32_t *dStatus = (uint32_t *)((uint8_t *)D.monoAux + 64);
  if (hipMemcpy(D.monoIn, head, dataStart, hipMemcpyHostToDevice) != hipSuccess || (body != 0u && hipMemcpy((uint8_t *)D.monoIn + dataStart, pIn, body, hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemset((uint8_t *)D.monoIn + expIn, 0, 128) != hipSuccess)
    return 0;
  uint32_t status[2] = { 1, 0 };
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (le_decode_async(D.monoIn, expIn, dataStart, expOut, D.monoOut, expOut, D.ws, D.wsSize, dStatus, attempt != 0, nullptr) != HSRLE_OK)
      return 0;
    if (hipMemcpy(status, dStatus, 8, hipMemcpyDeviceToHost) != hipSuccess)
      return 0;
    if (status[1] == 0u) break;
  }
  if (status[0] != 0u || status[1] != 0u)
    return 0;
  if (hipMemcpy(pOut, D.monoOut, expOut, hipMemcpyDeviceToHost) != hipSuccess)
    return 0;
  return expOut;
}

// device-resident forms (benchmarks, pipelines that keep the data on the GPU).  variant: bit 0 the Short form, bit 1 only_max_frequency.
uint64_t hsrle_low_entropy_workspace_size(uint32_t inSize) { return inSize == 0 ? 0 : plan_le(inSize, true).total; }
int hsrle_low_entropy_compress_dev_async(const void *dIn, uint32_t inSize, int variant, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *dStatus, void *stream)
{
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  if (variant < 0 || variant > 3) return HSRLE_ERR_ARGUMENT;
  return le_encode_async(dIn, inSize, dOut, outCapacity, dWorkspace, workspaceSize, dStatus, (variant & 1) ? 32u : 255u, (variant & 2) ? 1u : 0u, (hipStream_t)stream);
}
// reads the stream's header (synchronises the stream once), decodes, reads the verdict: HSRLE_OK / HSRLE_ERR_FORMAT / ...
int hsrle_low_entropy_decompress_dev(const void *dStream, uint64_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *pUncompressedSize, void *stream)
{
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  if (!dStream || !dOut || !dWorkspace || streamSize < 8u + 33u || workspaceSize < 256u) return HSRLE_ERR_ARGUMENT;
  uint8_t head[48];
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(head, dStream, 48, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return HSRLE_ERR_DEVICE;
  uint32_t expIn, expOut;
  memcpy(&expIn, head, 4); memcpy(&expOut, head + 4, 4);
  uint32_t listed = head[40]; if (listed == 0u) listed = 255u;
  const uint32_t dataStart = 8u + 33u + listed;
  if ((uint64_t)expIn > streamSize || dataStart > expIn || expOut == 0u || expIn > 0xFFFFFF00u) return HSRLE_ERR_FORMAT;   // (the last: the piece arithmetic of le_decode_async is 32 bit, as on the host path)
  if (outCapacity < expOut) return HSRLE_ERR_CAPACITY;
  uint32_t *dStatus = (uint32_t *)dWorkspace;                              // the first 256 bytes of the workspace: the status words
  uint32_t status[2] = { 1, 0 };
  for (int attempt = 0; attempt < 2; attempt++)
  {
    const int rc = le_decode_async(dStream, expIn, dataStart, expOut, dOut, outCapacity, (uint8_t *)dWorkspace + 256, workspaceSize - 256, dStatus, attempt != 0, st);
    if (rc != HSRLE_OK) return rc;
    if (hipMemcpyAsync(status, dStatus, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return HSRLE_ERR_DEVICE;
    if (status[1] == 0u) break;
  }
  if (status[0] != 0u || status[1] != 0u) return HSRLE_ERR_FORMAT;
  if (pUncompressedSize) *pUncompressedSize = expOut;
  return HSRLE_OK;
}
uint64_t hsrle_low_entropy_decompress_workspace_size(uint64_t streamSize) { const uint64_t a = plan_le(streamSize, false).total, b = plan_le(streamSize, false, (uint32_t)(streamSize > 0xFFFFFF00ull ? 0xFFFFFF00ull : streamSize) + 128u).total; return 256u + (a > b ? a : b); }

uint32_t rle8_low_entropy_compress_bounds(const uint32_t inSize) { return le_bounds(inSize); }
uint32_t rle8_low_entropy_short_compress_bounds(const uint32_t inSize) { return le_bounds(inSize); }
uint32_t rle8_low_entropy_decompressed_size(const uint8_t *pIn, const uint32_t inSize)
{
  if (pIn == nullptr || inSize < 8u) return 0;                                               // rle8_low_entropy_cpu.c:88-94
  uint32_t v; memcpy(&v, pIn + 4, 4); return v;
}
uint32_t rle8_low_entropy_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_compress(pIn, inSize, pOut, outSize, 255u, 0u); }
uint32_t rle8_low_entropy_compress_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_compress(pIn, inSize, pOut, outSize, 255u, 1u); }
uint32_t rle8_low_entropy_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_decompress(pIn, inSize, pOut, outSize); }
uint32_t rle8_low_entropy_short_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_compress(pIn, inSize, pOut, outSize, 32u, 0u); }
uint32_t rle8_low_entropy_short_compress_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_compress(pIn, inSize, pOut, outSize, 32u, 1u); }
uint32_t rle8_low_entropy_short_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return le_mono_decompress(pIn, inSize, pOut, outSize); }

bool rle8_low_entropy_get_compress_info(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_compress_info_t *pCompressInfo) { return le_get_info(pIn, inSize, pCompressInfo, 0u); }
bool rle8_low_entropy_get_compress_info_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_compress_info_t *pCompressInfo) { return le_get_info(pIn, inSize, pCompressInfo, 1u); }
uint32_t rle8_low_entropy_write_compress_info(rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize)
{
  // host only (rle8_low_entropy_cpu.c:441-472): 32 flag bytes, the count byte, the symbols (a count of 0 stands for 256 in use: 255 are written)
  if (pCompressInfo == nullptr || pOut == nullptr || outSize < 256u / 8u + 256u + 1u)
    return 0;
  memset(pOut, 0, 32);
  for (uint32_t i = 0; i < 256u; i++)
    if (((const uint8_t *)pCompressInfo->rle)[i]) pOut[i >> 3] |= (uint8_t)(1u << (i & 7u));
  pOut[32] = pCompressInfo->symbolCount;
  const uint32_t listed = pCompressInfo->symbolCount ? pCompressInfo->symbolCount : 255u;
  memcpy(pOut + 33, pCompressInfo->symbolsByProb, listed);
  return 33u + listed;
}
uint32_t rle8_low_entropy_read_decompress_info(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_decompress_info_t *pDecompressInfo)
{
  // host only (rle8_low_entropy_cpu.c:545-605): the listed symbols get the counts 0, 1, ... in the order they are listed, the others the counts behind them in
  // ascending order.  (The reference does not look at inSize beyond "not 0"; here a header that does not fit is refused.)
  if (pIn == nullptr || pDecompressInfo == nullptr || inSize < 33u)
    return 0;
  uint32_t listed = pIn[32];
  if (listed == 0u) listed = 255u;
  if (inSize < 33u + listed)
    return 0;
  bool isListed[256] = { false };
  for (uint32_t i = 0; i < 256u; i++) pDecompressInfo->rle[i] = ((pIn[i >> 3] >> (i & 7u)) & 1u) != 0u;
  for (uint32_t i = 0; i < listed; i++) { pDecompressInfo->symbolToCount[pIn[33u + i]] = (uint8_t)i; isListed[pIn[33u + i]] = true; }
  uint32_t next = listed;
  for (uint32_t sym = 0; sym < 256u; sym++)
    if (!isListed[sym]) pDecompressInfo->symbolToCount[sym] = (uint8_t)next++;
  return 33u + listed;
}
uint32_t rle8_low_entropy_compress_with_info(const uint8_t *pIn, const uint32_t inSize, const rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize) { return le_compress_with_info(pIn, inSize, pCompressInfo, pOut, outSize, 255u); }
uint32_t rle8_low_entropy_short_compress_with_info(const uint8_t *pIn, const uint32_t inSize, const rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize) { return le_compress_with_info(pIn, inSize, pCompressInfo, pOut, outSize, 32u); }
uint32_t rle8_low_entropy_decompress_with_info(const uint8_t *pIn, const uint8_t *pEnd, const rle8_low_entropy_decompress_info_t *pDecompressInfo, uint8_t *pOut, const uint32_t expectedOutSize) { return le_decompress_with_info(pIn, pEnd, pDecompressInfo, pOut, expectedOutSize); }
uint32_t rle8_low_entropy_short_decompress_with_info(const uint8_t *pIn, const uint8_t *pEnd, const rle8_low_entropy_decompress_info_t *pDecompressInfo, uint8_t *pOut, const uint32_t expectedOutSize) { return le_decompress_with_info(pIn, pEnd, pDecompressInfo, pOut, expectedOutSize); }

// ---- rle8m: names of the reference's GPU decode path (src/rle.h:464-466) and of its CPU twin (src/rle.h:63) ----
bool rle8m_opencl_init(const size_t inputDataSize, const size_t outputDataSize, const size_t maxSubsectionCount)
{
  (void)inputDataSize; (void)outputDataSize; (void)maxSubsectionCount;    // device buffers are (re)sized by the call that needs them
  return device_ok();
}
void rle8m_opencl_destroy(void) {}
uint32_t rle8m_opencl_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return rle8m_mono_decompress(pIn, inSize, pOut, outSize); }
uint32_t rle8m_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return rle8m_mono_decompress(pIn, inSize, pOut, outSize); }

uint32_t rle8m_compress_bounds(const uint32_t subSections, const uint32_t inSize) { return rle8m_bounds(subSections, inSize); }
uint32_t rle8m_compress(const uint32_t subSections, const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return rle8m_mono_compress(subSections, pIn, inSize, pOut, outSize); }

uint64_t hsrle_rle8m_compress_workspace_size(uint32_t inSize, uint32_t sections) { return (inSize == 0 || sections == 0) ? 0 : plan_rle8m(inSize, sections).total; }

int hsrle_rle8m_compress_dev_async(const void *dIn, uint32_t inSize, uint32_t sections, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *dStatus, void *stream)
{
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  return rle8m_encode_async(dIn, inSize, sections, dOut, outCapacity, dWorkspace, workspaceSize, dStatus, (hipStream_t)stream);
}

int hsrle_rle8m_info_dev(const void *dStream, uint64_t streamSize, hsrle_rle8m_info_t *pInfo, void *stream)
{
  if (!dStream || !pInfo || streamSize < 12) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  uint32_t h[3];
  if (hipMemcpyAsync(h, dStream, 12, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  if (h[0] > streamSize || h[2] == 0 || h[0] < 12ull + 4ull * (h[2] - 1) + 33ull) return HSRLE_ERR_FORMAT;
  pInfo->compressedSize = h[0]; pInfo->uncompressedSize = h[1]; pInfo->sections = h[2];
  return HSRLE_OK;
}

int hsrle_rle8m_decompress_dev_async(const void *dStream, const hsrle_rle8m_info_t *info, void *dOut, uint64_t outCapacity, uint32_t *dStatus, void *stream)
{
  if (!info) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  return rle8m_decode_async(dStream, info->compressedSize, info->uncompressedSize, info->sections, dOut, outCapacity, dStatus, (hipStream_t)stream);
}

uint32_t rle_compress_bounds(const uint32_t inSize) { return bounds32(inSize); }
uint32_t rle_decompress_additional_size(void) { return 128; }

uint32_t hsrle_compress_mono(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize) { return mono_compress(codec, pIn, inSize, pOut, outSize); }
uint32_t hsrle_decompress_mono(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize) { return mono_decompress(codec, pIn, inSize, pOut, outSize); }

#define HSRLE_DEF_PAIR(name, id)                                                                                                                             \
  uint32_t name##_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(id, pIn, inSize, pOut, outSize); } \
  uint32_t name##_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_decompress(id, pIn, inSize, pOut, outSize); }
#define HSRLE_DEF_WIDTH(W, base)                                                                                               \
  HSRLE_DEF_PAIR(rle##W##_sym, base + 0) HSRLE_DEF_PAIR(rle##W##_sym_packed, base + 1) HSRLE_DEF_PAIR(rle##W##_3symlut_sym, base + 2) \
  HSRLE_DEF_PAIR(rle##W##_7symlut_sym, base + 3) HSRLE_DEF_PAIR(rle##W##_byte, base + 4) HSRLE_DEF_PAIR(rle##W##_byte_packed, base + 5) \
  HSRLE_DEF_PAIR(rle##W##_3symlut_byte, base + 6) HSRLE_DEF_PAIR(rle##W##_7symlut_byte, base + 7)

uint32_t rle8_multi_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(HSRLE_RLE8_MULTI, pIn, inSize, pOut, outSize); }
uint32_t rle8_single_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(HSRLE_RLE8_SINGLE, pIn, inSize, pOut, outSize); }
uint32_t rle8_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_decompress(HSRLE_RLE8_MULTI, pIn, inSize, pOut, outSize); }
uint32_t rle8_packed_multi_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(HSRLE_RLE8_PACKED_MULTI, pIn, inSize, pOut, outSize); }
uint32_t rle8_packed_single_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(HSRLE_RLE8_PACKED_SINGLE, pIn, inSize, pOut, outSize); }
uint32_t rle8_packed_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_decompress(HSRLE_RLE8_PACKED_MULTI, pIn, inSize, pOut, outSize); }
HSRLE_DEF_PAIR(rle8_3symlut, HSRLE_RLE8_3SYMLUT)
HSRLE_DEF_PAIR(rle8_7symlut, HSRLE_RLE8_7SYMLUT)
HSRLE_DEF_WIDTH(16, 6)
HSRLE_DEF_WIDTH(24, 14)
HSRLE_DEF_WIDTH(32, 22)
HSRLE_DEF_WIDTH(48, 30)
HSRLE_DEF_WIDTH(64, 38)
HSRLE_DEF_PAIR(rle128_sym, HSRLE_RLE128_SYM)
HSRLE_DEF_PAIR(rle128_sym_packed, HSRLE_RLE128_SYM_PACKED)
HSRLE_DEF_PAIR(rle128_byte, HSRLE_RLE128_BYTE)
HSRLE_DEF_PAIR(rle128_byte_packed, HSRLE_RLE128_BYTE_PACKED)
// Short family
#define HSRLE_DEF_SHORT_WIDTH(W, base)                                                                                                   \
  HSRLE_DEF_PAIR(rle##W##_sym_short, base + 0) HSRLE_DEF_PAIR(rle##W##_1symlut_sym_short, base + 1) HSRLE_DEF_PAIR(rle##W##_3symlut_sym_short, base + 2) \
  HSRLE_DEF_PAIR(rle##W##_7symlut_sym_short, base + 3) HSRLE_DEF_PAIR(rle##W##_byte_short, base + 4) HSRLE_DEF_PAIR(rle##W##_1symlut_byte_short, base + 5) \
  HSRLE_DEF_PAIR(rle##W##_3symlut_byte_short, base + 6) HSRLE_DEF_PAIR(rle##W##_7symlut_byte_short, base + 7)
HSRLE_DEF_PAIR(rle8_multi_short, HSRLE_RLE8_MULTI_SHORT)
HSRLE_DEF_PAIR(rle8_1symlut_short, HSRLE_RLE8_1SYMLUT_SHORT)
HSRLE_DEF_PAIR(rle8_3symlut_short, HSRLE_RLE8_3SYMLUT_SHORT)
HSRLE_DEF_PAIR(rle8_7symlut_short, HSRLE_RLE8_7SYMLUT_SHORT)
HSRLE_DEF_SHORT_WIDTH(16, 54)
HSRLE_DEF_SHORT_WIDTH(24, 62)
HSRLE_DEF_SHORT_WIDTH(32, 70)
HSRLE_DEF_SHORT_WIDTH(48, 78)
HSRLE_DEF_SHORT_WIDTH(64, 86)
// Greedy encoders: only a compress entry point each (src/rle.h:398-416); their streams go to rle{W}_{K}symlut_byte_short_decompress
#define HSRLE_DEF_GREEDY(W, base)                                                                                                                                     \
  uint32_t rle##W##_1symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(base + 0, pIn, inSize, pOut, outSize); } \
  uint32_t rle##W##_3symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(base + 1, pIn, inSize, pOut, outSize); } \
  uint32_t rle##W##_7symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize) { return mono_compress(base + 2, pIn, inSize, pOut, outSize); }
HSRLE_DEF_GREEDY(16, 94)
HSRLE_DEF_GREEDY(24, 97)
HSRLE_DEF_GREEDY(32, 100)
HSRLE_DEF_GREEDY(48, 103)
HSRLE_DEF_GREEDY(64, 106)
HSRLE_DEF_PAIR(rle8_single_short, HSRLE_RLE8_SINGLE_SHORT)

// ---- container API ----

uint64_t hsrle_container_bound(uint64_t inSize, uint32_t blockSize)
{
  if (blockSize == 0) blockSize = HSRLE_DEFAULT_BLOCK_SIZE;
  if (!valid_block_size(blockSize) || inSize == 0) return 0;
  const uint64_t nb = block_count(inSize, blockSize);
  return HSRLE_CONTAINER_HEADER_SIZE + 8ull * (nb + 1) + nb * (uint64_t)slot_stride(blockSize) + HSRLE_CONTAINER_TAIL_PAD;
}

uint64_t hsrle_compress_workspace_size(uint64_t inSize, uint32_t blockSize)
{
  if (blockSize == 0) blockSize = HSRLE_DEFAULT_BLOCK_SIZE;
  if (!valid_block_size(blockSize) || inSize == 0) return 0;
  return plan_workspace(inSize, blockSize).total;
}

uint64_t hsrle_compress_workspace_size_codec(int codec, uint64_t inSize, uint32_t blockSize)
{
  if (blockSize == 0) blockSize = HSRLE_DEFAULT_BLOCK_SIZE;
  if (!valid_block_size(blockSize) || inSize == 0 || codec < 0 || codec >= kCodecCount) return 0;
  init_tables();
  // (round 6: 8 bit Single and 128 bit containers of <= 4 KiB blocks are position-parallel -- no split regions; rle8_single_short and the Greedy codecs with a one-symbol list still take them)
  const bool split = split_codec_small(codec) && blockSize <= 4096u && !pp_applies(codec, (uint32_t)block_count(inSize, blockSize), blockSize);
  // (... and the windowed encoders take blocks above 4 KiB without the split regions: 1.05 x the input instead of 3.4 x)
  return plan_workspace(inSize, blockSize, split, block_count(inSize, blockSize) <= 0xFFFFFFF0ull && ppw_applies(codec, (uint32_t)block_count(inSize, blockSize), blockSize)).total;
}

int hsrle_compress_dev_async(int codec, const void *dIn, uint64_t inSize, void *dOut, uint64_t outCapacity, uint32_t blockSize, void *dWorkspace,
                             uint64_t workspaceSize, void *stream)
{
  return compress_async(codec, dIn, inSize, dOut, outCapacity, blockSize, dWorkspace, workspaceSize, (hipStream_t)stream);
}

int hsrle_compress_dev(int codec, const void *dIn, uint64_t inSize, void *dOut, uint64_t outCapacity, uint32_t blockSize, uint64_t *pContainerSize, void *stream)
{
  int rc = compress_async(codec, dIn, inSize, dOut, outCapacity, blockSize, nullptr, 0, (hipStream_t)stream);
  if (rc != HSRLE_OK) return rc;
  ContainerHeader h;
  if (hipMemcpyAsync(&h, dOut, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  static const char magic[8] = { 'H', 'S', 'R', 'L', 'E', 'K', 'I', 'T' };
  if (memcmp(h.magic, magic, 8) != 0)
  {
    // the split encode struck its own container (a per-lane chunk encoder missed its boundary run: k_split_verdict): the synchronous entry sees it here
    // and encodes the container again with one lane per block -- as the monolithic path falls back; the enqueue-only entry cannot
    rc = compress_async(codec, dIn, inSize, dOut, outCapacity, blockSize, nullptr, 0, (hipStream_t)stream, true);
    if (rc != HSRLE_OK) return rc;
    if (hipMemcpyAsync(&h, dOut, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
      return HSRLE_ERR_DEVICE;
    if (memcmp(h.magic, magic, 8) != 0) return HSRLE_ERR_DEVICE;
  }
  if (pContainerSize) *pContainerSize = h.totalSize;
  return HSRLE_OK;
}

int hsrle_container_info_host(const void *pContainer, uint64_t containerSize, hsrle_container_info_t *pInfo)
{
  if (!pContainer || !pInfo || containerSize < HSRLE_CONTAINER_HEADER_SIZE) return HSRLE_ERR_ARGUMENT;
  ContainerHeader h;
  memcpy(&h, pContainer, sizeof(h));
  return check_info(h, containerSize, pInfo);
}

int hsrle_container_info_dev(const void *dContainer, uint64_t containerSize, hsrle_container_info_t *pInfo, void *stream)
{
  if (!dContainer || !pInfo || containerSize < HSRLE_CONTAINER_HEADER_SIZE) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  ContainerHeader h;
  if (hipMemcpyAsync(&h, dContainer, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  return check_info(h, containerSize, pInfo);
}

int hsrle_decompress_blocks_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, void *dOut,
                                      uint64_t outCapacity, uint32_t *dStatus, void *stream)
{
  return decompress_blocks_async(dContainer, info, firstBlock, blockCount, dOut, outCapacity, dStatus, (hipStream_t)stream);
}

int hsrle_decompress_dev_async(const void *dContainer, const hsrle_container_info_t *info, void *dOut, uint64_t outCapacity, uint32_t *dStatus, void *stream)
{
  if (!info) return HSRLE_ERR_ARGUMENT;
  return decompress_blocks_async(dContainer, info, 0, info->blockCount, dOut, outCapacity, dStatus, (hipStream_t)stream);
}

uint32_t hsrle_split_sub_block_size(const hsrle_container_info_t *info, uint32_t subBlockSize)
{
  if (!info || !valid_block_size(info->blockSize)) return 0;
  return split_sub_block(info, subBlockSize);
}

uint64_t hsrle_decompress_split_workspace_size(const hsrle_container_info_t *info, uint32_t blockCount, uint32_t subBlockSize)
{
  if (!info || !valid_block_size(info->blockSize)) return 0;
  const uint32_t SB = split_sub_block(info, subBlockSize);
  if (SB == info->blockSize) return 0;
  if (SB == HSRLE_SPLIT_PACKET_LIST) return packet_list_counts_bytes(blockCount) + (uint64_t)blockCount * packet_list_cap(info->blockSize) * 8ull;
  return (uint64_t)blockCount * (info->blockSize / SB) * 4ull * kEntryRecDwords;
}

int hsrle_decompress_split_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, void *dOut, uint64_t outCapacity,
                                     uint32_t *dStatus, void *dWorkspace, uint64_t workspaceSize, uint32_t subBlockSize, void *stream)
{
  return decompress_split_async(dContainer, info, firstBlock, blockCount, dOut, outCapacity, dStatus, dWorkspace, workspaceSize, subBlockSize, (hipStream_t)stream);
}

#ifdef HSRLE_EXPERIMENTS   // the wave-per-block decoder exists in experiment builds only (csrc/experiments/hsrle_decode_wave.hip.h)
int hsrle_decompress_wave_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, void *dOut, uint64_t outCapacity,
                                    uint32_t *dStatus, void *stream)
{
  return decompress_wave_async(dContainer, info, firstBlock, blockCount, dOut, outCapacity, dStatus, (hipStream_t)stream);
}
#endif

int hsrle_decompress_dev(const void *dContainer, uint64_t containerSize, void *dOut, uint64_t outCapacity, uint64_t *pUncompressedSize, void *stream)
{
  hsrle_container_info_t info;
  int rc = hsrle_container_info_dev(dContainer, containerSize, &info, stream);
  if (rc != HSRLE_OK) return rc;

  // a container with too few blocks to fill the GPU with one lane per block is decoded split (one lane per sub-block):
  // this call owns its scratch, so it can afford the records
  // ... or, for blocks of up to 16 KiB, by one WAVE per block (experiments/hsrle_decode_wave.hip.h): no records, no second kernel
  // (opt-in: measured on the 88 MB frame it is the slower of the two -- 427 us against 189 us split: the packet hops of ONE lane per
  //  wave are a latency chain of ~1 us per packet that the other 63 lanes wait for, while the split decode's record walk runs 64 such
  //  chains per wave)
  static const int waveMode = kExperiments ? (int)knob_u32("HSRLE_WAVE_DECODE", 0) : 0;
  const bool wave = waveMode != 0 && info.blockCount < kWaveDecodeBelow && info.blockSize <= 16384u && info.codec < (uint32_t)kCodecCount;
  const uint64_t recBytes = (!wave && info.blockCount < 131072u) ? hsrle_decompress_split_workspace_size(&info, info.blockCount, 0) : 0;
  uint8_t *scratch = (uint8_t *)scratch_alloc(256 + recBytes, (hipStream_t)stream);
  uint32_t *dStatus = (uint32_t *)scratch;
  if (!dStatus) return HSRLE_ERR_DEVICE;
  uint32_t status = 0;
  bool ok = hipMemsetAsync(dStatus, 0, 4, (hipStream_t)stream) == hipSuccess;
  if (ok)
  {
    if (wave)
      rc = decompress_wave_async(dContainer, &info, 0, info.blockCount, dOut, outCapacity, dStatus, (hipStream_t)stream);
    else if (recBytes != 0)
      rc = decompress_split_async(dContainer, &info, 0, info.blockCount, dOut, outCapacity, dStatus, scratch + 256, recBytes, 0, (hipStream_t)stream);
    else
      rc = decompress_blocks_async(dContainer, &info, 0, info.blockCount, dOut, outCapacity, dStatus, (hipStream_t)stream);
    ok = hipMemcpyAsync(&status, dStatus, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess && hipStreamSynchronize((hipStream_t)stream) == hipSuccess;
  }
  scratch_free(dStatus, (hipStream_t)stream);
  if (rc != HSRLE_OK) return rc;
  if (!ok) return HSRLE_ERR_DEVICE;
  if (status != 0) return HSRLE_ERR_FORMAT;
  if (pUncompressedSize) *pUncompressedSize = info.uncompressedSize;
  return HSRLE_OK;
}

int hsrle_compress_host(int codec, const void *pIn, uint64_t inSize, void *pOut, uint64_t outCapacity, uint32_t blockSize, uint64_t *pContainerSize)
{
  if (!pIn || !pOut || inSize == 0) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  const uint64_t bound = hsrle_container_bound(inSize, blockSize);
  if (bound == 0) return HSRLE_ERR_ARGUMENT;

  void *dIn = nullptr, *dOut = nullptr;
  int rc = HSRLE_ERR_DEVICE;
  uint64_t total = 0;
  if (hipMalloc(&dIn, inSize) == hipSuccess && hipMalloc(&dOut, bound) == hipSuccess && hipMemcpy(dIn, pIn, inSize, hipMemcpyHostToDevice) == hipSuccess)
  {
    rc = hsrle_compress_dev(codec, dIn, inSize, dOut, bound, blockSize, &total, nullptr);
    if (rc == HSRLE_OK)
    {
      if (total > outCapacity) rc = HSRLE_ERR_CAPACITY;
      else if (hipMemcpy(pOut, dOut, total, hipMemcpyDeviceToHost) != hipSuccess) rc = HSRLE_ERR_DEVICE;
    }
  }
  if (dIn) (void)hipFree(dIn);
  if (dOut) (void)hipFree(dOut);
  if (rc == HSRLE_OK && pContainerSize) *pContainerSize = total;
  return rc;
}

int hsrle_decompress_host(const void *pContainer, uint64_t containerSize, void *pOut, uint64_t outCapacity, uint64_t *pUncompressedSize)
{
  hsrle_container_info_t info;
  int rc = hsrle_container_info_host(pContainer, containerSize, &info);
  if (rc != HSRLE_OK) return rc;
  if (!pOut || outCapacity < info.uncompressedSize) return HSRLE_ERR_CAPACITY;
  if (!device_ok()) return HSRLE_ERR_DEVICE;

  void *dIn = nullptr, *dOut = nullptr;
  rc = HSRLE_ERR_DEVICE;
  if (hipMalloc(&dIn, info.totalSize) == hipSuccess && hipMalloc(&dOut, info.uncompressedSize) == hipSuccess &&
      hipMemcpy(dIn, pContainer, info.totalSize, hipMemcpyHostToDevice) == hipSuccess)
  {
    uint64_t U = 0;
    rc = hsrle_decompress_dev(dIn, info.totalSize, dOut, info.uncompressedSize, &U, nullptr);
    if (rc == HSRLE_OK && hipMemcpy(pOut, dOut, U, hipMemcpyDeviceToHost) != hipSuccess) rc = HSRLE_ERR_DEVICE;
    if (rc == HSRLE_OK && pUncompressedSize) *pUncompressedSize = U;
  }
  if (dIn) (void)hipFree(dIn);
  if (dOut) (void)hipFree(dOut);
  return rc;
}

void hsrle_mono_tuning(uint32_t blockSize, uint32_t regionSize, uint32_t lookBack)
{
  g_monoTune[0] = blockSize; g_monoTune[1] = regionSize; g_monoTune[2] = lookBack;
}

void hsrle_mono_encode_stats(uint32_t stats[4])
{
  for (int k = 0; k < 4; k++) stats[k] = g_monoEncLast[k];
}

uint64_t hsrle_compress_mono_workspace_size(int codec, uint32_t inSize)
{
  if (codec < 0 || codec >= kCodecCount || inSize == 0 || inSize > (1u << 30) || mono_cut_long(codec) == 0u) return 0;
  return plan_mono_encode(inSize, codec).total;
}

int hsrle_compress_mono_dev(int codec, const void *dIn, uint32_t inSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *pStreamSize,
                            uint32_t *pChunks, void *stream)
{
  if (!dIn || !dOut || !dWorkspace || !pStreamSize || codec < 0 || codec >= kCodecCount || inSize == 0) return HSRLE_ERR_ARGUMENT;
  if (inSize > (1u << 30) || mono_cut_long(codec) == 0u) return HSRLE_ERR_UNSUPPORTED;
  if (outCapacity < bounds32(inSize)) return HSRLE_ERR_CAPACITY;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  const MonoEncPlan m = plan_mono_encode(inSize, codec);
  if (workspaceSize < m.total) return HSRLE_ERR_CAPACITY;
  return mono_encode_dev(codec, (const uint8_t *)dIn, inSize, (uint8_t *)dOut, (uint8_t *)dWorkspace, m, pStreamSize, pChunks, (hipStream_t)stream);
}

int hsrle_compress_mono_dev_async(int codec, const void *dIn, uint32_t inSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *dStreamSize,
                                  void *stream)
{
  if (!dIn || !dOut || !dWorkspace || codec < 0 || codec >= kCodecCount || inSize == 0) return HSRLE_ERR_ARGUMENT;
  if (inSize > (1u << 30) || mono_cut_long(codec) == 0u) return HSRLE_ERR_UNSUPPORTED;
  if (outCapacity < bounds32(inSize)) return HSRLE_ERR_CAPACITY;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  const MonoEncPlan m = plan_mono_encode(inSize, codec);
  if (!m.windowed) return HSRLE_ERR_UNSUPPORTED;                          // (the other codecs' flows read chunk counts and list verdicts back)
  if (workspaceSize < m.total) return HSRLE_ERR_CAPACITY;
  return mono_encode_dev(codec, (const uint8_t *)dIn, inSize, (uint8_t *)dOut, (uint8_t *)dWorkspace, m, nullptr, dStreamSize, (hipStream_t)stream);
}

uint64_t hsrle_decompress_mono_workspace_size(int codec, uint32_t uncompressedSize, uint32_t compressedSize)
{
  if (codec < 0 || codec >= kCodecCount || uncompressedSize == 0 || compressedSize < 10u) return 0;
  return plan_mono(codec, uncompressedSize, compressedSize, codec_header_size(codec)).total + 16u;   // (+ 16: a workspace that is not 16-byte aligned is aligned here)
}

// the workspace is cleared with 16-byte stores (mono_prepare): a caller that sub-allocates from a byte pool may hand over any address -- it is rounded up, what
// is left must still hold the plan (hsrle_decompress_mono_workspace_size() includes the 16 bytes)
static inline void mono_align_workspace(void *&dWorkspace, uint64_t &workspaceSize)
{
  const uint64_t skip = (16u - (uint32_t)((uintptr_t)dWorkspace & 15u)) & 15u;
  dWorkspace = (uint8_t *)dWorkspace + skip;
  workspaceSize = workspaceSize > skip ? workspaceSize - skip : 0u;
}

int hsrle_decompress_mono_dev(int codec, const void *dStream, uint32_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize,
                              uint32_t *pUncompressedSize, uint32_t *pStats, void *stream)
{
  if (!dStream || !dOut || !dWorkspace || codec < 0 || codec >= kCodecCount || streamSize < codec_header_size(codec) || ((uintptr_t)dStream & 127u) != 0u)
    return HSRLE_ERR_ARGUMENT;
  mono_align_workspace(dWorkspace, workspaceSize);
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  uint8_t h16[16] = { 0 };
  if (hipMemcpyAsync(h16, dStream, streamSize < 16u ? streamSize : 16u, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  MonoHeader mh;
  if (!mono_header(codec, h16, streamSize, outCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)outCapacity, &mh))
    return HSRLE_ERR_FORMAT;
  const MonoPlan m = plan_mono(mh.codec, mh.U, mh.C, mh.p0);
  if (workspaceSize < m.total) return HSRLE_ERR_CAPACITY;
  const int rc = mono_decode_dev(mh, (const uint8_t *)dStream, (uint8_t *)dOut, (uint8_t *)dWorkspace, m, pStats, (hipStream_t)stream);
  if (rc == HSRLE_OK && pUncompressedSize) *pUncompressedSize = mh.U;
  return rc;
}

int hsrle_decompress_mono_dev_async(int codec, const void *dStream, const uint8_t *pHeader16, uint32_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace,
                                    uint64_t workspaceSize, uint32_t *pUncompressedSize, uint32_t *dStatus, void *stream)
{
  // (the same smallest stream as the synchronous entry: the codec's header; pHeader16 holds the stream's first min(16, streamSize) bytes, zeros behind them)
  if (!dStream || !pHeader16 || !dOut || !dWorkspace || !dStatus || codec < 0 || codec >= kCodecCount || streamSize < codec_header_size(codec) || ((uintptr_t)dStream & 127u) != 0u)
    return HSRLE_ERR_ARGUMENT;
  mono_align_workspace(dWorkspace, workspaceSize);
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  MonoHeader mh;
  if (!mono_header(codec, pHeader16, streamSize, outCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)outCapacity, &mh))
    return HSRLE_ERR_FORMAT;
  const MonoPlan m = plan_mono(mh.codec, mh.U, mh.C, mh.p0);
  if (workspaceSize < m.total) return HSRLE_ERR_CAPACITY;
  MonoRun run;
  int rc = mono_prepare(mh, (const uint8_t *)dStream, (uint8_t *)dOut, (uint8_t *)dWorkspace, m, &run, (hipStream_t)stream);
  if (rc == HSRLE_OK) rc = mono_enqueue_first_try(mh, (uint8_t *)dWorkspace, m, run, (hipStream_t)stream);
  if (rc != HSRLE_OK) return rc;
  hipLaunchKernelGGL(k_mono_status, dim3(1), dim3(1), 0, (hipStream_t)stream, (const uint32_t *)run.ctrl, dStatus);
  if (hipGetLastError() != hipSuccess) return HSRLE_ERR_DEVICE;
  if (pUncompressedSize) *pUncompressedSize = mh.U;
  return HSRLE_OK;
}

int hsrle_hash_blocks_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, uint64_t *dHashes, void *stream)
{
  if (!dContainer || !info || !dHashes || (uint64_t)firstBlock + blockCount > info->blockCount) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  if (blockCount == 0) return HSRLE_OK;
  const uint8_t *container = (const uint8_t *)dContainer;
  const uint8_t *payload = container + HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)info->blockCount + 1ull);
  hipLaunchKernelGGL(k_hash_blocks, dim3((blockCount + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, payload, (const uint64_t *)(container + HSRLE_CONTAINER_HEADER_SIZE),
                     info->payloadSize, firstBlock, blockCount, dHashes);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

int hsrle_synth_dev_async(int kind, int symbolBytes, uint64_t seed, void *dOut, uint64_t size, void *stream)
{
  if (!dOut || size == 0 || (kind != HSRLE_SYNTH_RUNS && kind != HSRLE_SYNTH_VIDEO) || symbolBytes < 1 || symbolBytes > 16) return HSRLE_ERR_ARGUMENT;
  if (!device_ok()) return HSRLE_ERR_DEVICE;
  const uint64_t chunks = (size + kSynthChunk - 1) / kSynthChunk;
  hipLaunchKernelGGL(k_synth, dim3((uint32_t)((chunks + 63) / 64)), dim3(64), 0, (hipStream_t)stream, kind, symbolBytes, seed, (uint8_t *)dOut, size);
  return hipGetLastError() == hipSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

} // extern "C"
