// hsrle_launch.h -- host-side launch table: codec id -> kernel launcher.  The kernels are instantiated per symbol
// width in inst_w*.hip (one translation unit per width so the library builds in parallel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsrle {

struct DecodeArgs
{
  const uint8_t *payload;
  const uint64_t *offsets;
  const uint8_t *payloadEnd; // payload + payloadSize + tail pad: last readable byte + 1
  uint8_t *out;
  uint64_t U;
  uint32_t B, firstBlock, blockCount;
  uint32_t *status;
};

struct EncodeArgs
{
  const uint8_t *in;
  uint64_t U;
  uint32_t B, nBlocks;
  uint8_t *slots;
  uint32_t slotStride;
  uint32_t *sizes;
};

typedef hipError_t (*DecodeLaunch)(const DecodeArgs &, hipStream_t);
typedef hipError_t (*EncodeLaunch)(const EncodeArgs &, hipStream_t);

constexpr int kCodecCount = 50;
#ifndef HSRLE_DECODE_TILE
#define HSRLE_DECODE_TILE 128
#endif
#ifndef HSRLE_DECODE_RING
#define HSRLE_DECODE_RING 128
#endif
#ifndef HSRLE_DECODE_STEP
#define HSRLE_DECODE_STEP 64
#endif
constexpr int kDecodeStep = HSRLE_DECODE_STEP; // output bytes per lane and decode/top-up step (k_decode_blocks Q)
constexpr int kDecodeTile = HSRLE_DECODE_TILE; // bytes produced per lane and round (k_decode_blocks T)
constexpr int kDecodeRing = HSRLE_DECODE_RING; // per-lane stream ring in LDS (k_decode_blocks R)

void register_w8(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w16(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w24(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w32(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w48(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w64(DecodeLaunch *dec, EncodeLaunch *enc);
void register_w128(DecodeLaunch *dec, EncodeLaunch *enc);

template <typename KERNEL>
inline hipError_t launch_decode(KERNEL k, const DecodeArgs &a, hipStream_t st)
{
  const uint32_t grid = (a.blockCount + 63u) / 64u;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, st, a.payload, a.offsets, a.payloadEnd, a.out, a.U, a.B, a.firstBlock, a.blockCount, a.status);
  return hipGetLastError();
}

template <typename KERNEL>
inline hipError_t launch_encode(KERNEL k, const EncodeArgs &a, hipStream_t st)
{
  const uint32_t grid = (a.nBlocks + 63u) / 64u;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes);
  return hipGetLastError();
}

} // namespace hsrle
