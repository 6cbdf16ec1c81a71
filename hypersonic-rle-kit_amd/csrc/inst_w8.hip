// 8 bit symbols: rle8_multi / rle8_single / rle8_packed_multi / rle8_packed_single / rle8_{3,7}symlut
// (reference: src/rle.h:101-103, :173-175, :199-208).  rle8_decompress / rle8_packed_decompress decode both the multi
// and the single mode, so the Single codec ids share the multi decode kernels.
#include "hsrle_decode.hip.h"
#ifdef HSRLE_DEC8_PE
#include "experiments/hsrle_decode8pe.hip.h"
#endif
#include "hsrle_encode.hip.h"
#include "hsrle_encode8.hip.h"
#include "hsrle_encode8r.hip.h"
#ifdef HSRLE_EXPERIMENTS
#include "experiments/hsrle_encode8w.hip.h"        // one wave per block: bit-exact, measured slower
#endif
#include "hsrle_encode8s.hip.h"
#include "hsrle_encode_greedy.hip.h"
#include "hsrle_index.hip.h"
#include "hsrle_launch.h"

namespace hsrle {

// ids 0 / 1 (rle8_multi, rle8_packed_multi): their encoders only write mode 0 -> the kernel without the Single mode (k_decode_blocks SGL);
// ids 4 / 5 (the Single codecs) and any stream whose mode byte says 1 (hsrle_capi.hip: mono_decompress) -> the general kernel
// the list-free 8 bit decoders' tile row / step (A/B builds: -DHSRLE_DEC8_TILE=64 -DHSRLE_DEC8_STEP=64)
#ifndef HSRLE_DEC8_TILE
#define HSRLE_DEC8_TILE HSRLE_DECODE_TILE
#endif
#ifndef HSRLE_DEC8_STEP
#define HSRLE_DEC8_STEP HSRLE_DECODE_STEP
#endif
#ifndef HSRLE_DEC8_RING
#define HSRLE_DEC8_RING HSRLE_DECODE_RING
#endif
constexpr int kDec8Tile = HSRLE_DEC8_TILE, kDec8Step = HSRLE_DEC8_STEP, kDec8Ring = HSRLE_DEC8_RING;
#ifdef HSRLE_DEC8_PE   // experiment build: parse + expand (csrc/experiments/hsrle_decode8pe.hip.h) for plain containers of the two list-free multi-symbol codecs
static hipError_t dec_plain(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PLAIN, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, false>, a, st) : launch_decode(k_decode8_pe<PLAIN>, a, st); }
static hipError_t dec_packed(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PACKED, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, false>, a, st) : launch_decode(k_decode8_pe<PACKED>, a, st); }
#else
static hipError_t dec_plain(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<PLAIN, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, false>, k_decode_blocks<PLAIN, 1, 0, kDec8Tile, 64, kDec8Step, false>, a, st); }
static hipError_t dec_packed(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<PACKED, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, false>, k_decode_blocks<PACKED, 1, 0, kDec8Tile, 64, kDec8Step, false>, a, st); }
#endif
static hipError_t dec_plain_any(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<PLAIN, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, true>, k_decode_blocks<PLAIN, 1, 0, kDec8Tile, 64, kDec8Step, true>, a, st); }
static hipError_t dec_packed_any(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<PACKED, 1, 0, kDec8Tile, kDec8Ring, kDec8Step, true>, k_decode_blocks<PACKED, 1, 0, kDec8Tile, 64, kDec8Step, true>, a, st); }
static hipError_t dec_lut3(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<LUT3, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<LUT3, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }
static hipError_t dec_lut7(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<LUT7, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<LUT7, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }

static hipError_t dec_short0(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<SHORT0, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<SHORT0, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }
static hipError_t dec_short1(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<SHORT1, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<SHORT1, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }
static hipError_t dec_short3(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<SHORT3, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<SHORT3, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }
static hipError_t dec_short7(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<SHORT7, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<SHORT7, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }

static hipError_t dec_short_single(const DecodeArgs &a, hipStream_t st) { return launch_decode_ring(k_decode_blocks<SHORT_SINGLE, 1, 0, kDecodeTile, kDecodeRing, kDecodeStep>, k_decode_blocks<SHORT_SINGLE, 1, 0, kDecodeTile, 64, kDecodeStep>, a, st); }

// rle8_multi / rle8_packed_multi / rle8_3symlut / rle8_7symlut: the ring encoder (one lane per block) for containers that fill the device with it, the run list encoder
// (hsrle_encode8r.hip.h: the whole wave per block) for smaller ones of 1 .. 4 KiB blocks.  Experiment builds: HSRLE_RUNLIST=1 / 2 = always / never.
template <int FAM>
static hipError_t enc_multi8(const EncodeArgs &a, hipStream_t st)
{
  // (round 6: plain / Packed / Short without and with a one-symbol list are position-parallel for every block the run list encoder took)
  if constexpr (FAM == SHORT3 || FAM == SHORT7)
  {
    static const uint32_t force = knob_u32("HSRLE_RUNLIST", 0u);
    if (run_list_applies(a.nBlocks, a.B, a.U, force) && a.residentWorkgroups == nullptr)
      return launch_run_list(k_encode8_runlist<FAM>, a, st);
  }
  return launch_encode_ring<1>(k_encode8_blocks<FAM, false, 256>, k_encode8_blocks<FAM, false, 128>, a, st);
}
static hipError_t enc_plain(const EncodeArgs &a, hipStream_t st) { return enc_multi8<PLAIN>(a, st); }
static hipError_t enc_short0(const EncodeArgs &a, hipStream_t st) { return enc_multi8<SHORT0>(a, st); }
static hipError_t enc_short1(const EncodeArgs &a, hipStream_t st) { return enc_multi8<SHORT1>(a, st); }
static hipError_t enc_short3(const EncodeArgs &a, hipStream_t st) { return enc_multi8<SHORT3>(a, st); }
static hipError_t enc_short7(const EncodeArgs &a, hipStream_t st) { return enc_multi8<SHORT7>(a, st); }
static hipError_t enc_packed(const EncodeArgs &a, hipStream_t st) { return enc_multi8<PACKED>(a, st); }
static hipError_t enc_lut3(const EncodeArgs &a, hipStream_t st) { return enc_multi8<LUT3>(a, st); }
static hipError_t enc_lut7(const EncodeArgs &a, hipStream_t st) { return enc_multi8<LUT7>(a, st); }
// Single: symbol pick by one wave per block, then the ring encoder (hsrle_encode8s.hip.h).  Blocks above kSinglePickMaxBlock (the one-lane
// drop-in path spans the whole input with one block) and HSRLE_SINGLE_V1=1 (A/B runs) use the first-generation kernel.
template <int MODE>   // 0 rle8_single, 1 rle8_packed_single, 2 rle8_single_short
static hipError_t enc_single_any(const EncodeArgs &a, hipStream_t st)
{
  static const bool v1 = knob_u32("HSRLE_SINGLE_V1", 0u) != 0u;
  if (v1 || a.B > kSinglePickMaxBlock)
  {
    if constexpr (MODE == 2) return launch_encode(k_encode_single_short_blocks<SHORT_SINGLE>, a, st, 0);
    else return launch_encode(k_encode_blocks<MODE == 1 ? PACKED_SINGLE : SINGLE, 1, 0>, a, st, 0);   // no residency cap: +40 % on run data, +20 % on noise, -8 % on video-shaped
  }
  if (a.residentWorkgroups == nullptr)
  {
    const uint32_t padded = (a.B + 63u) & ~63u;
    const uint32_t lds = 1024u + padded + 64u + (padded / 64u + 1u) * 8u;
    hipLaunchKernelGGL(k_single_pick, dim3(a.nBlocks), dim3(64), lds, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, MODE == 2 ? 8u : 9u, (uint64_t *)nullptr, (uint64_t *)nullptr,
                       (uint32_t *)nullptr, 0u, 0u);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
  }
  return launch_encode(k_encode8_single_blocks<MODE>, a, st, 0);
}
static hipError_t enc_single(const EncodeArgs &a, hipStream_t st) { return enc_single_any<0>(a, st); }
static hipError_t enc_packed_single(const EncodeArgs &a, hipStream_t st) { return enc_single_any<1>(a, st); }
static hipError_t enc_short_single(const EncodeArgs &a, hipStream_t st) { return enc_single_any<2>(a, st); }

static hipError_t idx_plain(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PLAIN, 1, 0>(a, records, st); }
static hipError_t idx_packed(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PACKED, 1, 0>(a, records, st); }
static hipError_t idx_lut3(const IndexArgs &a, int records, hipStream_t st) { return launch_index<LUT3, 1, 0>(a, records, st); }
static hipError_t idx_lut7(const IndexArgs &a, int records, hipStream_t st) { return launch_index<LUT7, 1, 0>(a, records, st); }
static hipError_t idx_short0(const IndexArgs &a, int records, hipStream_t st) { return launch_index<SHORT0, 1, 0>(a, records, st); }
static hipError_t idx_short1(const IndexArgs &a, int records, hipStream_t st) { return launch_index<SHORT1, 1, 0>(a, records, st); }
static hipError_t idx_short3(const IndexArgs &a, int records, hipStream_t st) { return launch_index<SHORT3, 1, 0>(a, records, st); }
static hipError_t idx_short7(const IndexArgs &a, int records, hipStream_t st) { return launch_index<SHORT7, 1, 0>(a, records, st); }
static hipError_t idx_short_single(const IndexArgs &a, int records, hipStream_t st) { return launch_index<SHORT_SINGLE, 1, 0>(a, records, st); }

static hipError_t sub_plain(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PLAIN, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_packed(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PACKED, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_plain_any(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PLAIN, 1, 0>(a, SB, 1u, rec, st); }
static hipError_t sub_packed_any(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PACKED, 1, 0>(a, SB, 1u, rec, st); }
static hipError_t sub_lut3(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<LUT3, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_lut7(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<LUT7, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_short0(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<SHORT0, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_short1(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<SHORT1, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_short3(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<SHORT3, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_short7(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<SHORT7, 1, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_short_single(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<SHORT_SINGLE, 1, 0>(a, SB, 0u, rec, st); }

static hipError_t menc_plain(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<PLAIN, true>, a, m, st); }
static hipError_t menc_packed(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<PACKED, true>, a, m, st); }
static hipError_t menc_short0(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<SHORT0, true>, a, m, st); }
static hipError_t menc_lut3(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<LUT3, true>, a, m, st); }
static hipError_t menc_lut7(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<LUT7, true>, a, m, st); }
// rle8_single_short as chunks of one monolithic stream (round 4; hsrle_encode_greedy.hip.h)
static hipError_t menc_single_short(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st)
{
  if (m.phase == 1u)
  {
    // split encode of a container: the blocks' symbols (a byte per block) and their cuts, as menc_single_any
    if (a.B > kSinglePickMaxBlock) return hipErrorInvalidValue;
    const uint32_t padded = (a.B + 63u) & ~63u;
    const uint32_t lds = 1024u + padded + 64u + (padded / 64u + 1u) * 8u;
    hipLaunchKernelGGL(k_single_pick, dim3(a.nBlocks), dim3(64), lds, st, a.in, a.U, a.B, a.nBlocks, (uint8_t *)const_cast<uint32_t *>(m.pick), 1u, 0u, m.cutPos, m.cutSym, m.cutFlags,
                       m.cutG, m.cutLong);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_encode_single_short_chunks<SHORT_SINGLE>), dim3((a.nBlocks + 63u) / 64u), dim3(64), 0, st, a.in, a.U, a.nBlocks, m.starts, m.slotOff, a.slots, a.sizes, m.pick,
                     m.jobs, m.jobCount, m.jobCap, a.B, (const uint32_t *)a.ringSel);
  if (m.jobs != nullptr)
    hipLaunchKernelGGL((k_copy_jobs<0>), dim3(2048), dim3(256), 0, st, a.in, a.slots, (const uint64_t *)m.jobs, (const uint32_t *)m.jobCount, m.jobCap);
  return hipGetLastError();
}
// 8 bit Single as chunks of one monolithic stream: the first-generation scanner, one lane per chunk (hsrle_encode.hip.h)
template <bool PACKEDSINGLE>
static hipError_t menc_single_any(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st)
{
  if (m.phase == 1u)
  {
    // split encode of a container: the blocks' symbols first (a byte per block; a.nBlocks = blocks here), the cut finder needs them
    if (a.B > kSinglePickMaxBlock) return hipErrorInvalidValue;
    const uint32_t padded = (a.B + 63u) & ~63u;
    const uint32_t lds = 1024u + padded + 64u + (padded / 64u + 1u) * 8u;
    hipLaunchKernelGGL(k_single_pick, dim3(a.nBlocks), dim3(64), lds, st, a.in, a.U, a.B, a.nBlocks, (uint8_t *)const_cast<uint32_t *>(m.pick), 1u, 0u, m.cutPos, m.cutSym, m.cutFlags,
                       m.cutG, m.cutLong);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_encode_single_chunks<PACKEDSINGLE>), dim3((a.nBlocks + 63u) / 64u), dim3(64), 0, st, a.in, a.U, a.nBlocks, m.starts, m.slotOff, a.slots, a.sizes, m.pick,
                     m.jobs, m.jobCount, m.jobCap, a.B, (const uint32_t *)a.ringSel);
  if (m.jobs != nullptr)
    hipLaunchKernelGGL((k_copy_jobs<0>), dim3(2048), dim3(256), 0, st, a.in, a.slots, (const uint64_t *)m.jobs, (const uint32_t *)m.jobCount, m.jobCap);
  return hipGetLastError();
}
static hipError_t menc_short1(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<SHORT1, true>, a, m, st); }
static hipError_t menc_short3(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<SHORT3, true>, a, m, st); }
static hipError_t menc_short7(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st) { return launch_mono_encode(k_encode8_blocks<SHORT7, true>, a, m, st); }

#ifdef HSRLE_EXPERIMENTS
static hipError_t wenc_plain(const WaveEncodeArgs &a, hipStream_t st) { return launch_wave_encode(k_encode8_wave<PLAIN>, a, st); }
static hipError_t wenc_packed(const WaveEncodeArgs &a, hipStream_t st) { return launch_wave_encode(k_encode8_wave<PACKED>, a, st); }
#endif

void register_w8(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc, WaveEncodeLaunch *wenc)
{
#ifdef HSRLE_EXPERIMENTS
  wenc[0] = wenc_plain; wenc[1] = wenc_packed;
#else
  (void)wenc;
#endif
  menc[4] = menc_single_any<false>; menc[5] = menc_single_any<true>; menc[kSingleShort] = menc_single_short;
  menc[0] = menc_plain; menc[1] = menc_packed; menc[kShortBase8 + 0] = menc_short0;
  menc[2] = menc_lut3; menc[3] = menc_lut7; menc[kShortBase8 + 1] = menc_short1; menc[kShortBase8 + 2] = menc_short3; menc[kShortBase8 + 3] = menc_short7;
  sub[0] = sub_plain; sub[1] = sub_packed; sub[2] = sub_lut3; sub[3] = sub_lut7; sub[4] = sub_plain_any; sub[5] = sub_packed_any;
  sub[kShortBase8 + 0] = sub_short0; sub[kShortBase8 + 1] = sub_short1; sub[kShortBase8 + 2] = sub_short3; sub[kShortBase8 + 3] = sub_short7;
  sub[kSingleShort] = sub_short_single;
  idx[0] = idx_plain; idx[1] = idx_packed; idx[2] = idx_lut3; idx[3] = idx_lut7; idx[4] = idx_plain; idx[5] = idx_packed;
  idx[kShortBase8 + 0] = idx_short0; idx[kShortBase8 + 1] = idx_short1; idx[kShortBase8 + 2] = idx_short3; idx[kShortBase8 + 3] = idx_short7;
  idx[kSingleShort] = idx_short_single;
  dec[0] = dec_plain;  enc[0] = enc_plain;
  dec[1] = dec_packed; enc[1] = enc_packed;
  dec[2] = dec_lut3;   enc[2] = enc_lut3;
  dec[3] = dec_lut7;   enc[3] = enc_lut7;
  dec[4] = dec_plain_any;  enc[4] = enc_single;
  dec[5] = dec_packed_any; enc[5] = enc_packed_single;
  // Short family (rle8_multi_short, rle8_{1,3,7}symlut_short; reference: src/rle.h:202-222)
  dec[kShortBase8 + 0] = dec_short0; enc[kShortBase8 + 0] = enc_short0;
  dec[kShortBase8 + 1] = dec_short1; enc[kShortBase8 + 1] = enc_short1;
  dec[kShortBase8 + 2] = dec_short3; enc[kShortBase8 + 2] = enc_short3;
  dec[kShortBase8 + 3] = dec_short7; enc[kShortBase8 + 3] = enc_short7;
  dec[kSingleShort] = dec_short_single; enc[kSingleShort] = enc_short_single;   // rle8_single_short (src/rle.h:223-224)
}

} // namespace hsrle

#ifdef HSRLE_ENC_STAMPS
extern "C" int hsrle_debug_enc_stamps(unsigned long long *out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hsrle::g_enc_stamps), sizeof(hsrle::g_enc_stamps)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(hsrle::g_enc_stamps), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
