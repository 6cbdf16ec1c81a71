// 128 bit symbols: rle128_{sym,byte}[_packed]  (reference: src/rle.h:124-125, :146-147, :168-169, :194-195)
#include "hsrle_decode.hip.h"
#include "hsrle_encode.hip.h"
#include "hsrle_encode128.hip.h"
#include "hsrle_index.hip.h"
#include "hsrle_launch.h"

namespace hsrle {

static hipError_t dec_sym(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PLAIN, 16, 1, kDecodeTile, kDecodeRing, kDecodeStep>, a, st) : launch_decode(k_decode_blocks<PLAIN, 16, 1, kDecodeTile, kDecodeRing, kDecodeStep, true, false>, a, st); }
static hipError_t dec_sym_packed(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PACKED, 16, 1, kDecodeTile, kDecodeRing, kDecodeStep>, a, st) : launch_decode(k_decode_blocks<PACKED, 16, 1, kDecodeTile, kDecodeRing, kDecodeStep, true, false>, a, st); }
static hipError_t dec_byte(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PLAIN, 16, 0, kDecodeTile, kDecodeRing, kDecodeStep>, a, st) : launch_decode(k_decode_blocks<PLAIN, 16, 0, kDecodeTile, kDecodeRing, kDecodeStep, true, false>, a, st); }
static hipError_t dec_byte_packed(const DecodeArgs &a, hipStream_t st) { return a.entries ? launch_decode(k_decode_blocks<PACKED, 16, 0, kDecodeTile, kDecodeRing, kDecodeStep>, a, st) : launch_decode(k_decode_blocks<PACKED, 16, 0, kDecodeTile, kDecodeRing, kDecodeStep, true, false>, a, st); }

// ring encoder (hsrle_encode128.hip.h) for block sizes up to 64 KiB; HSRLE_ENCODE128_V1=1 selects the first-generation kernel (A/B runs)
template <int FAM, int AL>
static hipError_t enc_any(const EncodeArgs &a, hipStream_t st)
{
  static const bool v1 = knob_u32("HSRLE_ENCODE128_V1", 0u) != 0u;
  if (v1 || a.B > 65536u) return launch_encode(k_encode_blocks<FAM, 16, AL>, a, st);   // (the one-block drop-in path spans the whole input with one block: one lane either way)
  return launch_encode(k_encode128_blocks<FAM == PACKED, AL>, a, st, 0);
}
static hipError_t enc_sym(const EncodeArgs &a, hipStream_t st) { return enc_any<PLAIN, 1>(a, st); }
static hipError_t enc_sym_packed(const EncodeArgs &a, hipStream_t st) { return enc_any<PACKED, 1>(a, st); }
static hipError_t enc_byte(const EncodeArgs &a, hipStream_t st) { return enc_any<PLAIN, 0>(a, st); }
static hipError_t enc_byte_packed(const EncodeArgs &a, hipStream_t st) { return enc_any<PACKED, 0>(a, st); }

static hipError_t idx_sym(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PLAIN, 16, 1>(a, records, st); }
static hipError_t idx_sym_packed(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PACKED, 16, 1>(a, records, st); }
static hipError_t idx_byte(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PLAIN, 16, 0>(a, records, st); }
static hipError_t idx_byte_packed(const IndexArgs &a, int records, hipStream_t st) { return launch_index<PACKED, 16, 0>(a, records, st); }

static hipError_t sub_sym(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PLAIN, 16, 1>(a, SB, 0u, rec, st); }
static hipError_t sub_sym_packed(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PACKED, 16, 1>(a, SB, 0u, rec, st); }
static hipError_t sub_byte(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PLAIN, 16, 0>(a, SB, 0u, rec, st); }
static hipError_t sub_byte_packed(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t st) { return launch_sub_or_wave<PACKED, 16, 0>(a, SB, 0u, rec, st); }

// chunks of one monolithic stream (hsrle_mono_encode.hip.h): the first-generation encoder, one lane per chunk, from its place behind a stored run
template <int FAM, int AL>
static hipError_t menc_any(const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st)
{
  hipLaunchKernelGGL((k_encode128_chunks<FAM, AL>), dim3((a.nBlocks + 63u) / 64u), dim3(64), 0, st, a.in, a.U, a.nBlocks, m.starts, m.syms, m.slotOff, a.slots, a.sizes, a.B, (const uint32_t *)a.ringSel);
  return hipGetLastError();
}

void register_w128(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc)
{
  menc[46] = menc_any<PLAIN, 1>; menc[47] = menc_any<PACKED, 1>; menc[48] = menc_any<PLAIN, 0>; menc[49] = menc_any<PACKED, 0>;
  sub[46] = sub_sym; sub[47] = sub_sym_packed; sub[48] = sub_byte; sub[49] = sub_byte_packed;
  idx[46] = idx_sym; idx[47] = idx_sym_packed; idx[48] = idx_byte; idx[49] = idx_byte_packed;
  dec[46] = dec_sym;         enc[46] = enc_sym;
  dec[47] = dec_sym_packed;  enc[47] = enc_sym_packed;
  dec[48] = dec_byte;        enc[48] = enc_byte;
  dec[49] = dec_byte_packed; enc[49] = enc_byte_packed;
}

} // namespace hsrle
