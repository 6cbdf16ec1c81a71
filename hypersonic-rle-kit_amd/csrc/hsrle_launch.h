// hsrle_launch.h -- host-side launch table: codec id -> kernel launcher.  The kernels are instantiated per symbol
// width in inst_w*.hip (one translation unit per width so the library builds in parallel).
#pragma once
#include <atomic>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "hsrle_ring_probe.hip.h"

namespace hsrle {

// Developer knobs (A/B runs, forcing a kernel variant in the parity suite) exist only in -DHSRLE_EXPERIMENTS builds (tools/build_variant.sh
// exp -DHSRLE_EXPERIMENTS): the shipped library reads no environment variable in any launch path and holds one kernel per codec and case.
#ifdef HSRLE_EXPERIMENTS
inline uint32_t knob_u32(const char *name, uint32_t dflt) { const char *e = getenv(name); return (e && *e) ? (uint32_t)strtoul(e, nullptr, 10) : dflt; }
constexpr bool kExperiments = true;
#else
constexpr uint32_t knob_u32(const char *, uint32_t dflt) { return dflt; }
constexpr bool kExperiments = false;
#endif

struct DecodeArgs
{
  const uint8_t *payload;
  const uint64_t *offsets;
  const uint8_t *payloadEnd; // payload + payloadSize + tail pad: last readable byte + 1
  uint8_t *out;
  uint64_t U;
  uint32_t B, firstBlock, blockCount;
  uint32_t *status;
  int *residentWorkgroups = nullptr; // query mode: no launch; receives the number of workgroups (= waves) of this kernel that fit on one CU
  const uint32_t *entries = nullptr; // lane b starts from entry record b - entryBase (hsrle_index.hip.h) instead of from a block stream header
  uint32_t entryBase = 0;
  const uint32_t *gate = nullptr;    // decode enqueued BEFORE the host has seen the index passes' verdict: gate[0] | gate[1] != 0 (regions to repair / malformed) -> the kernel returns at once
};

// index passes over one monolithic stream (hsrle_index.hip.h)
struct IndexArgs
{
  const uint8_t *stream;        // device pointer to the stream's first byte (>= 64 readable bytes in front, >= 64 behind)
  uint32_t C, p0, G, M, R;      // stream bytes, first packet, region size, look-back of the guess, regions
  uint32_t single, singleSym;   // 8 bit Single mode / rle8_single_short: mode flag and the stream's one symbol
  const uint32_t *list; uint32_t listCount; const uint32_t *fix;   // repair pass: regions to walk again and their true entries
  uint32_t *mark; uint32_t roundTag; uint32_t extMax;                           //              mark[r] == roundTag: region r is on this round's list
  uint32_t *g, *e; uint64_t *olen; uint32_t *t;                       // walk results per region
  const uint32_t *entry; const uint64_t *outStart; const uint32_t *stateIn; uint64_t U; uint32_t B; uint32_t *rec;   // record pass
  const uint32_t *gate = nullptr;   // record pass enqueued BEFORE the host has seen the resolve verdict: gate[0] | gate[1] != 0 (regions to repair / malformed) -> writes nothing
};
typedef hipError_t (*IndexLaunch)(const IndexArgs &, int records, hipStream_t);
// container blocks cut into sub-blocks of SB output bytes: entry records for [firstBlock, firstBlock + blockCount) of `a` into rec
typedef hipError_t (*SubBlockLaunch)(const DecodeArgs &a, uint32_t SB, uint32_t *rec, hipStream_t);

struct EncodeArgs
{
  const uint8_t *in;
  uint64_t U;
  uint32_t B, nBlocks;
  uint8_t *slots;
  uint32_t slotStride;
  uint32_t *sizes;
  int *residentWorkgroups = nullptr; // query mode, as in DecodeArgs
  uint32_t *ringSel = nullptr;       // 16 bytes of device scratch: the encoders of 1 / 2 byte symbols choose their history ring per input (hsrle_encode8.hip.h)
};

// chunks of ONE monolithic stream (hsrle_mono_encode.hip.h): EncodeArgs with nBlocks = chunks, plus the chunk table
struct MonoEncodeArgs
{
  const uint64_t *starts; uint64_t *syms; const uint64_t *slotOff; uint32_t steps;   // (syms: rewritten by k_encodeS_blocks when it settles the lists itself)
  uint64_t *listOut = nullptr; uint32_t dry = 0;     // codecs with a move-to-front list: syms / listOut hold 8 words per chunk
  const uint32_t *pick = nullptr;                    // 8 bit Single: the stream's symbol (device); split encode of a container: a BYTE per block
  uint32_t phase = 0;                                // split encode, Single codecs: 1 = only pick the blocks' symbols into `pick` -- and find the blocks' cuts:
  uint64_t *cutPos = nullptr, *cutSym = nullptr; uint32_t *cutFlags = nullptr; uint32_t cutG = 0, cutLong = 0;   // (phase 1) per piece of cutG bytes, as k_mono_cuts8 leaves them
  uint64_t *jobs = nullptr; uint32_t *jobCount = nullptr; uint32_t jobCap = 0;   // 8 bit Single: literal stretches noted for k_copy_jobs
};
typedef hipError_t (*MonoEncodeLaunch)(const EncodeArgs &, const MonoEncodeArgs &, hipStream_t);
// wave-per-block encoder (hsrle_encode8w.hip.h): writes offsets and payload of the container directly
struct WaveEncodeArgs
{
  const uint8_t *in; uint64_t U; uint32_t B, nBlocks;
  uint64_t *offsets; uint8_t *payload;
  unsigned long long *tiles; uint32_t *ticket;      // zeroed by the caller: one word per block, one counter
  int *residentWorkgroups = nullptr;                // query mode
};
typedef hipError_t (*WaveEncodeLaunch)(const WaveEncodeArgs &, hipStream_t);
// position-parallel encoder (hsrle_encode8p.hip.h): phase 0 = sizes[] and the blocks' records, phase 1 = -- behind the caller's scan -- the streams at
// payload + offsets[b].  scratch: pp_scratch_bytes(nBlocks), 256-byte aligned; it needs no initialisation
struct PpArgs
{
  const uint8_t *in; uint64_t U; uint32_t B, nBlocks;
  uint32_t *sizes; const uint64_t *offsets; uint8_t *payload; uint8_t *scratch;
};
inline uint32_t pp_record_stride(uint32_t B) { return B / 4u < kPpRecords ? B / 4u : kPpRecords; }      // (4 * stride + 4 bytes per block: less than a staging slot of B + 193)
inline uint64_t pp_scratch_bytes(uint64_t nBlocks, uint32_t B) { return (4ull * pp_record_stride(B) + 4ull) * nBlocks + 256ull; }
typedef hipError_t (*PpLaunch)(const PpArgs &, int phase, hipStream_t);
// the same encoder for units of any length, window by window (hsrle_encode8pw.hip.h): phase 0 = one wave per unit (sizes[], window states, records),
// phase 1 = one wave per window (the packets of the runs that end in it)
constexpr uint32_t kPpwWindow = kPpMaxBlock;                    // bytes per window
constexpr uint32_t kPpwStateWords = 8u;                          // per window: posW, lastRLE, openStart, lastSymbol, stored runs, unit, window in unit, -
constexpr uint32_t kPpwEmpty = 0xFFFFFFFEu;                      // state word 4: no such window
constexpr uint32_t kPpwStride = kPpRecords;                      // records per window

struct PpwArgs
{
  const uint8_t *in; uint64_t U;
  uint32_t B;                     // != 0: the units are the blocks of a container (header + terminator each); 0: the chunks of ONE stream (no header, the last one ends it)
  uint32_t nUnits;                // blocks / an upper bound of the chunks
  const uint64_t *starts;         // chunks: [count + 1] input positions
  const uint64_t *syms;           // chunks: lastSymbol in front of each
  const uint32_t *count;          // chunks: how many there are (device)
  uint32_t *sizes; const uint64_t *offsets; uint8_t *payload;
  uint32_t *states;               // [nWindows][kPpwStateWords]
  uint32_t *recs;                 // [nWindows][kPpwStride]
  uint32_t nWindows;
};
inline uint64_t ppw_scratch_bytes(uint64_t nWindows) { return nWindows * (4ull * kPpwStateWords + 4ull * kPpwStride) + 512ull; }

typedef hipError_t (*PpwLaunch)(const PpwArgs &, int phase, hipStream_t);
constexpr uint32_t kPpwLStateWords = 32u;                        // ... of the general LUT kernel's (hsrle_encodeLpw.hip.h): the list travels in the state record
constexpr uint32_t kPpwSStateWords = 16u;                        // ... of the 1 .. 8 byte symbol codecs' windowed encoder (hsrle_encodeSpw.hip.h): more state crosses a window's edge
typedef hipError_t (*DecodeLaunch)(const DecodeArgs &, hipStream_t);
typedef hipError_t (*EncodeLaunch)(const EncodeArgs &, hipStream_t);

constexpr int kSingleShort = 109;                 // rle8_single_short
constexpr int kCodecCount = 110;                 // 50 extreme codecs + 44 of the Short family + 15 Greedy encoders (8f-1)
constexpr int kGreedyBase = 94;                   // + 3 * index(W in 16,24,32,48,64) + {0 1symlut, 1 3symlut, 2 7symlut}: rle{W}_{K}symlut_byte_short_compress_greedy
constexpr int kShortBase8 = 50;                   // rle8_multi_short, rle8_{1,3,7}symlut_short
constexpr int kShortBaseW = 54;                   // + 8 * index(W in 16,24,32,48,64) + {0 sym, 1 1symlut_sym, 2 3symlut_sym, 3 7symlut_sym, 4 byte, 5 1symlut_byte, 6 3symlut_byte, 7 7symlut_byte}
constexpr uint32_t kEncodeLdsCap = 20000;       // bytes of dynamic LDS per encode workgroup (0 = no residency cap); tuned on MI355X
#ifndef HSRLE_DECODE_TILE
#define HSRLE_DECODE_TILE 128
#endif
#ifndef HSRLE_DECODE_RING
#define HSRLE_DECODE_RING 128
#endif
#ifndef HSRLE_DECODE_STEP
#define HSRLE_DECODE_STEP 128
#endif
constexpr int kDecodeStep = HSRLE_DECODE_STEP; // output bytes per lane and decode/top-up step (k_decode_blocks Q)
constexpr int kDecodeTile = HSRLE_DECODE_TILE; // bytes produced per lane and round (k_decode_blocks T)
constexpr int kDecodeRing = HSRLE_DECODE_RING; // per-lane stream ring in LDS (k_decode_blocks R)

void register_w8(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc, WaveEncodeLaunch *wenc);
void register_pp8(PpLaunch *pp);
void register_pp8w(PpwLaunch *ppw);   // [0] rle8_multi, [1] rle8_packed_multi
void register_ppLw(PpwLaunch *ppw);   // [codec id]: the codecs of hsrle_encodeLp.hip.h, blocks above 4 KiB
void register_ppSw(PpwLaunch *ppw);   // [codec id]: the codecs of hsrle_encodeSp.hip.h, blocks above 4 KiB
void register_pp8s(PpLaunch *pp);
void register_pp128(PpLaunch *pp);
void register_ppL(PpLaunch *pp);
void register_ppS(PpLaunch *pp);
void register_w16(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);
void register_w24(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);
void register_w32(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);
void register_w48(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);
void register_w64(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);
void register_w128(DecodeLaunch *dec, EncodeLaunch *enc, IndexLaunch *idx, SubBlockLaunch *sub, MonoEncodeLaunch *menc);

template <typename... A>
constexpr int kernel_arity(void (*)(A...)) { return (int)sizeof...(A); }

template <typename KERNEL>
inline hipError_t launch_decode(KERNEL k, const DecodeArgs &a, hipStream_t st)
{
  if (a.residentWorkgroups != nullptr)
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(a.residentWorkgroups, k, 64, 0);
  const uint32_t grid = (a.blockCount + 63u) / 64u;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, st, a.payload, a.offsets, a.payloadEnd, a.out, a.U, a.B, a.firstBlock, a.blockCount, a.status, a.entries, a.entryBase, a.gate);
  return hipGetLastError();
}

// Decoders: containers whose streams shrink to less than a quarter (1 / 2 byte symbols) or a fifth take the instantiation with a 64-byte
// stream ring (12 instead of 9 waves per CU; a lane rarely needs more than 64 stream bytes for 128 output bytes there, and one that
// does finishes in the second pass).  8 GiB video-shaped rle8_packed (ratio 0.18): 3.90 -> 3.34 ms; at ratio 0.43 it would lose 9 %,
// on run-distributed data 9 - 25 %.  Blocks decoded from entry records (monolithic streams, split decode) keep the 128-byte ring.
// HSRLE_DEC_RING=64 / 128 in the environment forces one (tests, A/B).
// (PER_MILLE: the ratio below which the small ring is taken.  1 / 2 byte symbols: 250 -- rle16_sym at 0.23 still gains 6 %; wider symbols:
//  3 / 4 byte symbols: 215 (round 4; 200 before: rle32_3symlut_sym video-shaped at 0.2042 gains 9 % with the small ring) -- 8 GiB video-shaped rle32_3symlut_byte (0.17) +16 %, rle24_7symlut_byte_short (0.19) +12 %, every 24 / 32 bit
//  row of the sweep below 0.2 gains 10 - 20 %, but rle24_sym (0.26) -3 %, rle32_sym (0.30) -8 %.  6 / 8 byte symbols: never)
template <int PER_MILLE = 250, typename K128, typename K64>
inline hipError_t launch_decode_ring(K128 k128, K64 k64, const DecodeArgs &a, hipStream_t st)
{
  static const int forced = (int)knob_u32("HSRLE_DEC_RING", 0);
  const uint64_t payloadBytes = (uint64_t)(a.payloadEnd - a.payload);
  const bool small = forced ? forced == 64 : (a.entries == nullptr && a.residentWorkgroups == nullptr && payloadBytes * 1000u < a.U * (uint64_t)PER_MILLE);
  if (small && a.entries == nullptr) return launch_decode(k64, a, st);
  return launch_decode(k128, a, st);
}

// rle8_multi / rle8_packed_multi: which containers the run list encoder takes (hsrle_encode8r.hip.h; force: 1 always, 2 never -- experiment builds)
inline bool run_list_applies(uint64_t nBlocks, uint32_t B, uint64_t U, uint32_t force = 0u)
{
  if (B < 1024u || B > 4096u || U < 1024u || force == 2u) return false;
  return force == 1u || nBlocks < 131072u;
}

// blocks per wave: all of a small container's waves resident at once.  How many waves of a kernel a CU holds is asked once per kernel
// (LDS and registers decide: 7 .. 10 for the run list encoders) and remembered in a small table; racing first calls write the same value.
inline uint32_t run_list_waves_per_cu(const void *kernel, hipStream_t st)
{
  struct Entry { std::atomic<const void *> k; std::atomic<int> v; };
  static Entry table[256];
  for (auto &e : table)
  {
    const void *have = e.k.load(std::memory_order_acquire);
    if (have == kernel) { const int v = e.v.load(std::memory_order_relaxed); if (v > 0) return (uint32_t)v; break; }
    if (have == nullptr) break;
  }
  // (not hipOccupancyMaxActiveBlocksPerMultiprocessor: it divides the LDS without the 1 280-byte allocation granule and says 11 or 12 where the
  //  hardware holds 10 -- and a grid just above what is resident leaves a tail of waves that run alone: 88 MB frame 215 us instead of 161)
  int n = 8;
  // (a first call that is being captured into a graph asks nothing: the query is not a stream operation, and what may and may not be called
  //  under capture is the runtime's business -- the default serves, the next eager call fills the table)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return (uint32_t)n;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kernel) == hipSuccess)
  {
    const uint32_t lds = (((uint32_t)fa.sharedSizeBytes + 1279u) / 1280u) * 1280u, regs = (((uint32_t)fa.numRegs + 7u) / 8u) * 8u;
    const uint32_t byLds = lds ? 163840u / lds : 32u, byRegs = regs ? 4u * (512u / regs > 8u ? 8u : 512u / regs) : 32u;
    n = (int)(byLds < byRegs ? byLds : byRegs);
    if (n < 1) n = 1;
    if (knob_u32("HSRLE_RL_DEBUG", 0u)) fprintf(stderr, "run list kernel %p: lds %zu regs %d -> %d waves per CU\n", kernel, (size_t)fa.sharedSizeBytes, fa.numRegs, n);
  }
  for (auto &e : table)
  {
    const void *expected = nullptr;
    if (e.k.load(std::memory_order_acquire) == kernel || e.k.compare_exchange_strong(expected, kernel)) { e.v.store(n, std::memory_order_relaxed); break; }
  }
  return (uint32_t)n;
}

template <typename KERNEL>
inline hipError_t launch_run_list(KERNEL k, const EncodeArgs &a, hipStream_t st)
{
  const uint32_t resident = 256u * run_list_waves_per_cu((const void *)k, st);
  const uint32_t want = (a.nBlocks + resident - 1u) / resident;
  const uint32_t bpw = knob_u32("HSRLE_RL_BPW", want > 64u ? 64u : want);
  hipLaunchKernelGGL(k, dim3((a.nBlocks + bpw - 1u) / bpw), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes, bpw, (const uint32_t *)nullptr);
  return hipGetLastError();
}

template <typename KERNEL>
inline hipError_t launch_encode(KERNEL k, const EncodeArgs &a, hipStream_t st, int capResidency = 1)
{
  const uint32_t grid = (a.nBlocks + 63u) / 64u;
  // The encoder keeps one 128-byte line per lane open on the read side and one on the write side.  With the full 32 waves per CU
  // those open lines (32 CUs x 32 waves x 64 lanes x 2 x 128 B = 16 MiB per XCD) thrash the 4 MiB L2 and every line is fetched /
  // written several times (measured: 6.8x read, 5.9x write amplification).  A dynamic LDS reservation caps the residency.
  static const uint32_t ldsCap = knob_u32("HSRLE_ENCODE_LDS_CAP", kEncodeLdsCap);
  static const uint32_t lds8 = knob_u32("HSRLE_ENCODE8_LDS", 0u);
  if (a.residentWorkgroups != nullptr)
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(a.residentWorkgroups, k, 64, capResidency ? ldsCap : lds8);
  if constexpr (kernel_arity(KERNEL{}) == 14)   // kernels with a MONO mode (hsrle_encode8.hip.h): block mode = no chunk table
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), capResidency ? ldsCap : lds8, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes, (const uint64_t *)nullptr,
                       (uint64_t *)nullptr, (const uint64_t *)nullptr, 0u, (uint64_t *)nullptr, 0u, (const uint32_t *)nullptr);
  else
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), capResidency ? ldsCap : lds8, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes);
  return hipGetLastError();
}

// the encoders of 1 / 2 byte symbols: probe a sample of the input, then both instantiations -- the one that was not chosen returns at once.
// HSRLE_ENC_RING=128 / 256 in the environment forces one (tests, A/B).  Without scratch (a.ringSel == nullptr): the 256-byte ring.
template <int S, typename K256, typename K128>
inline hipError_t launch_encode_ring(K256 k256, K128 k128, const EncodeArgs &a, hipStream_t st)
{
  static const int forced = (int)knob_u32("HSRLE_ENC_RING", 0);
  // (below ~131 072 blocks the device is not full with 9 waves per CU either: more waves bring nothing, and the probe + the second launch
  //  are ~25 us of a call that short)
  if (a.residentWorkgroups != nullptr || forced == 256 || (forced == 0 && (a.ringSel == nullptr || a.nBlocks < 131072u)))
    return launch_encode(k256, a, st, 0);
  if (forced == 128)
    return launch_encode(k128, a, st, 0);
  if (zero_async(a.ringSel, 16, st) != hipSuccess) return hipErrorUnknown;   // (a kernel, not hipMemsetAsync: hsrle_common.hip.h zero_async)
  const uint32_t samples = a.nBlocks < 256u ? a.nBlocks : 256u;
  hipLaunchKernelGGL((k_ring_probe<S>), dim3(samples), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.ringSel);
  hipLaunchKernelGGL((k_ring_decide<S>), dim3(1), dim3(64), 0, st, a.ringSel);
  const uint32_t grid = (a.nBlocks + 63u) / 64u;
  hipLaunchKernelGGL(k256, dim3(grid), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint64_t *)nullptr, 0u,
                     (uint64_t *)nullptr, 0u, (const uint32_t *)a.ringSel);
  hipLaunchKernelGGL(k128, dim3(grid), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint64_t *)nullptr, 0u,
                     (uint64_t *)nullptr, 0u, (const uint32_t *)a.ringSel);
  return hipGetLastError();
}

#ifdef HSRLE_EXPERIMENTS
template <typename KERNEL>
inline hipError_t launch_wave_encode(KERNEL k, const WaveEncodeArgs &a, hipStream_t st)
{
  static int perCu = 0, cus = 0;
  if (perCu == 0)
  {
    int dev = 0, n = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, 0) != hipSuccess ||
        hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0 || c <= 0)
      return hipErrorInvalidValue;
    perCu = n; cus = c;
  }
  if (a.residentWorkgroups != nullptr) { *a.residentWorkgroups = perCu; return hipSuccess; }
  // resident waves take blocks in ticket order
  const uint64_t resident = (uint64_t)perCu * (uint64_t)cus;
  const uint32_t grid = (uint32_t)(a.nBlocks < resident ? a.nBlocks : resident);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.offsets, a.payload, a.tiles, a.ticket);
  return hipGetLastError();
}

#endif

template <typename KERNEL>
inline hipError_t launch_mono_encode(KERNEL k, const EncodeArgs &a, const MonoEncodeArgs &m, hipStream_t st)
{
  // (a.B != 0: chunks of a container's blocks -- a.ringSel[0] is then the number of chunks: hsrle_encode8.hip.h)
  hipLaunchKernelGGL(k, dim3((a.nBlocks + 63u) / 64u), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.slots, a.slotStride, a.sizes, m.starts, m.syms, m.slotOff, m.steps, m.listOut, m.dry, (const uint32_t *)a.ringSel);
  return hipGetLastError();
}

} // namespace hsrle
