/*
 * hsrle.h -- C ABI of libhsrle_hip.so: the MI355X-native implementation of the hypersonic-rle-kit
 * `rleX_extreme` encode / decode hot path (8/16/24/32/48/64/128 bit symbols; plain, Packed, 3/7 symbol LUT
 * and Single variants).
 *
 * Three layers, all `extern "C"`, plain pointers and sizes:
 *
 *  1. DROP-IN entry points with exactly the names, signatures and error behaviour of the reference's public
 *     header (reference: src/rle.h:100-394; registered in src/codec_funcs.h:270-410).  Host pointers in, host
 *     pointers out, one monolithic reference stream.  A maintainer links libhsrle_hip.so instead of
 *     rle8_extreme_cpu.c / rleX_extreme_cpu.c / rle{24,48,128}_extreme_cpu.c / rleX_Xsl.c (INTEGRATION.md has the link recipe;
 *     the reference's own `hsrlekit` built that way passes its `--test` run on the GPU, tests/test_gpu_parity.py).
 *
 *  2. Device-resident block API (`hsrle_*_dev`): the input is cut into fixed-size blocks, every block is encoded
 *     as an independent, self-terminating reference stream (bit-exact with what the reference encoder produces
 *     for that block) and the streams are concatenated behind an offset table.  This is the throughput path; its
 *     layout follows the reference's own sub-section container for rle8m
 *     (reference: src/rle8_low_entropy_cpu.c:131-250).
 *
 *  3. Host convenience wrappers around 2 (`hsrle_compress_host`, `hsrle_decompress_host`).
 *
 *  4. Multi-GPU: one container from the containers of W ranks and back, over RCCL (section 4 below).
 *
 * Threads and streams.  The device-pointer functions (2, 4, hsrle_decompress_mono_dev) keep NO library-owned state between calls: they
 * work on the caller's buffers, and what they need beyond those (the compression workspace when dWorkspace is NULL, status words) is
 * allocated stream-ordered on the caller's stream for the duration of the call.  Any number of threads may call them at the same time,
 * on the same or on different streams and devices (the device is the calling thread's current HIP device).  The host-pointer drop-in
 * functions (1, 3) stage through per-device buffers and hold that device's lock for the whole call: they are safe to call from several
 * threads, and calls on one device run one after the other (the reference's own callers are single threaded: src/main.c, src/rle_fuzz.c).
 * Device state is created lazily on first use (like the reference's rle8m_opencl_decompress does, src/rle8_ocl.c:324).  The library never
 * falls back to a CPU codec: if no HIP device is usable every entry point fails (drop-in functions return 0, hsrle_* an error code).
 */
#ifndef HSRLE_H
#define HSRLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------- */
/* codec ids.  One per (compress, decompress) pair of the reference's table that belongs to the hot path.      */

typedef enum hsrle_codec
{
  /* ids 0 / 1 name the multi-symbol ENCODERS: the blocks of such a container are mode-0 streams (what those encoders write); a     */
  /* Single-mode block in it is reported as a format error.  The drop-in rle8_decompress / rle8_packed_decompress take either mode. */
  HSRLE_RLE8_MULTI = 0,          /* rle8_multi_compress / rle8_decompress                     rle.h:101,103 */
  HSRLE_RLE8_PACKED_MULTI = 1,   /* rle8_packed_multi_compress / rle8_packed_decompress       rle.h:173,175 */
  HSRLE_RLE8_3SYMLUT = 2,        /* rle8_3symlut_*                                            rle.h:199-200 */
  HSRLE_RLE8_7SYMLUT = 3,        /* rle8_7symlut_*                                            rle.h:207-208 */
  HSRLE_RLE8_SINGLE = 4,         /* rle8_single_compress / rle8_decompress                    rle.h:102,103 */
  HSRLE_RLE8_PACKED_SINGLE = 5,  /* rle8_packed_single_compress / rle8_packed_decompress      rle.h:174,175 */

  /* for W in 16,24,32,48,64:  base = 6 + 8 * index(W);  base + k with k = */
  /*   0 sym  1 sym_packed  2 3symlut_sym  3 7symlut_sym  4 byte  5 byte_packed  6 3symlut_byte  7 7symlut_byte */
  HSRLE_RLE16_SYM = 6, HSRLE_RLE16_SYM_PACKED, HSRLE_RLE16_3SYMLUT_SYM, HSRLE_RLE16_7SYMLUT_SYM,
  HSRLE_RLE16_BYTE, HSRLE_RLE16_BYTE_PACKED, HSRLE_RLE16_3SYMLUT_BYTE, HSRLE_RLE16_7SYMLUT_BYTE,
  HSRLE_RLE24_SYM = 14, HSRLE_RLE24_SYM_PACKED, HSRLE_RLE24_3SYMLUT_SYM, HSRLE_RLE24_7SYMLUT_SYM,
  HSRLE_RLE24_BYTE, HSRLE_RLE24_BYTE_PACKED, HSRLE_RLE24_3SYMLUT_BYTE, HSRLE_RLE24_7SYMLUT_BYTE,
  HSRLE_RLE32_SYM = 22, HSRLE_RLE32_SYM_PACKED, HSRLE_RLE32_3SYMLUT_SYM, HSRLE_RLE32_7SYMLUT_SYM,
  HSRLE_RLE32_BYTE, HSRLE_RLE32_BYTE_PACKED, HSRLE_RLE32_3SYMLUT_BYTE, HSRLE_RLE32_7SYMLUT_BYTE,
  HSRLE_RLE48_SYM = 30, HSRLE_RLE48_SYM_PACKED, HSRLE_RLE48_3SYMLUT_SYM, HSRLE_RLE48_7SYMLUT_SYM,
  HSRLE_RLE48_BYTE, HSRLE_RLE48_BYTE_PACKED, HSRLE_RLE48_3SYMLUT_BYTE, HSRLE_RLE48_7SYMLUT_BYTE,
  HSRLE_RLE64_SYM = 38, HSRLE_RLE64_SYM_PACKED, HSRLE_RLE64_3SYMLUT_SYM, HSRLE_RLE64_7SYMLUT_SYM,
  HSRLE_RLE64_BYTE, HSRLE_RLE64_BYTE_PACKED, HSRLE_RLE64_3SYMLUT_BYTE, HSRLE_RLE64_7SYMLUT_BYTE,

  HSRLE_RLE128_SYM = 46,         /* rle128_sym_*          rle.h:124-125 */
  HSRLE_RLE128_SYM_PACKED = 47,  /* rle128_sym_packed_*   rle.h:146-147 */
  HSRLE_RLE128_BYTE = 48,        /* rle128_byte_*         rle.h:168-169 */
  HSRLE_RLE128_BYTE_PACKED = 49, /* rle128_byte_packed_*  rle.h:194-195 */

  /* Short family: one-byte packed headers with a 0 / 1 / 3 / 7 symbol move-to-front list.                 */
  HSRLE_RLE8_MULTI_SHORT = 50,   /* rle8_multi_short_*     rle.h:221-222 */
  HSRLE_RLE8_1SYMLUT_SHORT = 51, /* rle8_1symlut_short_*   rle.h:216-217 */
  HSRLE_RLE8_3SYMLUT_SHORT = 52, /* rle8_3symlut_short_*   rle.h:202-203 */
  HSRLE_RLE8_7SYMLUT_SHORT = 53, /* rle8_7symlut_short_*   rle.h:210-211 */
  /* for W in 16,24,32,48,64:  base = 54 + 8 * index(W);  base + k with k =                                              */
  /*   0 sym_short  1 1symlut_sym_short  2 3symlut_sym_short  3 7symlut_sym_short                                        */
  /*   4 byte_short 5 1symlut_byte_short 6 3symlut_byte_short 7 7symlut_byte_short            rle.h:228-348              */
  HSRLE_RLE16_SYM_SHORT = 54, HSRLE_RLE24_SYM_SHORT = 62, HSRLE_RLE32_SYM_SHORT = 70, HSRLE_RLE48_SYM_SHORT = 78, HSRLE_RLE64_SYM_SHORT = 86,

  /* Greedy encoders (rle.h:398-416): base = 94 + 3 * index(W) + {0 1symlut, 1 3symlut, 2 7symlut}; compress =                */
  /* rle{W}_{K}symlut_byte_short_compress_greedy, decompress = rle{W}_{K}symlut_byte_short_decompress (codec_funcs.h:298-388) */
  HSRLE_RLE16_1SYMLUT_BYTE_SHORT_GREEDY = 94, HSRLE_RLE24_1SYMLUT_BYTE_SHORT_GREEDY = 97, HSRLE_RLE32_1SYMLUT_BYTE_SHORT_GREEDY = 100,
  HSRLE_RLE48_1SYMLUT_BYTE_SHORT_GREEDY = 103, HSRLE_RLE64_1SYMLUT_BYTE_SHORT_GREEDY = 106,

  HSRLE_RLE8_SINGLE_SHORT = 109, /* rle8_single_short_*    rle.h:223-224 */

  HSRLE_CODEC_COUNT = 110
} hsrle_codec_t;

typedef enum hsrle_status
{
  HSRLE_OK = 0,
  HSRLE_ERR_ARGUMENT = 1,    /* NULL pointer, zero size, unknown codec, bad block size                       */
  HSRLE_ERR_CAPACITY = 2,    /* output / workspace buffer too small                                          */
  HSRLE_ERR_FORMAT = 3,      /* container or block stream header inconsistent / stream malformed             */
  HSRLE_ERR_DEVICE = 4,      /* no usable HIP device, or a HIP runtime call failed                           */
  HSRLE_ERR_UNSUPPORTED = 5  /* kernel for this codec is not available in this build                         */
} hsrle_status_t;

/* name <-> id, using the reference's naming ("rle8_packed_multi", "rle64_3symlut_byte", ...).  -1 if unknown. */
int hsrle_codec_from_name(const char *name);
const char *hsrle_codec_name(int codec);
const char *hsrle_status_string(int status);

/* ---------------------------------------------------------------------------------------------------------- */
/* 1. drop-in entry points (reference: src/rle.h).  0 = failure.                                               */

uint32_t rle_compress_bounds(const uint32_t inSize);   /* rle.h:100, rle8_extreme_cpu.c:22-28 */
uint32_t rle_decompress_additional_size(void);         /* rle.h:105, rle8_extreme_cpu.c:17-20 */

#define HSRLE_DECL_PAIR(name) \
  uint32_t name##_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize); \
  uint32_t name##_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
#define HSRLE_DECL_WIDTH(W) \
  HSRLE_DECL_PAIR(rle##W##_sym) HSRLE_DECL_PAIR(rle##W##_sym_packed) HSRLE_DECL_PAIR(rle##W##_byte) HSRLE_DECL_PAIR(rle##W##_byte_packed) \
  HSRLE_DECL_PAIR(rle##W##_3symlut_sym) HSRLE_DECL_PAIR(rle##W##_7symlut_sym) HSRLE_DECL_PAIR(rle##W##_3symlut_byte) HSRLE_DECL_PAIR(rle##W##_7symlut_byte)

/* 8 bit (irregular names; rle.h:101-103, :173-175, :199-200, :207-208) */
uint32_t rle8_multi_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_single_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_packed_multi_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_packed_single_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_packed_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
HSRLE_DECL_PAIR(rle8_3symlut)
HSRLE_DECL_PAIR(rle8_7symlut)

/* 16 / 24 / 32 / 48 / 64 bit (rle.h:107-122, :129-144, :151-166, :177-192, :250-258, :284-292, :318-326, :352-360, :386-394) */
HSRLE_DECL_WIDTH(16)
HSRLE_DECL_WIDTH(24)
HSRLE_DECL_WIDTH(32)
HSRLE_DECL_WIDTH(48)
HSRLE_DECL_WIDTH(64)

/* 128 bit (rle.h:124-125, :146-147, :168-169, :194-195) */
HSRLE_DECL_PAIR(rle128_sym)
HSRLE_DECL_PAIR(rle128_sym_packed)
HSRLE_DECL_PAIR(rle128_byte)
HSRLE_DECL_PAIR(rle128_byte_packed)

/* Short family: rle.h:202-203, :210-211, :216-217, :221-222 (8 bit), :228-348 (16..64 bit) */
#define HSRLE_DECL_SHORT_WIDTH(W) \
  HSRLE_DECL_PAIR(rle##W##_sym_short) HSRLE_DECL_PAIR(rle##W##_byte_short) HSRLE_DECL_PAIR(rle##W##_1symlut_sym_short) HSRLE_DECL_PAIR(rle##W##_1symlut_byte_short) \
  HSRLE_DECL_PAIR(rle##W##_3symlut_sym_short) HSRLE_DECL_PAIR(rle##W##_3symlut_byte_short) HSRLE_DECL_PAIR(rle##W##_7symlut_sym_short) HSRLE_DECL_PAIR(rle##W##_7symlut_byte_short)
HSRLE_DECL_PAIR(rle8_multi_short)
HSRLE_DECL_PAIR(rle8_single_short)
HSRLE_DECL_PAIR(rle8_1symlut_short)
HSRLE_DECL_PAIR(rle8_3symlut_short)
HSRLE_DECL_PAIR(rle8_7symlut_short)
HSRLE_DECL_SHORT_WIDTH(16)
HSRLE_DECL_SHORT_WIDTH(24)
HSRLE_DECL_SHORT_WIDTH(32)
HSRLE_DECL_SHORT_WIDTH(48)
HSRLE_DECL_SHORT_WIDTH(64)

/* Greedy encoders of the byte-aligned LUT Short codecs: rle.h:398-416 */
#define HSRLE_DECL_GREEDY(W) \
  uint32_t rle##W##_1symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize); \
  uint32_t rle##W##_3symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize); \
  uint32_t rle##W##_7symlut_byte_short_compress_greedy(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
HSRLE_DECL_GREEDY(16)
HSRLE_DECL_GREEDY(24)
HSRLE_DECL_GREEDY(32)
HSRLE_DECL_GREEDY(48)
HSRLE_DECL_GREEDY(64)
#undef HSRLE_DECL_GREEDY

#undef HSRLE_DECL_SHORT_WIDTH
#undef HSRLE_DECL_WIDTH
#undef HSRLE_DECL_PAIR

/* Generic form of the above (what the macro-declared functions forward to): one monolithic reference stream,
 * host pointers.  Mirrors `codecCallbacks[codec].compress_func / .decompress_func` (codec_funcs.h:262-266). */
uint32_t hsrle_compress_mono(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);
uint32_t hsrle_decompress_mono(int codec, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);

/*
 * Device-resident form of <codec>_decompress: ONE monolithic reference stream (what the reference's encoders and `hsrlekit` write:
 * src/main.c:835, :970; decoded there by src/rle8_extreme_cpu.h:702-764, src/rleX_extreme_cpu_decode.h:27-164, src/rleX_Xsl.h:1848-1881)
 * in device memory, decoded into device memory.  A stream has no random access, so the library first builds an entry-point index
 * (speculative packet walks per stream region, then a pass that PROVES the chain and repairs wrong guesses: csrc/hsrle_index.hip.h)
 * and then runs the block kernel from those entry points.  dStream must be 128-byte aligned and readable up to streamSize + 64.
 * dWorkspace >= hsrle_decompress_mono_workspace_size() bytes at any address (the library rounds it up to 16 bytes itself; the size includes that slack).  Synchronises `stream` (once, at the end, when every entry guess holds).  pStats (optional, host, 4 values): stream regions,
 * repair rounds, regions walked again, final look-back of the entry guess.  Returns HSRLE_OK, HSRLE_ERR_FORMAT (malformed stream / sizes do not match the header), ...
 */
uint64_t hsrle_decompress_mono_workspace_size(int codec, uint32_t uncompressedSize, uint32_t compressedSize);
/*
 * Device-resident form of <codec>_compress for the multi-symbol codecs of 8 to 64 bit symbols (plain, Packed, 3 / 7 symbol LUT and
 * the Short family; not Single, 128 bit, Greedy): ONE monolithic reference stream, byte-identical with the reference encoder's
 * (src/rle8_extreme_cpu.h:86-344, :936-1099, src/rleX_extreme_cpu_encode.h, src/rleX_Xsl.h, src/rleX_Xsl_short.h), written by many
 * lanes -- the input is cut behind runs that every encoder state stores, and the pieces are encoded by the block kernels
 * (csrc/hsrle_mono_encode.hip.h).  The codecs with a move-to-front list get the list in front of every piece from a dry pass, and the
 * result is checked piece by piece (wrong guesses are encoded again); HSRLE_ERR_UNSUPPORTED if that does not settle, and for the other
 * codecs (the drop-in functions then use one lane).
 * Round 6: rle8_multi / rle8_packed_multi do not use the block kernels' chunk mode any more -- a wave per chunk, then a wave per 4 KiB window
 * (csrc/hsrle_encode8pw.hip.h), nothing read back before the end.
 * dOut capacity >= rle_compress_bounds(inSize); dWorkspace >= hsrle_compress_mono_workspace_size(); synchronises `stream`.
 * hsrle_mono_encode_stats: of the calling thread's last list-codec encode: extra rounds, pieces encoded again in rounds 1 and 2,
 * pieces the final check rejected (0 unless the library is wrong).
 */
uint64_t hsrle_compress_mono_workspace_size(int codec, uint32_t inSize);
int hsrle_compress_mono_dev(int codec, const void *dIn, uint32_t inSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *pStreamSize,
                            uint32_t *pChunks, void *stream);
void hsrle_mono_encode_stats(uint32_t stats[4]);
/*
 * rle8_multi / rle8_packed_multi (round 6: their chunks go to the windowed position-parallel encoder, csrc/hsrle_encode8pw.hip.h -- no chunk count, no step bound
 * and no list verdict has to come back to the host): the same encode without the host in the loop.  Nothing here synchronises or reads device memory, a HIP
 * graph can capture the call.  The stream's size is in its own header (bytes 4 .. 7, as the reference writes it: src/rle8_extreme_cpu.h:330-338) and, if
 * dStreamSize (DEVICE, 4 bytes) is not NULL, there.  HSRLE_ERR_UNSUPPORTED for every other codec (use hsrle_compress_mono_dev).
 */
int hsrle_compress_mono_dev_async(int codec, const void *dIn, uint32_t inSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *dStreamSize,
                                  void *stream);
int hsrle_decompress_mono_dev(int codec, const void *dStream, uint32_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize,
                              uint32_t *pUncompressedSize, uint32_t *pStats, void *stream);
/*
 * The same without the host in the loop (nothing here synchronises or reads device memory: a HIP graph can capture the call).  The walk,
 * the proof, the entry records and the decode are enqueued in one go; the records pass is gated ON THE DEVICE by the proof's verdict.
 * pHeader16: the stream's first 16 bytes in HOST memory (zeros behind a stream shorter than that; the smallest stream is the codec's header, as for the
 * synchronous entry) (the caller read or received the stream; the launch geometry comes from the two
 * sizes and the mode byte in it -- rle8_extreme_cpu.h:702-764).  dStatus (device, 4 bytes) receives HSRLE_MONO_DONE (dOut holds the
 * output), HSRLE_MONO_MALFORMED, or HSRLE_MONO_NEEDS_REPAIR: an entry guess of the walk did not hold on this stream -- dOut is
 * unspecified, call hsrle_decompress_mono_dev (which repairs; codecs whose junk walks do not die on random literals -- the non-Packed,
 * non-7-bit-range formats -- see this more often because the synchronous function's pilot pass is skipped here).
 */
#define HSRLE_MONO_DONE 0
#define HSRLE_MONO_MALFORMED 1
#define HSRLE_MONO_NEEDS_REPAIR 2
int hsrle_decompress_mono_dev_async(int codec, const void *dStream, const uint8_t *pHeader16, uint32_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace,
                                    uint64_t workspaceSize, uint32_t *pUncompressedSize, uint32_t *dStatus, void *stream);
/* tuning / test knob of the monolithic decode: output bytes per decode lane (multiple of 128), stream bytes per index lane, look-back
 * bytes of the entry guess; 0 = the library's choice.  Any values give the same output: the index is proven, not assumed.
 * PROCESS-GLOBAL and meant for tests: the one exception to "no library-owned state" -- a thread that changes it between another thread's
 * hsrle_*_mono_workspace_size() and its hsrle_*_mono_dev() can leave that call with too small a workspace (it then returns
 * HSRLE_ERR_CAPACITY, it never writes out of bounds). */
void hsrle_mono_tuning(uint32_t blockSize, uint32_t regionSize, uint32_t lookBack);

/* ---------------------------------------------------------------------------------------------------------- */
/* 2. device-resident block container API                                                                      */

/*
 * Container layout (little endian):
 *   [ 0] char     magic[8]  = "HSRLEKIT"
 *   [ 8] uint32_t version   = 1
 *   [12] uint32_t codec     (hsrle_codec_t)
 *   [16] uint64_t uncompressedSize
 *   [24] uint32_t blockSize (uncompressed bytes per block; the last block may be shorter)
 *   [28] uint32_t blockCount
 *   [32] uint64_t payloadSize (sum of the block streams)
 *   [40] uint64_t totalSize   (header + table + payload + 32 bytes of zero padding)
 *   [48] uint8_t  reserved[16]
 *   [64] uint64_t offset[blockCount + 1]   (relative to the payload start; offset[blockCount] == payloadSize)
 *   [64 + 8 * (blockCount + 1)] payload: block streams back to back, each a complete reference stream
 *   32 zero bytes (lets the decoder use 16-byte vector loads up to the last stream byte)
 */
#define HSRLE_CONTAINER_HEADER_SIZE 64u
#define HSRLE_CONTAINER_TAIL_PAD 32u
#define HSRLE_DEFAULT_BLOCK_SIZE 4096u
#define HSRLE_MIN_BLOCK_SIZE 128u
#define HSRLE_MAX_BLOCK_SIZE (1u << 30)

typedef struct hsrle_container_info
{
  uint32_t version, codec;
  uint64_t uncompressedSize;
  uint32_t blockSize, blockCount;
  uint64_t payloadSize, totalSize;
} hsrle_container_info_t;

/* A block size for `inSize` input bytes that fills the GPU: the default (4096) when that gives at least 65536 blocks, else 2048
 * or 1024.  One lane walks one block, so a buffer with few blocks is bound by the per-block latency (32 steps of ~8 us for 4 KiB),
 * not by bandwidth: an 88 MB frame decodes at 250 GiB/s with 4 KiB blocks and at 770 GiB/s with 1 KiB blocks -- for 9 % more
 * compressed bytes (every block carries its own stream header and terminator).  The choice is the caller's; nothing in the
 * library calls this on its own. */
uint32_t hsrle_suggest_block_size(uint64_t inSize);

/* Upper bound of the container size for `inSize` input bytes cut into blocks of `blockSize` (0 = default). */
uint64_t hsrle_container_bound(uint64_t inSize, uint32_t blockSize);
/* Scratch the compressor needs in device memory (per-block staging streams, sizes, scan partials). */
uint64_t hsrle_compress_workspace_size(uint64_t inSize, uint32_t blockSize);
/* The same for ONE codec.  Equal to the above except for the 8-bit Single codecs (rle8_single_short too), the 128-bit codecs and the Greedy encoders with a one-symbol
 * list (rle{16..64}_1symlut_byte_short_compress_greedy) on containers of fewer than 131 072 blocks of 1 .. 4 KiB: they have no run list encoder, and their small containers are encoded chunk by chunk (cuts behind long runs inside the blocks)
 * only if the workspace holds the chunk tables and staging area (about 2.3 x the input); with the general size they take one lane per block. */
uint64_t hsrle_compress_workspace_size_codec(int codec, uint64_t inSize, uint32_t blockSize);

/*
 * Enqueue compression of dIn[0, inSize) (device memory) into the container at dOut (device memory, capacity
 * outCapacity >= hsrle_container_bound()).  dWorkspace may be NULL: the library then keeps a cached allocation of
 * its own (not graph-capturable).  `stream` is a hipStream_t (NULL = default stream).  Asynchronous: the
 * container (including its header with totalSize) is complete when the stream reaches this point.
 */
int hsrle_compress_dev_async(int codec, const void *dIn, uint64_t inSize, void *dOut, uint64_t outCapacity,
                             uint32_t blockSize, void *dWorkspace, uint64_t workspaceSize, void *stream);
/* Same, then waits for the stream and returns the container size. */
int hsrle_compress_dev(int codec, const void *dIn, uint64_t inSize, void *dOut, uint64_t outCapacity,
                       uint32_t blockSize, uint64_t *pContainerSize, void *stream);

/* Read (device -> host, synchronising `stream`) and validate a container header. */
int hsrle_container_info_dev(const void *dContainer, uint64_t containerSize, hsrle_container_info_t *pInfo, void *stream);
/* Validate a container header that is already in host memory. */
int hsrle_container_info_host(const void *pContainer, uint64_t containerSize, hsrle_container_info_t *pInfo);

/*
 * Enqueue decompression of a container in device memory into dOut (device memory, capacity >= uncompressedSize;
 * no slack is needed and nothing outside [0, uncompressedSize) is written).  `info` must describe the container
 * (from hsrle_container_info_dev, or remembered from compression).  dStatus (device uint32_t, may be NULL) receives 0 or
 * a non-zero error mask if a block stream is malformed.
 */
int hsrle_decompress_dev_async(const void *dContainer, const hsrle_container_info_t *info, void *dOut, uint64_t outCapacity,
                               uint32_t *dStatus, void *stream);
/* Same, including the header read, a wait on the stream and the status check. */
int hsrle_decompress_dev(const void *dContainer, uint64_t containerSize, void *dOut, uint64_t outCapacity,
                         uint64_t *pUncompressedSize, void *stream);

/* Decode only blocks [firstBlock, firstBlock + blockCount) of a container into dOut + firstBlock * blockSize
 * (used by the multi-GPU sharding: every rank decodes its contiguous block range). */
int hsrle_decompress_blocks_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount,
                                      void *dOut, uint64_t outCapacity, uint32_t *dStatus, void *stream);

/*
 * Split decode: for containers with too few blocks to fill the GPU with one lane per block (one lane walks one block's packet chain:
 * ~10^5 blocks are needed; BASELINE config 3, an 88 MB frame in 4 KiB blocks, has 21 600).  One lane per block first walks the block's
 * packets and leaves the decoder state at every subBlockSize output bytes (csrc/hsrle_index.hip.h), then the block kernel runs one lane
 * per SUB-block.  The container, its block size and its compression ratio are what they are: nothing is re-encoded.
 * Round 4 -- the packet list decode (subBlockSize = HSRLE_SPLIT_PACKET_LIST, the library's choice for blocks of 256 bytes .. 16 KiB): the
 * walk leaves one 64-bit entry per PACKET (where its output starts, its literals, its symbol) instead of decoder states, and a second kernel
 * builds the output with one lane per 16 output bytes -- the packet chain is walked once, nothing else is sequential.
 *   subBlockSize: HSRLE_SPLIT_PACKET_LIST, or a multiple of 128 that divides blockSize (records at every subBlockSize bytes); 0 = the
 *   library's choice (hsrle_split_sub_block_size tells which; when it returns blockSize no split happens and no workspace is needed).
 *   dWorkspace (16-byte aligned) >= hsrle_decompress_split_workspace_size(info, blockCount, subBlockSize) bytes -- for the packet list
 *   about the uncompressed size of the blocks (blockSize / 8 + 2 entries per block: a block with more packets than that is finished by
 *   the walking lane, byte by byte).  Only enqueues kernels: graph-capturable.  dStatus as in hsrle_decompress_dev_async.
 */
#define HSRLE_SPLIT_PACKET_LIST 1u
uint32_t hsrle_split_sub_block_size(const hsrle_container_info_t *info, uint32_t subBlockSize);
uint64_t hsrle_decompress_split_workspace_size(const hsrle_container_info_t *info, uint32_t blockCount, uint32_t subBlockSize);
int hsrle_decompress_split_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, void *dOut, uint64_t outCapacity,
                                     uint32_t *dStatus, void *dWorkspace, uint64_t workspaceSize, uint32_t subBlockSize, void *stream);

#ifdef HSRLE_EXPERIMENTS
/*
 * EXPERIMENT BUILDS ONLY (-DHSRLE_EXPERIMENTS; the shipped library does not export this symbol).  Wave decode: one WAVE per block
 * (csrc/experiments/hsrle_decode_wave.hip.h) -- the block's stream goes to LDS, one lane hops through its packets leaving a descriptor
 * each, all 64 lanes expand them to whole aligned 16-byte stores.  Block sizes up to 16 KiB.  Measured slower than the split decode on the
 * 88 MB frame (427 against 189 us), which is why it does not ship.
 */
int hsrle_decompress_wave_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, void *dOut, uint64_t outCapacity,
                                    uint32_t *dStatus, void *stream);
#endif

/*
 * 64 bit hash of every block stream of blocks [firstBlock, firstBlock + blockCount) of a device-resident container into dHashes
 * (device, blockCount values): an integrity check that covers EVERY block (the big-config manifests under tests/golden/big/ hold the
 * reference encoder's values).  For the bytes p[0, len) of a stream:
 *   h = 0x9E3779B97F4A7C15 ^ (len * 0xD6E8FEB86659FD93);
 *   for every 8-byte little-endian word w (the last one zero padded): h = rotl64(h ^ w, 27) * 0x9E3779B97F4A7C15 + 0x165667B19E3779F9;
 *   h ^= h >> 31
 */
int hsrle_hash_blocks_dev_async(const void *dContainer, const hsrle_container_info_t *info, uint32_t firstBlock, uint32_t blockCount, uint64_t *dHashes, void *stream);

/* ---------------------------------------------------------------------------------------------------------- */
/* 3. host convenience wrappers (allocate device buffers, copy, run 2, copy back)                              */

int hsrle_compress_host(int codec, const void *pIn, uint64_t inSize, void *pOut, uint64_t outCapacity, uint32_t blockSize, uint64_t *pContainerSize);
int hsrle_decompress_host(const void *pContainer, uint64_t containerSize, void *pOut, uint64_t outCapacity, uint64_t *pUncompressedSize);

/* ---------------------------------------------------------------------------------------------------------- */
/* 4. multi-GPU: the path shards by independent blocks -- rank r of W encodes / decodes the contiguous block range
 *    [r * n / W, (r + 1) * n / W) with sections 2 of this header and needs no collective for that.  The ONE exchange step is
 *    assembling one container from the per-rank containers (and cutting one into per-rank containers); these functions do it over
 *    RCCL (xGMI inside a node): an all-gather of the per-rank sizes, then grouped point-to-point transfers of [offset table | payload]
 *    straight into their final places, offsets re-based by a small kernel.  One process per GPU; the communicator belongs to the caller.
 *    RCCL is loaded on first use (librccl.so.1); HSRLE_ERR_UNSUPPORTED if it is not there.  The reference has nothing comparable (it
 *    is single device); its closest precedent is the sub-section container of rle8m (src/rle8_low_entropy_cpu.c:131-191).             */

#define HSRLE_RCCL_ID_BYTES 128
/* rank 0: a fresh id (ncclGetUniqueId) to be handed to every rank by whatever the host program uses (MPI, torch.distributed, a file) */
int hsrle_rccl_unique_id(void *id128);
/* every rank, on its current HIP device: ncclCommInitRank.  *pComm is an ncclComm_t; a communicator the program already has works too */
int hsrle_rccl_comm_create(const void *id128, int worldSize, int rank, void **pComm);
int hsrle_rccl_comm_destroy(void *comm);
/* what the communicator itself says it spans (ncclCommCount / ncclCommUserRank): the figure a benchmark line should carry, not WORLD_SIZE */
int hsrle_rccl_comm_ranks(void *comm, int *pWorldSize, int *pRank);
/*
 * Collective over `comm`.  dLocal / localSize: this rank's container (device memory; NULL / 0 if the rank owns no block).  On the root,
 * dOut (device, capacity >= the sum of the parts, at most hsrle_container_bound(totalUncompressedSize, blockSize)) receives the one
 * container; *pTotalSize its size (every rank learns it).  All non-empty ranks must agree on codec and block size (HSRLE_ERR_FORMAT).
 * Enqueues on `stream` after two small synchronising reads (header, sizes).
 */
int hsrle_gather_container_rccl(void *comm, int root, const void *dLocal, uint64_t localSize, uint64_t totalUncompressedSize, void *dOut, uint64_t outCapacity,
                                uint64_t *pTotalSize, void *stream);
/* The inverse: the root's container is cut into one container per rank (contiguous block ranges, blocks renumbered from 0). */
int hsrle_scatter_container_rccl(void *comm, int root, const void *dContainer, uint64_t containerSize, void *dLocal, uint64_t localCapacity, uint64_t *pLocalSize, void *stream);

/* ---------------------------------------------------------------------------------------------------------- */
/* synthetic workloads of BASELINE.json generated directly in device memory (bench / tests)     */

#define HSRLE_SYNTH_RUNS 0  /* run-distributed(W): ~50 % of the bytes in runs, mean segment ~32 symbols          */
#define HSRLE_SYNTH_VIDEO 1 /* video-frame-shaped: zero dominated, short bursts of small values                  */
int hsrle_synth_dev_async(int kind, int symbolBytes, uint64_t seed, void *dOut, uint64_t size, void *stream);

/* library / device introspection */
int hsrle_device_count(void);

/* 1 if the library was built with -DHSRLE_EXPERIMENTS (developer builds: environment knobs that force kernel variants, kernels that were
 * measured slower than the shipped ones), 0 for the shipped build: no environment variable is read in any launch path. */
/* Memory the library keeps: calls that are given no workspace (hsrle_compress_dev, hsrle_decompress_dev, the host-pointer functions)
 * take their scratch stream-ordered from a memory pool the LIBRARY owns on the calling thread's device -- never from the device's default
 * pool, whose settings belong to the application.  Freed scratch stays in that pool up to the retention (default 2 GiB; UINT64_MAX keeps
 * everything, 0 returns everything at the next synchronisation); hsrle_trim() returns what the pool holds to the driver now.  Calls that
 * are handed their workspace (the *_async forms) allocate nothing.
 * hsrle_scratch_retention(bytes) applies the value to the pool of the calling thread's CURRENT device and makes it the default of every
 * pool the library creates later (one per device, on first use); pools that already exist on other devices keep their setting until the
 * function is called with that device current.  hsrle_trim() likewise acts on the current device's pool only. */
int hsrle_scratch_retention(uint64_t bytes);
int hsrle_trim(void);

int hsrle_experiments_enabled(void);

/* Which encoder hsrle_compress_dev[_async] uses for a container of `uncompressedSize` bytes in blocks of `blockSize` (no device needed;
 * the streams are the reference's whichever it is -- this is for capacity planning and for reading profiles):
 *   HSRLE_PATH_RING      one lane per block (the ring encoders): containers of >= 131 072 blocks, and whatever the other two do not take
 *   HSRLE_PATH_SPLIT     small containers, chunks inside the blocks
 *   HSRLE_PATH_RUN_LIST  small containers of 1 .. 4 KiB blocks, a wave per block
 * -1: bad codec / block size / size. */
#define HSRLE_PATH_RING 0
#define HSRLE_PATH_SPLIT 1
#define HSRLE_PATH_RUN_LIST 2
#define HSRLE_PATH_POSITION_PARALLEL 3   /* round 5: one wave per block, blocks of at most 4 KiB, any number of them: rle8_multi, rle8_packed_multi, the plain / Packed codecs of 2 .. 8 byte symbols, the 3 symbol LUT codecs of 3 .. 8 byte symbols, the Short codecs with no / one symbol in the list (1 .. 8 byte symbols) and with three (6 / 8 byte symbols): 56 codecs */
/* round 6: ... and, with blocks ABOVE 4 KiB walked in 4 KiB windows (a wave per block, then a wave per window: csrc/hsrle_encode8pw.hip.h, hsrle_encodeSpw.hip.h,
 * hsrle_encodeLpw.hip.h): rle8_multi / rle8_packed_multi and the plain / Packed codecs of 2 .. 8 byte symbols with blocks of any size; every LUT codec and the
 * position-parallel Short codecs with blocks below 1 MiB: 86 codecs (not: the 8 bit Single codecs, the 128 bit codecs) */
int hsrle_encode_path(int codec, uint64_t uncompressedSize, uint32_t blockSize);

/* A hash of the library's sources and build flags (set by the Makefile; "unknown" for other build recipes): measurement files that
 * describe one build of the kernels (profiles/r03_traffic.json) carry it, and bench.py refuses a file of another build. */
const char *hsrle_build_id(void);
/* Workgroups (= wavefronts: every kernel here runs one wave per workgroup) of the codec's decode (decode != 0) or encode kernel
 * that are resident on one CU at a time, as the HIP runtime computes it from the kernel's LDS and register use; 0 on error.
 * The kernels are latency bound, so this is the first number to look at when a build got slower. */
/* ---------------------------------------------------------------------------------------------------------- */
/* rle8m: the reference's own GPU decode path.  `rle8m_opencl_*` are the names of      */
/* src/rle.h:464-466 (src/rle8_ocl.c:56, :185, :265); `rle8m_decompress` is the CPU twin of the same format   */
/* (src/rle.h:63, src/rle8_low_entropy_cpu.c:193-250).  Host pointers; one wave (few or large sub-sections)  */
/* or one lane (many small ones) decodes one sub-section.                                                     */
bool rle8m_opencl_init(const size_t inputDataSize, const size_t outputDataSize, const size_t maxSubsectionCount);
void rle8m_opencl_destroy(void);
uint32_t rle8m_opencl_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8m_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);

/* rle8m encode, the GPU twin of the reference's CPU function (src/rle.h:61-62, src/rle8_low_entropy_cpu.c:126-191): same stream,
 * same failure rule (0 when a section outgrows what is left of the output at its turn) -- plus one the reference lacks: 0 as well
 * when a section's STREAM (up to twice the section) does not fit, where the reference writes behind pOut + outSize.           */
uint32_t rle8m_compress_bounds(const uint32_t subSections, const uint32_t inSize);
uint32_t rle8m_compress(const uint32_t subSections, const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
/* device-resident encode: only enqueues kernels; *dStatus != 0 afterwards = the reference would have returned 0 */
/* The UNSECTIONED forms of the same codec (src/rle.h:53-57, :90-93; src/rle8_low_entropy_cpu.c:6-124,
 * src/rle8_low_entropy_short_cpu.c:16-124): [u32 compressedSize][u32 uncompressedSize][info][ONE stream], i.e. an rle8m stream of one
 * section without its section-count field.  Same names, arguments and return convention as the reference (0 = failure); the streams are
 * the reference's byte for byte, the decoders read the reference's streams.  The Short form cuts runs every 32 bytes instead of 255;
 * *_only_max_frequency lets one symbol -- the one that saves the most -- carry repeat codes.  The one stream is worked on by many waves:
 * the encoder cuts its input at run boundaries (no token crosses one), the decoder cuts the stream anywhere and finds out from the byte
 * values in front of a cut whether it starts with a symbol or a repeat code (csrc/hsrle_rle8m.hip.h).  Unlike the reference a stream that
 * outgrows `outSize` is a failure, never a write behind the caller's buffer (rle8_low_entropy_cpu.c:476 only asks for outSize >= inSize).
 * The split-phase helpers that pass the codec's tables through host structs follow below. */
uint32_t rle8_low_entropy_compress_bounds(const uint32_t inSize);
uint32_t rle8_low_entropy_decompressed_size(const uint8_t *pIn, const uint32_t inSize);
uint32_t rle8_low_entropy_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_compress_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_short_compress_bounds(const uint32_t inSize);
uint32_t rle8_low_entropy_short_compress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_short_compress_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_short_decompress(const uint8_t *pIn, const uint32_t inSize, uint8_t *pOut, const uint32_t outSize);

/* The split-phase helpers of the same codec (src/rle.h:67-96; src/rle8_low_entropy_cpu.c:254-600, :930-1022, src/rle8_low_entropy_short_cpu.c:128-534):
 * the statistics pass, the header writer / reader and the two stream bodies as separate calls, tables handed through host structs with the
 * reference's layout.  rle8_low_entropy_compress is exactly  [u32 size][u32 inSize] + write_compress_info(get_compress_info) +
 * compress_with_info,  as in the reference (rle8_low_entropy_cpu.c:11-50), and the tests check that identity.
 *   get_compress_info*      statistics + tables on the GPU (k_rle8m_stats_wave, k_rle8m_info), false = bad arguments / no device
 *   write_compress_info     host only: 32 flag bytes, the symbol count (a uint8: 256 symbols are written as 0 and then 255 of them, sic), the symbols
 *   compress_with_info      the stream body for ANY tables (rle[] / symbolsByProb[] need not come from get_compress_info); returns its size,
 *                           0 = failure.  Unlike the reference a body that outgrows `outSize` is a failure, not a write behind the buffer.
 *   read_decompress_info    host only; returns the bytes of the info (33 + listed symbols)
 *   decompress_with_info    pIn .. pEnd = the body; returns expectedOutSize, 0 = malformed body or tables that are not a permutation
 *                           (the reference returns expectedOutSize whatever it read)                                                  */
typedef struct rle8_low_entropy_compress_info_t { bool rle[256]; uint8_t symbolsByProb[256]; uint8_t symbolCount; } rle8_low_entropy_compress_info_t;   /* src/rle.h:67-72 */
typedef struct rle8_low_entropy_decompress_info_t { bool rle[256]; uint8_t symbolToCount[256]; } rle8_low_entropy_decompress_info_t;                     /* src/rle.h:81-85 */
bool rle8_low_entropy_get_compress_info(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_compress_info_t *pCompressInfo);
bool rle8_low_entropy_get_compress_info_only_max_frequency(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_compress_info_t *pCompressInfo);
uint32_t rle8_low_entropy_write_compress_info(rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_compress_with_info(const uint8_t *pIn, const uint32_t inSize, const rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_read_decompress_info(const uint8_t *pIn, const uint32_t inSize, rle8_low_entropy_decompress_info_t *pDecompressInfo);
uint32_t rle8_low_entropy_decompress_with_info(const uint8_t *pIn, const uint8_t *pEnd, const rle8_low_entropy_decompress_info_t *pDecompressInfo, uint8_t *pOut, const uint32_t expectedOutSize);
uint32_t rle8_low_entropy_short_compress_with_info(const uint8_t *pIn, const uint32_t inSize, const rle8_low_entropy_compress_info_t *pCompressInfo, uint8_t *pOut, const uint32_t outSize);
uint32_t rle8_low_entropy_short_decompress_with_info(const uint8_t *pIn, const uint8_t *pEnd, const rle8_low_entropy_decompress_info_t *pDecompressInfo, uint8_t *pOut, const uint32_t expectedOutSize);

/* The same, device resident (variant: bit 0 the Short form, bit 1 only_max_frequency).  Compress only enqueues (the stream's size is its
 * first u32; *dStatus != 0: the stream did not fit `outCapacity`); decompress reads the header and the verdict (two stream synchronisations). */
uint64_t hsrle_low_entropy_workspace_size(uint32_t inSize);
int hsrle_low_entropy_compress_dev_async(const void *dIn, uint32_t inSize, int variant, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *dStatus, void *stream);
uint64_t hsrle_low_entropy_decompress_workspace_size(uint64_t streamSize);
int hsrle_low_entropy_decompress_dev(const void *dStream, uint64_t streamSize, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize, uint32_t *pUncompressedSize, void *stream);

uint64_t hsrle_rle8m_compress_workspace_size(uint32_t inSize, uint32_t sections);
int hsrle_rle8m_compress_dev_async(const void *dIn, uint32_t inSize, uint32_t sections, void *dOut, uint64_t outCapacity, void *dWorkspace, uint64_t workspaceSize,
                                   uint32_t *dStatus, void *stream);

/* device-resident form: the stream header is read once (synchronises), the decode only enqueues a kernel */
typedef struct hsrle_rle8m_info { uint32_t compressedSize, uncompressedSize, sections; } hsrle_rle8m_info_t;
int hsrle_rle8m_info_dev(const void *dStream, uint64_t streamSize, hsrle_rle8m_info_t *pInfo, void *stream);
int hsrle_rle8m_decompress_dev_async(const void *dStream, const hsrle_rle8m_info_t *info, void *dOut, uint64_t outCapacity, uint32_t *dStatus, void *stream);

int hsrle_kernel_waves_per_cu(int codec, int decode);
const char *hsrle_version(void);

#ifdef __cplusplus
}
#endif

#endif /* HSRLE_H */
