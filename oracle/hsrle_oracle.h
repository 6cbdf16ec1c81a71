/*
 * hsrle_oracle.h -- CPU restatement of the rleX_extreme codecs.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle of the repository: a plain, scalar C99 restatement of the
 * reference algorithm (rainerzufalldererste/hypersonic-rle-kit, src/rle8_extreme_cpu.{c,h},
 * src/rleX_extreme_cpu*.h, src/rle{24,48,128}_extreme_cpu*.h, src/rleX_Xsl*.h).
 * Nothing in the product path (hypersonic-rle-kit_amd/) may include, link or call it; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte against the compiled
 * reference (oracle/_ref/libhsrle_ref.so, built from /root/reference/src by oracle/Makefile)
 * by tests/test_oracle_vs_ref.py, and against committed golden vectors minted from that
 * build (tests/golden/, tests/golden/make_golden.py) by tests/test_oracle_golden.py.
 */
#ifndef HSRLE_ORACLE_H
#define HSRLE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* codec families (reference: §2.1 of SURVEY.md; src/rle.h:100-394) */
enum {
  HSO_PLAIN = 0,         /* rle8_multi / rle{W}_{sym,byte}                 */
  HSO_PACKED = 1,        /* rle8_packed_multi / rle{W}_{sym,byte}_packed   */
  HSO_LUT3 = 2,          /* rle{W}_3symlut[_sym|_byte]                     */
  HSO_LUT7 = 3,          /* rle{W}_7symlut[_sym|_byte]                     */
  HSO_SINGLE = 4,        /* rle8_single        (8 bit only)                */
  HSO_PACKED_SINGLE = 5, /* rle8_packed_single (8 bit only)                */
  /* Short family (src/rle.h:202-348): one-byte packed headers, 0 / 1 / 3 / 7 symbol LUT */
  HSO_SHORT0 = 6,        /* rle8_multi_short / rle{W}_{sym,byte}_short                */
  HSO_SHORT1 = 7,        /* rle8_1symlut_short / rle{W}_1symlut_{sym,byte}_short      */
  HSO_SHORT3 = 8,        /* rle8_3symlut_short / rle{W}_3symlut_{sym,byte}_short      */
  HSO_SHORT7 = 9,        /* rle8_7symlut_short / rle{W}_7symlut_{sym,byte}_short      */
  /* Greedy encoders (src/rle.h:398-416): rle{W}_{1,3,7}symlut_byte_short_compress_greedy; their streams are Short streams and
   * decode with rle{W}_{1,3,7}symlut_byte_short_decompress (src/codec_funcs.h:298-388); symAligned must be 0, W in 16..64 */
  HSO_GREEDY1 = 10, HSO_GREEDY3 = 11, HSO_GREEDY7 = 12,
  HSO_SINGLE_SHORT = 13  /* rle8_single_short (8 bit only; src/rle.h:223-224) */
};

/* reference: rle_compress_bounds, src/rle8_extreme_cpu.c:22-28 */
uint32_t hso_compress_bounds(uint32_t inSize);
/* reference: rle_decompress_additional_size, src/rle8_extreme_cpu.c:17-20 */
uint32_t hso_decompress_additional_size(void);

/*
 * family: HSO_*;  symbolBytes: 1,2,3,4,6,8,16;  symAligned: 1 = "sym" variants, 0 = "byte"
 * (UNBOUND) variants (ignored for symbolBytes == 1).
 * Same contract as the reference functions: returns 0 on failure, otherwise the stream size
 * (compress) or the uncompressed size (decompress).
 * Positions >= inSize never match anything (SURVEY.md §8c guard-padding rule).
 */
uint32_t hso_compress(int family, int symbolBytes, int symAligned,
                      const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);
uint32_t hso_decompress(int family, int symbolBytes, int symAligned,
                        const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);

/* Call by the reference's own function name, e.g. "rle8_packed_multi_compress",
 * "rle64_3symlut_byte_decompress".  Returns 0xFFFFFFFF if the name is unknown. */
uint32_t hso_call(const char *rleName, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);

/* Resolve a reference function name into (family, symbolBytes, symAligned, isDecompress).
 * Returns 1 if known.  For "rle8_decompress"/"rle8_packed_decompress" the family is the
 * multi one; the decoder switches on the stream's mode byte like the reference does. */
int hso_resolve(const char *rleName, int *family, int *symbolBytes, int *symAligned, int *isDecompress);

/* Block container helpers used by tests and by the CPU baseline: encode/decode `nBlocks`
 * independent blocks of `blockSize` input bytes (last one shorter); streams are written at
 * pOut + i * stride, sizes into pSizes[i].  Single threaded. */
uint32_t hso_compress_blocks(int family, int symbolBytes, int symAligned,
                             const uint8_t *pIn, uint64_t inSize, uint32_t blockSize,
                             uint8_t *pOut, uint32_t stride, uint32_t *pSizes);

/* Decode nBlocks block streams (payload + offsets[i] .. offsets[i+1]) into pOut + i * blockSize; returns bytes produced. */
uint64_t hso_decompress_blocks(int family, int symbolBytes, int symAligned, const uint8_t *payload, const uint64_t *offsets,
                               uint64_t nBlocks, uint32_t blockSize, uint8_t *pOut, uint64_t outSize);

/* rle8m: the sub-sectioned low-entropy codec, the format of the reference's GPU (OpenCL) decoder (SURVEY.md 8a row a14;
 * reference: rle8m_compress_bounds / rle8m_compress / rle8m_decompress, src/rle8_low_entropy_cpu.c:126-250) */
uint32_t hso_rle8m_compress_bounds(uint32_t subSections, uint32_t inSize);
uint32_t hso_rle8m_compress(uint32_t subSections, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);
uint32_t hso_rle8m_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);

/* the unsectioned forms of the low-entropy codec (reference: rle8_low_entropy[_short]_compress[_only_max_frequency] /
 * _decompress, src/rle8_low_entropy_cpu.c:6-124, src/rle8_low_entropy_short_cpu.c:16-124).  variant: bit 0 Short form, bit 1 only_max_frequency */
uint32_t hso_low_entropy_compress_bounds(uint32_t inSize);
uint32_t hso_low_entropy_compress(int variant, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);
uint32_t hso_low_entropy_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize);

/* oracle/hsrle_synth.c: bytes [offset, offset + size) of a deterministic synthetic workload (offset multiple of 64 KiB) */
int hso_synth(int kind, int symbolBytes, uint64_t seed, uint64_t offset, uint8_t *out, uint64_t size);

#ifdef __cplusplus
}
#endif

#endif
