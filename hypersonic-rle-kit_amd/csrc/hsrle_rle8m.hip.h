// hsrle_rle8m.hip.h -- decoder for rle8m streams: the low-entropy codec in its sub-sectioned form, the format of the reference's
// own GPU path.
//
// Replaces: src/rle8_ocl.c:265-413 (rle8m_opencl_decompress: upload, one work-item per section, blocking read-back),
//           src/rle8_ocl_kernel.h:8-80 (the two OpenCL kernels), and on the CPU side src/rle8_low_entropy_cpu.c:193-250
//           (rle8m_decompress) with :545-606 (read_decompress_info) and :930-1021 (decompress_with_info).
//
// Stream: [u32 compressedSize][u32 uncompressedSize][u32 sections][u32 end offset of section 0 .. sections-2][32-byte bitmap of the
// symbols that carry a repeat code][u8 n (0 = 255)][n symbols ordered by run count][section streams back to back].  A section is
// a byte sequence in which every flagged symbol is followed by the code of how many more of it follow (0..254; code = the count's
// position in the symbol order); section k decodes to uncompressedSize / sections bytes (the last one to the remainder).
//
// One lane decodes one section (the reference's OpenCL kernel does the same with one work-item per section and byte accesses to
// global memory); here the input is read through a 16-byte register window and the output is assembled in a 16-byte register
// accumulator, so global memory only sees 16-byte accesses, and the two lookup tables live in LDS.  First-generation data path
// (per-lane global access): the sections of neighbouring lanes are section-size apart.
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

enum Rle8mError : uint32_t { RLE8M_ERR_HEADER = 1u, RLE8M_ERR_STREAM = 2u };

__global__ __launch_bounds__(64) void k_rle8m_decode(const uint8_t *__restrict__ s, uint64_t streamBytes, uint8_t *__restrict__ out, uint32_t *__restrict__ status)
{
  __shared__ uint32_t rleBits[8];          // bit b: symbol b is followed by a repeat code
  __shared__ uint8_t codeToCount[256];     // repeat code -> count (rle8_low_entropy_cpu.c:569-600)
  __shared__ uint8_t listed[256];
  __shared__ uint32_t hdr[4];              // data start, or 0 when the header is inconsistent

  const uint32_t lane = threadIdx.x;
  const uint32_t expIn = ld32(s), expOut = ld32(s + 4), sections = ld32(s + 8);
  const uint64_t info = 12ull + 4ull * (uint64_t)(sections - 1u);

  // ---- header: every workgroup rebuilds the two tables (a few hundred instructions against ~1e6 for its 64 sections) ----
  bool bad = expIn > streamBytes || sections == 0u || info + 33u > (uint64_t)expIn;
  uint32_t listedCount = 0;
  if (!bad)
  {
    listedCount = s[info + 32];
    if (listedCount == 0u) listedCount = 255u;                          // sic: 256 listed symbols are stored as 0 and read as 255
    bad = info + 33u + listedCount > (uint64_t)expIn;
  }
  if (!bad)
  {
    if (lane < 8u) rleBits[lane] = ld32(s + info + 4u * lane);
    for (uint32_t k = lane; k < 256u; k += 64u) listed[k] = 0;
    __syncthreads();
    for (uint32_t k = lane; k < listedCount; k += 64u)
    {
      const uint32_t sym = s[info + 33u + k];
      codeToCount[sym] = (uint8_t)k;
      listed[sym] = 1;
    }
    __syncthreads();
    if (lane == 0u)
    {
      uint32_t next = listedCount;                                       // the symbols that are not listed take the remaining counts in ascending order
      for (uint32_t k = 0; k < 256u; k++)
        if (!listed[k]) codeToCount[k] = (uint8_t)next++;
      hdr[0] = (uint32_t)(info + 33u + listedCount);
    }
  }
  else if (lane == 0u)
    hdr[0] = 0u;
  __syncthreads();

  const uint32_t dataStart = hdr[0];
  const uint32_t k = blockIdx.x * 64u + lane;
  if (dataStart == 0u)
  {
    if (k == 0u && status) atomicOr(status, RLE8M_ERR_HEADER);
    return;
  }
  if (k >= sections)
    return;

  const uint32_t ss = expOut / sections;
  const uint32_t begin = (k == 0u) ? dataStart : ld32(s + 12u + 4u * (k - 1u));
  const uint32_t end = (k + 1u < sections) ? ld32(s + 12u + 4u * k) : expIn;
  const uint32_t want = (k + 1u < sections) ? ss : expOut - ss * (sections - 1u);
  uint8_t *const o = out + (uint64_t)k * ss;

  if (begin < dataStart || end < begin || end > expIn)
  {
    if (status) atomicOr(status, RLE8M_ERR_HEADER);
    return;
  }

  // ---- input window: 16 stream bytes at wpos; output accumulator: the 16-byte chunk that contains section position op ----
  uint32_t wpos = 0xFFFFFF00u;
  uint64_t wlo = 0, whi = 0;
  auto in_byte = [&](uint32_t pos) -> uint32_t {
    if (pos - wpos >= 16u)
    {
      wpos = pos;
      if ((uint64_t)pos + 16u <= streamBytes) { wlo = ld64(s + pos); whi = ld64(s + pos + 8); }
      else
      {
        wlo = 0; whi = 0;
        for (uint32_t j = 0; j < 16u && (uint64_t)pos + j < streamBytes; j++)
        {
          if (j < 8u) wlo |= (uint64_t)s[pos + j] << (8u * j); else whi |= (uint64_t)s[pos + j] << (8u * (j - 8u));
        }
      }
    }
    const uint32_t d = pos - wpos;
    return (uint32_t)(((d & 8u) ? whi : wlo) >> (8u * (d & 7u))) & 0xFFu;
  };

  uint64_t alo = 0, ahi = 0;
  uint32_t op = 0;
  auto put = [&](uint32_t b) {
    const uint32_t sh = 8u * (op & 7u);
    if (op & 8u) ahi |= (uint64_t)b << sh; else alo |= (uint64_t)b << sh;
    op++;
    if ((op & 15u) == 0u)
    {
      st64(o + op - 16u, alo); st64(o + op - 8u, ahi);
      alo = 0; ahi = 0;
    }
  };

  uint32_t ip = begin;
  bool err = false;
  while (ip < end)
  {
    const uint32_t b = in_byte(ip++);
    if (op >= want) { err = true; break; }
    put(b);
    if ((rleBits[b >> 5] >> (b & 31u)) & 1u)
    {
      if (ip >= end) { err = true; break; }
      uint32_t count = codeToCount[in_byte(ip++)];
      if (count > want - op) { err = true; break; }
      // the run is merged into the accumulator with byte masks (the accumulator is zero above op), whole chunks are stored as vectors
      const uint64_t bb = (uint64_t)b * 0x0101010101010101ull;
      while (count != 0u)
      {
        const uint32_t c = op & 15u;
        if (c == 0u && count >= 16u)
        {
          for (; count >= 16u; count -= 16u, op += 16u) { st64(o + op, bb); st64(o + op + 8u, bb); }
          continue;
        }
        const uint32_t take = (16u - c < count) ? 16u - c : count, e = c + take;   // chunk bytes [c, e) become b
        const uint64_t fromLo = (c < 8u) ? (~0ull << (8u * c)) : 0ull, fromHi = (c < 8u) ? ~0ull : (~0ull << (8u * (c - 8u)));
        const uint64_t belowLo = (e >= 8u) ? ~0ull : ~(~0ull << (8u * e)), belowHi = (e <= 8u) ? 0ull : ((e == 16u) ? ~0ull : ~(~0ull << (8u * (e - 8u))));
        alo |= bb & fromLo & belowLo;
        ahi |= bb & fromHi & belowHi;
        op += take;
        count -= take;
        if ((op & 15u) == 0u)
        {
          st64(o + op - 16u, alo); st64(o + op - 8u, ahi);
          alo = 0; ahi = 0;
        }
      }
    }
  }
  if (!err && op != want) err = true;
  for (uint32_t j = op & ~15u; j < op && !err; j++)                    // the last partial chunk
    o[j] = (uint8_t)(((j & 8u) ? ahi : alo) >> (8u * (j & 7u)));
  if (err && status) atomicOr(status, RLE8M_ERR_STREAM);
}

} // namespace hsrle
