// hsrle_rle8m.hip.h -- decoder and encoder for rle8m streams: the low-entropy codec in its sub-sectioned form, the format of the reference's
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
// Two forms of each direction (the choice is made in hsrle_capi.hip).  Many small sections: one LANE per section, like the reference's OpenCL kernel (one
// work-item per section, byte accesses to global memory) -- but the input comes through a per-lane LDS ring that all lanes of a wave
// top up in the same loop trip (LaneRing, hsrle_common.hip.h) and the output is assembled in a 16-byte register accumulator, so global
// memory only sees 16-byte accesses; the lookup tables live in LDS.  Fewer, larger sections: one WAVE per section, 64 stream bytes
// per step (k_rle8m_decode_wave, k_rle8m_encode_wave): both grammars are position-parallel, see the kernels.
#pragma once

#include "hsrle_common.hip.h"

namespace hsrle {

enum Rle8mError : uint32_t { RLE8M_ERR_HEADER = 1u, RLE8M_ERR_STREAM = 2u };

// header of an rle8m stream -> the two lookup tables in LDS; returns the offset of the first section stream, 0 when the header is
// inconsistent.  Every workgroup rebuilds the tables (a few hundred instructions against ~1e6 for its sections).
__device__ __forceinline__ uint32_t rle8m_tables(const uint8_t *__restrict__ s, uint64_t streamBytes, uint32_t lane, uint32_t *rleBits, uint8_t *codeToCount, uint8_t *listed, uint32_t *hdr)
{
  const uint32_t expIn = ld32(s), sections = ld32(s + 8);
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
    {
      uint32_t next = listedCount;                                       // the symbols that are not listed take the remaining counts in ascending order:
      for (uint32_t c = 0; c < 256u; c += 64u)                           // 64 symbols per step, ranks from the ballot of the unlisted ones
      {
        const bool unl = !listed[c + lane];
        const uint64_t m = __builtin_amdgcn_ballot_w64(unl);
        if (unl) codeToCount[c + lane] = (uint8_t)(next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)));
        next += (uint32_t)__builtin_popcountll(m);
      }
      if (lane == 0u) hdr[0] = (uint32_t)(info + 33u + listedCount);
    }
  }
  else if (lane == 0u)
    hdr[0] = 0u;
  __syncthreads();

  return hdr[0];
}

__global__ __launch_bounds__(64) void k_rle8m_decode(const uint8_t *__restrict__ s, uint64_t streamBytes, uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                      uint32_t wantOut, uint32_t wantSections)
{
  __shared__ uint32_t rleBits[8];          // bit b: symbol b is followed by a repeat code
  __shared__ uint8_t codeToCount[256];     // repeat code -> count (rle8_low_entropy_cpu.c:569-600)
  __shared__ uint8_t listed[256];
  __shared__ uint32_t hdr[4];              // data start, or 0 when the header is inconsistent
  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * kLaneRingStride];

  const uint32_t lane = threadIdx.x;
  const uint32_t expIn = ld32(s), expOut = ld32(s + 4), sections = ld32(s + 8);
  const uint32_t dataStart = rle8m_tables(s, streamBytes, lane, rleBits, codeToCount, listed, hdr);
  const uint32_t k = blockIdx.x * 64u + lane;
  // the grid and the output capacity were sized from the caller's info: a stream header that says something else is an error, not a write
  if (dataStart == 0u || expOut != wantOut || sections != wantSections)
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

  // ---- input: LaneRing over the stream (at most two bytes per trip: top-up every 32 trips); output accumulator: the 16-byte chunk
  // that contains section position op ----
  LaneRing in{ s, expIn, ring + lane * kLaneRingStride, begin & ~15u };
  auto in_byte = [&](uint32_t pos) -> uint32_t { return in.get(pos); };

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
  for (uint32_t trip = 0; ip < end; trip++)
  {
    if ((trip & 31u) == 0u) in.topup<5>(ip);
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


// ------------------------------------------------------------------------------------------------------------------
// rle8m decode, one WAVE per section: for streams with few, large sections (one lane per section leaves the GPU to a lane's dependent
// instruction chain: 16 384 sections of 64 KiB decode at 97 GiB/s).  64 stream bytes per step, one per lane:
//   * which bytes are symbols and which are repeat codes follows from the flagged bytes alone: a byte is a code iff the run of
//     flagged-valued bytes right before it has odd length (the run starts with a symbol; symbol, code, symbol, ... while the values stay
//     flagged).  The run length below a lane comes from the ballot of the flagged lanes, its parity across windows is one carried bit;
//   * a flagged symbol in lane 63 has its code in the next window: it emits its own byte now, lane 0 of the next window the repeats;
//   * output offsets = exclusive wave scan of the packet lengths (1 + count); every lane writes its own packet.
__global__ __launch_bounds__(64) void k_rle8m_decode_wave(const uint8_t *__restrict__ s, uint64_t streamBytes, uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                      uint32_t wantOut, uint32_t wantSections)
{
  __shared__ uint32_t rleBits[8];
  __shared__ uint8_t codeToCount[256];
  __shared__ uint8_t listed[256];
  __shared__ uint32_t hdr[4];

  const uint32_t lane = threadIdx.x;
  const uint32_t expIn = ld32(s), expOut = ld32(s + 4), sections = ld32(s + 8);
  const uint32_t dataStart = rle8m_tables(s, streamBytes, lane, rleBits, codeToCount, listed, hdr);
  const uint32_t k = blockIdx.x;
  if (dataStart == 0u || expOut != wantOut || sections != wantSections)    // (see k_rle8m_decode)
  {
    if (k == 0u && lane == 0u && status) atomicOr(status, RLE8M_ERR_HEADER);
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
    if (lane == 0u && status) atomicOr(status, RLE8M_ERR_HEADER);
    return;
  }

  uint32_t op = 0, carry = 0, pendSym = 0;
  bool err = false;
  uint32_t bNext = (begin + lane < end) ? (uint32_t)s[begin + lane] : 0u;
  for (uint32_t ip = begin; ip < end; ip += 64u)
  {
    const uint32_t pos = ip + lane;
    const bool valid = pos < end;
    const uint32_t b = bNext;
    bNext = (pos + 64u < end) ? (uint32_t)s[pos + 64u] : 0u;              // the next window is on its way while this one is decoded
    const bool fl = valid && ((rleBits[b >> 5] >> (b & 31u)) & 1u);
    const uint64_t F = __builtin_amdgcn_ballot_w64(fl);
    if (F == 0ull && carry == 0u)                                         // no repeat code in this window: its bytes are the output
    {
      const uint32_t nv = (end - ip < 64u) ? end - ip : 64u;
      if (nv > want - op) { err = true; break; }
      if (valid) o[op + lane] = (uint8_t)b;
      op += nv;
      continue;
    }
    // length of the run of flagged lanes right below this lane
    uint32_t below = 0;
    if (lane != 0u)
    {
      const uint64_t x = ~F << (64u - lane);                              // bit 63 = lane - 1
      below = (x == 0ull) ? lane : (uint32_t)__builtin_clzll(x);
      if (below > lane) below = lane;
    }
    const bool isCode = valid && (((below + ((below == lane) ? carry : 0u)) & 1u) != 0u);
    const bool isSym = valid && !isCode;
    const uint32_t nb = (uint32_t)__shfl_down((int)b, 1);
    uint32_t runLen = 0, v = b;
    bool bad = false;
    if (isSym)
    {
      runLen = 1u;
      if (fl)
      {
        if (pos + 1u >= end) bad = true;                                  // a flagged symbol without its code
        else if (lane != 63u) runLen += codeToCount[nb];
      }
    }
    else if (isCode && lane == 0u)                                        // (only lane 0 can be a code whose symbol sits in the window before)
    {
      runLen = codeToCount[b];
      v = pendSym;
    }
    // exclusive scan of the packet lengths
    uint32_t incl = runLen;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1)
    {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += t;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull || total > want - op) { err = true; break; }
    // every lane writes its packet: bytes up to the next 16-byte line of the output, whole lines, the rest
    {
      uint8_t *q = o + op + (incl - runLen);
      uint32_t left = runLen;
      const uint32_t vv = v * 0x01010101u;
      const u32x4 line{ vv, vv, vv, vv };
      // at most four stores up to the next 16-byte line, whole lines, at most four stores for the rest (a byte loop here was most
      // of the kernel's instructions)
      const uint64_t v8 = (uint64_t)vv * 0x0000000100000001ull;
      if (((uintptr_t)q & 1u) && left >= 1u) { *q = (uint8_t)v; q += 1; left -= 1u; }
      if (((uintptr_t)q & 2u) && left >= 2u) { st16(q, vv & 0xFFFFu); q += 2; left -= 2u; }
      if (((uintptr_t)q & 4u) && left >= 4u) { st32(q, vv); q += 4; left -= 4u; }
      if (((uintptr_t)q & 8u) && left >= 8u) { st64(q, v8); q += 8; left -= 8u; }
      for (; left >= 16u; left -= 16u, q += 16) st128(q, line);
      if (left & 8u) { st64(q, v8); q += 8; }
      if (left & 4u) { st32(q, vv); q += 4; }
      if (left & 2u) { st16(q, vv & 0xFFFFu); q += 2; }
      if (left & 1u) *q = (uint8_t)v;
    }
    op += total;
    const uint32_t top = (F == ~0ull) ? 64u : (uint32_t)__builtin_clzll(~F);   // flagged lanes at the top of the window
    carry = (top + ((top == 64u) ? carry : 0u)) & 1u;
    pendSym = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
  }
  if (!err && (op != want || carry != 0u)) err = true;
  if (err && lane == 0u && status) atomicOr(status, RLE8M_ERR_STREAM);
}


// ------------------------------------------------------------------------------------------------------------------
// rle8m encoder (the reference's rle8m_compress is a CPU function, src/rle8_low_entropy_cpu.c:131-191; this is its GPU twin, so that
// an rle8m stream can be produced and consumed device resident).  Four kernels:
//   k_rle8m_stats   run statistics over the WHOLE input (get_compress_info, :254-338): every maximal run adds its length to prob[sym]
//                   and length / 255 + 1 to pcount[sym]; one lane per section counts the runs that START in its section (and follows
//                   them across section ends), LDS histograms per workgroup, one global atomic per symbol and workgroup
//   k_rle8m_info    which symbols carry a repeat code (prob / pcount >= 2), the symbols ordered by pcount (ties: lower symbol first;
//                   unused symbols last, ascending), the info bytes of the stream header (:441-472)
//   k_rle8m_encode  one lane per section (compress_with_info, :474-543) into a staging slot of twice the section size
//   k_rle8m_place   compaction into the stream behind the header + the header itself + the capacity rule of the reference (a section
//                   must fit into what is left of the output when its turn comes, :476) -> status bit, stream size 0
// byte reader over global memory through a 16-byte register window.  `limit` = readable bytes behind `base` (>= 16 on the fast path:
// the last window of the buffer is read at limit - 16 and shifted, so there is no byte loop anywhere near the hot path)
struct ByteWindow
{
  const uint8_t *base;
  uint32_t limit;
  uint32_t wpos = 0xFFFFFF00u;
  uint64_t lo = 0, hi = 0;
  __device__ __forceinline__ uint32_t get(uint32_t pos)
  {
    if (pos - wpos >= 16u)
    {
      wpos = pos;
      if (limit >= 16u)
      {
        const uint32_t at = (pos + 16u <= limit) ? pos : limit - 16u;
        const uint32_t sh = 8u * (pos - at);                              // 0, or 8 .. 120 bits for the last window
        const uint64_t a = ld64(base + at), b = ld64(base + at + 8);
        lo = (sh == 0u) ? a : ((sh < 64u) ? ((a >> sh) | (b << (64u - sh))) : (b >> (sh - 64u)));
        hi = (sh < 64u) ? (b >> sh) : 0ull;
      }
      else
      {
        lo = 0; hi = 0;
        for (uint32_t j = 0; j < 16u && pos + j < limit; j++)
        {
          if (j < 8u) lo |= (uint64_t)base[pos + j] << (8u * j); else hi |= (uint64_t)base[pos + j] << (8u * (j - 8u));
        }
      }
    }
    const uint32_t d = pos - wpos;
    const uint64_t v = (d & 8u) ? hi : lo;
    return (uint32_t)(v >> (8u * (d & 7u))) & 0xFFu;
  }
};

struct Rle8mTables                          // device scratch shared by the kernels (in the workspace)
{
  uint32_t prob[256], pcount[256];
  uint8_t rle[256], order[256];
  uint32_t rleBits[8];                      // rle[] as a bitmap (bit b of word b / 32), what the encode kernels read
  uint32_t listed;                          // symbols written to the header (1..255; 256 are written as 255, sic)
  uint32_t headerSize;
};

// (`sections` here is any partition of the input into pieces: the statistics are over the whole input)
__global__ __launch_bounds__(64) void k_rle8m_stats(const uint8_t *__restrict__ d, uint32_t n, uint32_t sections, Rle8mTables *__restrict__ t, uint32_t maxLen)
{
  __shared__ uint32_t prob[256], pcount[256];
  const uint32_t lane = threadIdx.x;
  for (uint32_t k = lane; k < 256u; k += 64u) { prob[k] = 0; pcount[k] = 0; }
  __syncthreads();

  const uint32_t k = blockIdx.x * 64u + lane;
  if (k < sections)
  {
    const uint32_t ss = n / sections;
    const uint64_t a = (uint64_t)k * ss, b = (k + 1u < sections) ? a + ss : (uint64_t)n;
    // one byte per loop trip and lane (a loop per run inside a loop over the runs makes every lane wait for the longest run of its
    // wave at every run end: 10x slower on zero-dominated data)
    ByteWindow w{ d, n };
    bool skipping = k > 0u;                                               // the run that came in from the section before is not mine
    const uint32_t prev = skipping ? (uint32_t)d[a - 1] : 0u;
    uint32_t sym = 0xFFFFFFFFu, len = 0;
    for (uint32_t i = (uint32_t)a; i < n; i++)
    {
      const uint32_t c = w.get(i);
      if (skipping)
      {
        if (i < (uint32_t)b && c == prev) continue;
        skipping = false;
      }
      if (c == sym) { len++; continue; }                                  // the run goes on (also across section ends)
      if (sym != 0xFFFFFFFFu)
      {
        atomicAdd(&prob[sym], len);
        atomicAdd(&pcount[sym], len / maxLen + 1u);
        sym = 0xFFFFFFFFu;
      }
      if (i >= (uint32_t)b) break;                                        // a run that starts behind my section is not mine
      sym = c; len = 1;
    }
    if (sym != 0xFFFFFFFFu)                                               // the run that reaches the end of the input counts once, however
    {                                                                     // long it is (rle8_low_entropy_cpu.c: the statement behind the loop)
      atomicAdd(&prob[sym], len);
      atomicAdd(&pcount[sym], 1u);
    }
  }
  __syncthreads();
  for (uint32_t s2 = lane; s2 < 256u; s2 += 64u)
  {
    if (prob[s2]) atomicAdd(&t->prob[s2], prob[s2]);
    if (pcount[s2]) atomicAdd(&t->pcount[s2], pcount[s2]);
  }
}

// The same statistics, position-parallel: persistent waves take 4 KiB pieces of the input.  A piece goes to LDS, 64 positions per lane
// become equality bits (d[i] == d[i + 1]), run starts are the positions whose predecessor differs, and every lane adds the runs that
// START in its 64 positions -- length = 1 + the set bits from the start on (followed through the piece and, for the rare run that
// leaves it, through global memory) -- to LDS histograms; one flush of the histograms per wave.  (k_rle8m_stats above walks a byte
// per lane and trip: 1.87 ms per GiB, as long as the encode kernel; kept for HSRLE_RLE8M_STATS=1 A/B runs.)
constexpr uint32_t kRle8mStatsPieces = 15;    // 4 KiB pieces per wave at most (15 * 4096 < 65536: the packed LDS counters)
__global__ __launch_bounds__(64) void k_rle8m_stats_wave(const uint8_t *__restrict__ d, uint32_t n, Rle8mTables *__restrict__ t, uint32_t maxLen, uint32_t *__restrict__ lastB4,
                                                          uint32_t *__restrict__ firstB4)
{
  constexpr uint32_t P = 4096u;
  __shared__ __attribute__((aligned(16))) uint8_t bytes[P + 16u];
  __shared__ uint64_t eqw[P / 64u + 1u];
  __shared__ uint32_t pk[256];
  const uint32_t lane = threadIdx.x;
  for (uint32_t k = lane; k < 256u; k += 64u) pk[k] = 0;
  const uint32_t pieces = (uint32_t)(((uint64_t)n + P - 1u) / P);
  for (uint32_t piece = blockIdx.x; piece < pieces; piece += gridDim.x)
  {
    const uint32_t a = piece * P;
    const uint32_t len = (n - a < P) ? n - a : P;
    __syncthreads();                                                      // (the piece before is done with the buffers)
    // piece + the byte behind it (zero filled beyond the input: masked below)
    for (uint32_t q = lane * 16u; q < P + 16u; q += 64u * 16u)
    {
      u32x4 v = u32x4{ 0, 0, 0, 0 };
      if ((uint64_t)a + q + 16u <= (uint64_t)n) v = ld128(d + a + q);
      else if ((uint64_t)a + q < (uint64_t)n)
      {
        uint32_t tt[4] = { 0, 0, 0, 0 };
        for (uint32_t k = 0; (uint64_t)a + q + k < (uint64_t)n && k < 16u; k++) tt[k >> 2] |= (uint32_t)d[a + q + k] << (8u * (k & 3u));
        v = u32x4{ tt[0], tt[1], tt[2], tt[3] };
      }
      lds_st128(bytes + q, v);
    }
    const uint32_t prevByte = (a > 0u) ? (uint32_t)d[a - 1u] : 0x100u;    // (0x100: the input's first byte starts a run)
    __syncthreads();
    {
      // equality bits of my 64 positions: bit i = (d[a + i] == d[a + i + 1]) and a + i + 1 < n
      const uint32_t w = lane;
      uint64_t e64 = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++)
      {
        const u32x4 x = lds_ld128(bytes + w * 64u + j * 16u);
        const uint32_t x4 = lds_ld32(bytes + w * 64u + j * 16u + 16u);
        const uint32_t zm16 = zero_mask16(x.x ^ alignbyte(x.y, x.x, 1), x.y ^ alignbyte(x.z, x.y, 1), x.z ^ alignbyte(x.w, x.z, 1), x.w ^ alignbyte(x4, x.w, 1));
        e64 |= (uint64_t)zm16 << (16u * j);
      }
      const uint32_t base = w * 64u;
      const uint64_t left = (uint64_t)n - 1u - a;                         // positions a + i with a + i + 1 < n: i < left
      const uint32_t valid = (left > base) ? (uint32_t)((left - base < 64u) ? left - base : 64u) : 0u;
      if (valid < 64u) e64 &= (1ull << valid) - 1ull;
      eqw[w] = e64;
    }
    __syncthreads();
    {
      const uint32_t w = lane;
      const uint32_t base = w * 64u;
      const uint64_t m = eqw[w];
      // run starts among my positions: the predecessor differs (position 0 of the piece: the byte in front of the piece)
      const uint64_t prevEq = (w > 0u) ? eqw[w - 1u] >> 63 : ((prevByte == (uint32_t)bytes[0]) ? 1ull : 0ull);
      uint64_t starts = ~((m << 1) | prevEq);
      const uint32_t mine = (len > base) ? ((len - base < 64u) ? len - base : 64u) : 0u;
      if (mine < 64u) starts &= (mine == 0u) ? 0ull : ((1ull << mine) - 1ull);
      // Round 6: a run is counted PIECE BY PIECE -- the part of it inside this piece, clipped at the piece's end -- instead of being followed through global
      // memory by the one lane that holds its start (all-equal input: one lane walked the whole buffer).  prob is a byte count; pcount = runs + the token
      // boundaries inside runs (start + j maxLen, j >= 1): a run that STARTS here adds 1 + (its bytes in this piece) / maxLen; the boundaries of the run that
      // ENTERS the piece from the left depend on where that run started: the piece's first and last run start go to firstB4 / lastB4, k_le_scan_last turns them
      // into every piece's run start, k_le_stats_fixup adds what is missing (and takes the final run's boundaries out again: it counts once, :295-296).
      uint32_t ownFirst = (starts != 0ull) ? a + base + (uint32_t)__builtin_ctzll(starts) : 0xFFFFFFFFu;
      uint32_t ownLast = (starts != 0ull) ? a + base + 63u - (uint32_t)__builtin_clzll(starts) : 0u;
      while (starts != 0ull)
      {
        const uint32_t p = (uint32_t)__builtin_ctzll(starts);
        starts &= starts - 1ull;
        // set bits from p on
        const uint64_t sh = m >> p;
        uint32_t ones = (p == 0u && m == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~sh | ((p == 0u) ? 0ull : (1ull << (64u - p))));
        uint32_t total = ones;
        if (ones >= 64u - p)
        {
          // through the following words of the piece
          uint32_t w2 = w + 1u;
          while (w2 < P / 64u)
          {
            const uint64_t mm = eqw[w2];
            if (mm == ~0ull) { total += 64u; w2++; }
            else { total += (uint32_t)__builtin_ctzll(~mm); break; }
          }
        }
        const uint32_t endIn = (base + p + 1u + total < len) ? base + p + 1u + total : len;   // the run's end inside the piece
        const uint32_t Lc = endIn - (base + p);
        const uint32_t sy = bytes[base + p];
        atomicAdd(&pk[sy], ((1u + Lc / maxLen) << 16) | Lc);               // (a wave sees at most kRle8mStatsPieces * 4096 bytes: neither half overflows)
      }
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1)
      {
        const uint32_t f = (uint32_t)__shfl_xor((int)ownFirst, dd, 64), l = (uint32_t)__shfl_xor((int)ownLast, dd, 64);
        ownFirst = f < ownFirst ? f : ownFirst;
        ownLast = l > ownLast ? l : ownLast;
      }
      if (lane == 0u)
      {
        lastB4[piece] = ownLast;
        firstB4[piece] = ownFirst;
        // the bytes of the run that enters from the left (its token boundaries: k_le_stats_fixup)
        const uint32_t f = ownFirst == 0xFFFFFFFFu ? len : ownFirst - a;
        if (f != 0u) atomicAdd(&pk[bytes[0]], f);
      }
    }
  }
  __syncthreads();
  for (uint32_t s2 = lane; s2 < 256u; s2 += 64u)
  {
    const uint32_t v = pk[s2];
    if (v != 0u) { atomicAdd(&t->prob[s2], v & 0xFFFFu); atomicAdd(&t->pcount[s2], v >> 16); }
  }
}

// onlyMax: the rule of the *_only_max_frequency encoders (rle8_low_entropy_cpu.c:340-439, rle8_low_entropy_short_cpu.c:622-720): ONE symbol
// carries repeat codes -- the first one with the most bytes saved (prob - 2 * pcount, among the symbols with prob / pcount > 2), if any
__global__ __launch_bounds__(256) void k_rle8m_info(Rle8mTables *__restrict__ t, uint32_t sections, uint8_t *__restrict__ out, uint32_t onlyMax)
{
  __shared__ uint32_t pc[256];
  __shared__ uint8_t order[256], flag[256];
  __shared__ uint64_t saved[256];
  const uint32_t i = threadIdx.x;
  const uint32_t p = t->prob[i], c = t->pcount[i];
  pc[i] = c;
  flag[i] = (!onlyMax && c > 0u && p / c >= 2u) ? 1 : 0;
  saved[i] = (c > 0u && p / c > 2u) ? (uint64_t)p - 2ull * (uint64_t)c : 0ull;   // (size_t arithmetic of uint32 operands in the reference: prob - pcount * 2 with the product taken in 32 bits)
  __syncthreads();
  if (onlyMax && i == 0u)
  {
    uint64_t best = 0; uint32_t at = 0;
    for (uint32_t j = 0; j < 256u; j++)
      if (saved[j] > best) { best = saved[j]; at = j; }
    if (best > 0ull) flag[at] = 1;
  }
  __syncthreads();
  uint32_t used = 0, rank = 0, unusedBelow = 0;
  for (uint32_t j = 0; j < 256u; j++)
  {
    const uint32_t cj = pc[j];
    used += cj != 0u;
    if (c != 0u) rank += (cj > c) || (cj == c && j < i);
    else unusedBelow += (cj == 0u && j < i);
  }
  order[(c != 0u) ? rank : used + unusedBelow] = (uint8_t)i;
  __syncthreads();
  t->rle[i] = flag[i];
  t->order[i] = order[i];
  const uint32_t listed = (used & 0xFFu) ? (used & 0xFFu) : 255u;        // symbolCount is a uint8: 256 -> 0 -> 255 symbols written (sic)
  const uint32_t info = 12u + 4u * (sections - 1u);
  if (i < 32u)
  {
    uint32_t v = 0;
    for (uint32_t j = 0; j < 8u; j++) v |= (uint32_t)flag[i * 8u + j] << j;
    out[info + i] = (uint8_t)v;
  }
  if (i < 8u)
  {
    uint32_t v = 0;
    for (uint32_t j = 0; j < 32u; j++) v |= (uint32_t)flag[i * 32u + j] << j;
    t->rleBits[i] = v;
  }
  if (i == 0u)
  {
    out[info + 32u] = (uint8_t)used;
    t->listed = listed;
    t->headerSize = info + 33u + listed;
  }
  if (i < listed) out[info + 33u + i] = order[i];
}

__global__ __launch_bounds__(64) void k_rle8m_encode(const uint8_t *__restrict__ d, uint32_t n, uint32_t sections, const Rle8mTables *__restrict__ t,
                                                      uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes, uint32_t maxLen)
{
  __shared__ uint32_t rleBits[8];
  __shared__ __attribute__((aligned(4))) uint8_t order[256];
  __shared__ __attribute__((aligned(16))) uint8_t ring[64 * kLaneRingStride];
  const uint32_t lane = threadIdx.x;
  if (lane < 8u) rleBits[lane] = t->rleBits[lane];
  reinterpret_cast<uint32_t *>(order)[lane] = reinterpret_cast<const uint32_t *>(t->order)[lane];   // 64 lanes x 4 bytes
  __syncthreads();

  const uint32_t k = blockIdx.x * 64u + lane;
  if (k >= sections)
    return;
  const uint32_t ss = n / sections;
  const uint8_t *const p = d + (uint64_t)k * ss;
  const uint32_t len = (k + 1u < sections) ? ss : n - ss * (sections - 1u);
  uint8_t *const o = slots + (uint64_t)k * slotStride;
  const uint32_t target = (len >= 256u) ? len - 256u : 0u;

  uint64_t alo = 0, ahi = 0;
  uint32_t op = 0;
  auto put = [&](uint32_t b) {
    const uint32_t sh = 8u * (op & 7u);
    if (op & 8u) ahi |= (uint64_t)b << sh; else alo |= (uint64_t)b << sh;
    op++;
    if ((op & 15u) == 0u) { st64(o + op - 16u, alo); st64(o + op - 8u, ahi); alo = 0; ahi = 0; }
  };

  // one byte per loop trip and lane, as in k_rle8m_stats: either the next symbol is emitted, or the repeat scan behind a flagged one
  // looks at one more byte; input through the LaneRing (the read cursor moves by at most one byte per trip: top-up every 64 trips).
  // Reading behind the section (inside the input) is harmless.
  LaneRing in{ p, n - k * ss, ring + lane * kLaneRingStride, 0u };
  auto in_byte = [&](uint32_t pos) -> uint32_t { return in.get(pos); };

  uint32_t i = 0, j = 0, count = 0, range = 0, runSym = 0;
  bool scanning = false;
  for (uint32_t trip = 0;; trip++)
  {
    if ((trip & 63u) == 0u) in.topup<5>(scanning ? i + j : i);
    if (!scanning)
    {
      if (i >= len) break;
      const uint32_t b = in_byte(i);
      put(b);
      if ((rleBits[b >> 5] >> (b & 31u)) & 1u)
      {
        const uint32_t left = len - i - 1u;
        range = (i < target) ? maxLen : (left < maxLen ? left : maxLen);   // :497 / :521 (maxLen 255; the Short form: 32)
        runSym = b; count = 0; j = 1; scanning = true;
      }
      else
        i++;
    }
    else if (j < range && in_byte(i + j) == runSym) { count++; j++; }
    else
    {
      i += j;
      put(order[count]);
      scanning = false;
    }
  }
  for (uint32_t j = op & ~15u; j < op; j++)
    o[j] = (uint8_t)(((j & 8u) ? ahi : alo) >> (8u * (j & 7u)));
  sizes[k] = op;
}

// k_rle8m_encode with one WAVE per section (few, large sections).  The token grammar is position-parallel: a symbol without a repeat
// code is its own token; within a maximal run of a flagged symbol -- cut at the section start and before the section's last byte, which
// the reference's scan never reaches (range = left, j < range) -- tokens start every 255 bytes (count <= 254).  Every lane looks at one
// byte, decides whether a token ENDS there (no token starts inside another, so emitting at the end keeps the order) and the count is the
// distance to the run start modulo 255; run starts come from the ballot of the run breaks, offsets from a wave scan.
__global__ __launch_bounds__(64) void k_rle8m_encode_wave(const uint8_t *__restrict__ d, uint32_t n, uint32_t sections, const Rle8mTables *__restrict__ t,
                                                           uint8_t *__restrict__ slots, uint32_t slotStride, uint32_t *__restrict__ sizes, uint32_t maxLen)
{
  __shared__ uint32_t rleBits[8];
  __shared__ __attribute__((aligned(4))) uint8_t order[256];
  const uint32_t lane = threadIdx.x;
  if (lane < 8u) rleBits[lane] = t->rleBits[lane];
  reinterpret_cast<uint32_t *>(order)[lane] = reinterpret_cast<const uint32_t *>(t->order)[lane];   // 64 lanes x 4 bytes
  __syncthreads();

  const uint32_t k = blockIdx.x;
  if (k >= sections)
    return;
  const uint32_t ss = n / sections;
  const uint8_t *const p = d + (uint64_t)k * ss;
  const uint32_t len = (k + 1u < sections) ? ss : n - ss * (sections - 1u);
  uint8_t *const o = slots + (uint64_t)k * slotStride;

  uint32_t op = 0, carryPrev = 0, carryRunStart = 0;
  uint32_t bNext = (lane < len) ? (uint32_t)p[lane] : 0u;
  for (uint32_t ip = 0; ip < len; ip += 64u)
  {
    const uint32_t pos = ip + lane;
    const bool valid = pos < len;
    const uint32_t b = bNext;
    bNext = (pos + 64u < len) ? (uint32_t)p[pos + 64u] : 0u;
    const bool fl = valid && ((rleBits[b >> 5] >> (b & 31u)) & 1u);
    const uint32_t up = (uint32_t)__shfl_up((int)b, 1), down = (uint32_t)__shfl_down((int)b, 1), first = (uint32_t)__shfl((int)bNext, 0);
    const uint32_t prevb = (lane == 0u) ? carryPrev : up;
    const uint32_t nextb = (lane == 63u) ? first : down;
    const bool brk = valid && (pos == 0u || b != prevb || pos == len - 1u);
    const uint64_t B = __builtin_amdgcn_ballot_w64(brk);
    const uint64_t mine = B & (~0ull >> (63u - lane));                   // breaks at or below this lane
    const uint32_t runStart = mine ? ip + (63u - (uint32_t)__builtin_clzll(mine)) : carryRunStart;
    const uint32_t rel = pos - runStart;
    const bool nextBreaks = pos + 1u >= len - 1u || nextb != b;           // (the last byte of the section is a break)
    const bool isEnd = valid && (!fl || pos == len - 1u || nextBreaks || (rel + 1u) % maxLen == 0u);
    // a token is one byte, or two with a repeat code: the offsets are population counts of the lanes below (no scan needed)
    const uint64_t E = __builtin_amdgcn_ballot_w64(isEnd), E2 = __builtin_amdgcn_ballot_w64(isEnd && fl);
    const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(E >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E, 0u)) +
                         __builtin_amdgcn_mbcnt_hi((uint32_t)(E2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E2, 0u));
    if (isEnd)
    {
      uint8_t *q = o + op + off;
      q[0] = (uint8_t)b;
      if (fl) q[1] = order[rel % maxLen];
    }
    op += (uint32_t)__builtin_popcountll(E) + (uint32_t)__builtin_popcountll(E2);
    carryPrev = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    carryRunStart = (uint32_t)__builtin_amdgcn_readlane((int)runStart, 63);
  }
  if (lane == 0u) sizes[k] = op;
}

// one wave per section: copy its staged stream behind the header; wave 0 of every workgroup of 4 also writes header fields
__global__ __launch_bounds__(256) void k_rle8m_place(const uint8_t *__restrict__ slots, uint32_t slotStride, const uint64_t *__restrict__ offsets, const Rle8mTables *__restrict__ t,
                                                     uint8_t *__restrict__ out, uint64_t outCapacity, uint32_t n, uint32_t sections, uint32_t *__restrict__ status)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (b >= sections)
    return;
  const uint32_t H = t->headerSize;
  const uint64_t off = offsets[b];
  const uint32_t size = (uint32_t)(offsets[b + 1] - off);
  const uint32_t ss = n / sections;
  const uint32_t len = (b + 1u < sections) ? ss : n - ss * (sections - 1u);
  // the reference gives up when a section is larger than what is left of the output at its turn (compress_with_info: outSize < inSize).
  // It does not look at the section's STREAM, which can be twice as long (every byte a flagged symbol with a zero repeat code): there
  // the reference writes behind its caller's buffer.  That is not reproduced: a stream that does not fit is a failure here.
  const bool fits = (uint64_t)H + off <= outCapacity && outCapacity - ((uint64_t)H + off) >= (uint64_t)len && (uint64_t)H + off + size <= outCapacity &&
                    (uint64_t)H + off + size <= 0xFFFFFFFFull && len != 0u;
  if (!fits)
  {
    if (lane == 0u && status) atomicOr(status, RLE8M_ERR_STREAM);
    return;
  }
  const uint8_t *src = slots + (uint64_t)b * slotStride;
  uint8_t *dst = out + H + off;
  for (uint32_t k = lane * 16u; k + 16u <= size; k += 64u * 16u) st128(dst + k, ld128(src + k));
  const uint32_t body = size & ~15u;
  if (lane < size - body) dst[body + lane] = src[body + lane];
  if (lane == 0u)
  {
    if (b + 1u < sections) st32(out + 12u + 4u * b, (uint32_t)(H + off + size));   // end offset of this section
    else st32(out, (uint32_t)(H + off + size));                                   // total size
    if (b == 0u) { st32(out + 4, n); st32(out + 8, sections); }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The UNSECTIONED low-entropy streams (rle8_low_entropy[_short]_compress[_only_max_frequency] / _decompress,
// src/rle8_low_entropy_cpu.c:6-124, src/rle8_low_entropy_short_cpu.c:16-124) by many waves (round 4).  The format offers no independent
// units -- [u32 size][u32 inSize][info][ONE stream] -- but both directions can be cut without changing a byte:
//   encode  the sequential encoder (compress_with_info, :474-543) stands at every run boundary of the input sooner or later (a run of a
//           flagged symbol is consumed whole, 255 / 32 bytes per token counted from the run's start; everything else byte by byte).  So the
//           input is cut at the first run boundary at or behind every G-th byte (k_le_cuts), the pieces are encoded independently by the
//           wave kernel's grammar -- the end-of-input rule (the last byte is never part of a repeat count, :521) only in the piece that
//           ends the input -- and the piece streams are the stream, back to back (k_le_place).
//   decode  which stream bytes are symbols and which are repeat codes follows from the byte VALUES alone: a byte is a code iff the stretch
//           of flagged-valued bytes right in front of it has odd length (k_rle8m_decode_wave).  A piece may therefore start at any stream
//           byte once the parity in front of it is known (k_le_carry: a short backward scan); a dry pass of the decoder gives every
//           piece's output size, a scan the output positions, the second pass writes.
constexpr uint32_t kLeNone = 0xFFFFFFFFu;
constexpr uint32_t kLeCarryLimit = 1u << 16;    // flagged-valued stretch in front of a piece longer than this: the caller falls back to one wave

// runStart4[k] = the last run start in front of 4 KiB piece k = max(lastB4[0 .. k - 1]) (0: the run that covers the piece's first byte starts at position 0).
// One workgroup walks the table 4 096 entries at a time (262 144 pieces per GiB).  Round 6.
__global__ __launch_bounds__(1024) void k_le_scan_last(const uint32_t *__restrict__ lastB4, uint32_t P, uint32_t *__restrict__ runStart4)
{
  __shared__ uint32_t waveMax[16];
  __shared__ uint32_t carryS;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid == 0u) carryS = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < P; base += 4096u)
  {
    const uint32_t k0 = base + 4u * tid;
    uint32_t own[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) own[j] = (k0 + j < P) ? lastB4[k0 + j] : 0u;
    const uint32_t m01 = own[0] > own[1] ? own[0] : own[1], m23 = own[2] > own[3] ? own[2] : own[3];
    uint32_t v = m01 > m23 ? m01 : m23;                                    // inclusive max over the wave's threads
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1)
    {
      const uint32_t o = (uint32_t)__shfl_up((int)v, dd, 64);
      if ((int)lane >= dd && o > v) v = o;
    }
    if (lane == 63u) waveMax[wv] = v;
    __syncthreads();
    uint32_t front = carryS;                                              // everything in front of this wave
    for (uint32_t w = 0; w < wv; w++) front = waveMax[w] > front ? waveMax[w] : front;
    const uint32_t exclInWave = (uint32_t)__shfl_up((int)v, 1, 64);
    uint32_t run = front;
    if (lane != 0u && exclInWave > run) run = exclInWave;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
    {
      if (k0 + j < P) runStart4[k0 + j] = run;
      run = own[j] > run ? own[j] : run;
    }
    __syncthreads();
    if (tid == 1023u) carryS = run;
    __syncthreads();
  }
}

// what k_rle8m_stats_wave could not know: the token boundaries of the run that enters a 4 KiB piece from the left (it started at runStart4[piece]) -- and the
// final run of the input counts ONCE (rle8_low_entropy_cpu.c:295-296), so its boundaries are taken out again.  One lane per piece; nearly every lane adds nothing.
__global__ __launch_bounds__(256) void k_le_stats_fixup(const uint8_t *__restrict__ d, uint32_t n, uint32_t P, const uint32_t *__restrict__ lastB4, const uint32_t *__restrict__ firstB4,
                                                        const uint32_t *__restrict__ runStart4, Rle8mTables *__restrict__ t, uint32_t maxLen)
{
  // (collected per workgroup first: with the Short form's 32-byte tokens every fourth piece of a video-shaped buffer adds to pcount[0] -- 65 536 atomics on one
  //  address per GiB cost more than the statistics pass)
  __shared__ uint32_t add[256];
  add[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k < P)
  {
    const uint64_t a = (uint64_t)k * 4096u;
    const uint64_t b = (a + 4096u < (uint64_t)n) ? a + 4096u : (uint64_t)n;
    const uint32_t fb = firstB4[k];
    if (k != 0u && (uint64_t)fb != a)
    {
      const uint64_t rs = runStart4[k], f = fb == 0xFFFFFFFFu ? b : (uint64_t)fb;
      const uint32_t cnt = (uint32_t)((f - rs) / maxLen - (a - rs) / maxLen);
      if (cnt != 0u) atomicAdd(&add[d[a]], cnt);
    }
    if (k == P - 1u)
    {
      const uint64_t rsF = lastB4[k] > runStart4[k] ? lastB4[k] : runStart4[k];
      const uint32_t sub = (uint32_t)(((uint64_t)n - rsF) / maxLen);
      if (sub != 0u) atomicSub(&t->pcount[d[n - 1u]], sub);
    }
  }
  __syncthreads();
  const uint32_t v = add[threadIdx.x];
  if (v != 0u) atomicAdd(&t->pcount[threadIdx.x], v);
}

// cuts[k], k >= 1: where piece k begins -- the first position at or behind k G at which the sequential encoder (compress_with_info, :474-543) starts a token
// whatever came before: a run boundary; any position inside a run of a symbol that is NOT flagged (every byte is a token of its own); inside a run of a flagged
// symbol the positions runStart + j maxLen (the tokens of a long run are counted from its start).  Within maxLen bytes there is always one, so all-equal input
// -- the codec's best case -- gets a piece per G bytes too.  Cuts inside a run keep clear of the input's last 256 bytes (the end-of-input rule, :515-540, belongs
// to the piece that ends the input); kLeNone: no cut (the piece is part of its predecessor).
__global__ __launch_bounds__(64) void k_le_cuts(const uint8_t *__restrict__ d, uint32_t n, uint32_t G, uint32_t P, const uint32_t *__restrict__ runStart, const Rle8mTables *__restrict__ t,
                                                uint32_t maxLen, uint32_t *__restrict__ cuts)
{
  const uint32_t k = blockIdx.x * 64u + threadIdx.x;
  if (k >= P) return;
  if (k == 0u) { cuts[0] = 0u; return; }
  const uint64_t a = (uint64_t)k * G;
  const uint64_t e = (a + G < (uint64_t)n) ? a + G : (uint64_t)n;
  ByteWindow w{ d, n };
  uint32_t prev = w.get((uint32_t)a - 1u);
  const uint32_t c0 = w.get((uint32_t)a);
  uint32_t cut = kLeNone;
  if (c0 != prev) cut = (uint32_t)a;
  else
  {
    const bool flagged = ((t->rleBits[c0 >> 5] >> (c0 & 31u)) & 1u) != 0u;
    const uint64_t clear = n > 256u ? (uint64_t)n - 256u : 0ull;         // cuts inside a run: in front of this position only
    const uint64_t rs = runStart[a >> 12];                                // (a table entry per 4 KiB: k_le_scan_last)
    const uint64_t grid = flagged ? rs + ((a - rs + maxLen - 1u) / maxLen) * (uint64_t)maxLen : a;   // the first token start at or behind a if the run goes on
    for (uint64_t i = a; i < e; i++)
    {
      if (i == grid) { if (i < clear) cut = (uint32_t)i; break; }
      const uint32_t c = w.get((uint32_t)i);
      if (c != prev) { cut = (uint32_t)i; break; }
      prev = c;
    }
    if (cut == kLeNone && grid >= clear)
    {
      // (near the end of the input only run boundaries are cuts: look for one in the rest of the piece)
      for (uint64_t i = a + 1u; i < e; i++)
      {
        const uint32_t c = w.get((uint32_t)i);
        if (c != c0) { cut = (uint32_t)i; break; }
      }
    }
  }
  cuts[k] = cut;
}

// k_rle8m_encode_wave over the pieces of ONE stream: piece k = [cuts[k], next cut that exists) -- both run boundaries, so no token crosses
// them and the distance to the run start is the sequential encoder's -- into slots + 2 * cuts[k] (a stream is at most twice its input)
__global__ __launch_bounds__(64) void k_le_encode_wave(const uint8_t *__restrict__ d, uint32_t n, const uint32_t *__restrict__ cuts, uint32_t P, const Rle8mTables *__restrict__ t,
                                                        uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes, uint32_t maxLen)
{
  __shared__ uint32_t rleBits[8];
  __shared__ __attribute__((aligned(4))) uint8_t order[256];
  const uint32_t lane = threadIdx.x;
  if (lane < 8u) rleBits[lane] = t->rleBits[lane];
  reinterpret_cast<uint32_t *>(order)[lane] = reinterpret_cast<const uint32_t *>(t->order)[lane];
  __syncthreads();

  const uint32_t k = blockIdx.x;
  if (k >= P) return;
  const uint32_t a = cuts[k];
  if (a == kLeNone) { if (lane == 0u) sizes[k] = 0u; return; }
  uint32_t b = n;
  for (uint32_t j = k + 1u; j < P; j++) { const uint32_t c = cuts[j]; if (c != kLeNone) { b = c; break; } }
  const uint8_t *const p = d + a;
  const uint32_t len = b - a;
  const bool tail = b == n;                                               // the piece that ends the input: its last byte is never part of a repeat count
  uint8_t *const o = slots + 2ull * (uint64_t)a;

  uint32_t op = 0, carryPrev = 0, carryRunStart = 0;
  uint32_t bNext = (lane < len) ? (uint32_t)p[lane] : 0u;
  for (uint32_t ip = 0; ip < len; ip += 64u)
  {
    const uint32_t pos = ip + lane;
    const bool valid = pos < len;
    const uint32_t bb = bNext;
    bNext = (pos + 64u < len) ? (uint32_t)p[pos + 64u] : 0u;
    const bool fl = valid && ((rleBits[bb >> 5] >> (bb & 31u)) & 1u);
    const uint32_t up = (uint32_t)__shfl_up((int)bb, 1), down = (uint32_t)__shfl_down((int)bb, 1), first = (uint32_t)__shfl((int)bNext, 0);
    const uint32_t prevb = (lane == 0u) ? carryPrev : up;
    const uint32_t nextb = (lane == 63u) ? first : down;
    const bool lastOfInput = tail && pos == len - 1u;
    const bool brk = valid && (pos == 0u || bb != prevb || lastOfInput);
    const uint64_t B = __builtin_amdgcn_ballot_w64(brk);
    const uint64_t mine = B & (~0ull >> (63u - lane));
    const uint32_t runStart = mine ? ip + (63u - (uint32_t)__builtin_clzll(mine)) : carryRunStart;
    const uint32_t rel = pos - runStart;
    // the next byte starts a new token: the piece ends here (its successor differs: a cut), the input's last byte stands alone, or the byte differs
    const bool nextBreaks = pos + 1u >= len || (tail && pos + 1u >= len - 1u) || nextb != bb;
    const bool isEnd = valid && (!fl || lastOfInput || nextBreaks || (rel + 1u) % maxLen == 0u);
    const uint64_t E = __builtin_amdgcn_ballot_w64(isEnd), E2 = __builtin_amdgcn_ballot_w64(isEnd && fl);
    const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(E >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E, 0u)) +
                         __builtin_amdgcn_mbcnt_hi((uint32_t)(E2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E2, 0u));
    if (isEnd)
    {
      uint8_t *q = o + op + off;
      q[0] = (uint8_t)bb;
      if (fl) q[1] = order[rel % maxLen];
    }
    op += (uint32_t)__builtin_popcountll(E) + (uint32_t)__builtin_popcountll(E2);
    carryPrev = (uint32_t)__builtin_amdgcn_readlane((int)bb, 63);
    carryRunStart = (uint32_t)__builtin_amdgcn_readlane((int)runStart, 63);
  }
  if (lane == 0u) sizes[k] = op;
}

// piece streams -> the stream behind [u32 size][u32 inSize][info]; one wave per piece.  The info bytes were written by k_rle8m_info for an
// rle8m header of one section (at byte 12): they move to byte 8 here (infoFrom = 12).
__global__ __launch_bounds__(256) void k_le_place(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ cuts, const uint64_t *__restrict__ offsets, const Rle8mTables *__restrict__ t,
                                                  uint8_t *__restrict__ out, uint64_t outCapacity, uint32_t n, uint32_t P, uint32_t *__restrict__ status)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (k >= P) return;
  const uint32_t H = t->headerSize - 4u;                                  // (the tables count an rle8m header of one section: 4 bytes more)
  const uint64_t off = offsets[k], total = offsets[P];
  const uint32_t size = (uint32_t)(offsets[k + 1] - off);
  if ((uint64_t)H + total > outCapacity || (uint64_t)H + total > 0xFFFFFFFFull)
  {
    if (k == 0u && lane == 0u && status) atomicOr(status, RLE8M_ERR_STREAM);
    return;
  }
  if (k == 0u && lane == 0u) { st32(out, (uint32_t)(H + total)); st32(out + 4, n); }
  if (size == 0u) return;
  const uint8_t *src = slots + 2ull * (uint64_t)cuts[k];
  uint8_t *dst = out + H + off;
  for (uint32_t j = lane * 16u; j + 16u <= size; j += 64u * 16u) st128(dst + j, ld128(src + j));
  const uint32_t body = size & ~15u;
  if (lane < size - body) dst[body + lane] = src[body + lane];
}

// the info bytes of an unsectioned stream: k_rle8m_info writes them at byte 12 of ITS output (an rle8m header of one section); here they go
// to byte 8 of the real one.  One workgroup.
__global__ __launch_bounds__(64) void k_le_move_info(const uint8_t *__restrict__ tmpInfo, const Rle8mTables *__restrict__ t, uint8_t *__restrict__ out)
{
  const uint32_t len = t->headerSize - 12u;                               // 33 + listed
  for (uint32_t j = threadIdx.x; j < len; j += 64u) out[8u + j] = tmpInfo[12u + j];
}

// header of an unsectioned stream -> the decoder's tables (the rle8m routine with the info at byte 8 and one section)
__device__ __forceinline__ uint32_t le_tables(const uint8_t *__restrict__ s, uint64_t streamBytes, uint32_t lane, uint32_t *rleBits, uint8_t *codeToCount, uint8_t *listed, uint32_t *hdr)
{
  const uint32_t expIn = ld32(s);
  const uint64_t info = 8ull;
  bool bad = expIn > streamBytes || info + 33u > (uint64_t)expIn;
  uint32_t listedCount = 0;
  if (!bad)
  {
    listedCount = s[info + 32];
    if (listedCount == 0u) listedCount = 255u;
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
    {
      uint32_t next = listedCount;
      for (uint32_t c = 0; c < 256u; c += 64u)
      {
        const bool unl = !listed[c + lane];
        const uint64_t m = __builtin_amdgcn_ballot_w64(unl);
        if (unl) codeToCount[c + lane] = (uint8_t)(next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)));
        next += (uint32_t)__builtin_popcountll(m);
      }
      if (lane == 0u) hdr[0] = (uint32_t)(info + 33u + listedCount);
    }
  }
  else if (lane == 0u)
    hdr[0] = 0u;
  __syncthreads();
  return hdr[0];
}

// carry[k]: parity of the stretch of flagged-valued bytes right in front of piece k (piece k starts at dataStart + k G): the piece's
// first byte is a repeat code iff it is odd.  One lane per piece, a backward scan that ends at the first byte whose value is not flagged
// (typically within a few bytes); a stretch of kLeCarryLimit bytes sets RLE8M_ERR_HEADER in status[1] and the caller decodes with one wave.
__global__ __launch_bounds__(64) void k_le_carry(const uint8_t *__restrict__ s, uint64_t streamBytes, uint32_t G, uint32_t Q, uint32_t *__restrict__ carry, uint32_t *__restrict__ status)
{
  __shared__ uint32_t rleBits[8];
  const uint32_t lane = threadIdx.x;
  const uint32_t expIn = ld32(s);
  if (lane < 8u) rleBits[lane] = ld32(s + 8u + 4u * lane);
  __syncthreads();
  uint32_t listedCount = s[8u + 32u];
  if (listedCount == 0u) listedCount = 255u;
  const uint32_t dataStart = 8u + 33u + listedCount;
  const uint32_t k = blockIdx.x * 64u + lane;
  if (k >= Q || expIn > streamBytes || dataStart > expIn) return;
  uint32_t cnt = 0;
  if (k != 0u)
  {
    const uint64_t begin = (uint64_t)dataStart + (uint64_t)k * G;
    for (uint64_t i = begin; i > dataStart && cnt < kLeCarryLimit; i--)
    {
      const uint32_t b = s[i - 1u];
      if (!((rleBits[b >> 5] >> (b & 31u)) & 1u)) break;
      cnt++;
    }
    if (cnt >= kLeCarryLimit && status) atomicOr(status + 1, 1u);
  }
  carry[k] = cnt & 1u;
}

// k_rle8m_decode_wave over the pieces of ONE stream.  DRY: sizes[k] = the piece's output bytes, nothing is written; else the piece writes
// out + outStart[k] .. outStart[k + 1].  A flagged symbol in a piece's last byte leaves its repeats to the next piece (whose carry is 1).
template <bool DRY>
__global__ __launch_bounds__(64) void k_le_decode_wave(const uint8_t *__restrict__ s, uint64_t streamBytes, uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t wantOut,
                                                        uint32_t G, uint32_t Q, const uint32_t *__restrict__ carryIn, const uint64_t *__restrict__ outStart, uint32_t *__restrict__ sizes)
{
  __shared__ uint32_t rleBits[8];
  __shared__ uint8_t codeToCount[256];
  __shared__ uint8_t listed[256];
  __shared__ uint32_t hdr[4];

  const uint32_t lane = threadIdx.x;
  const uint32_t expIn = ld32(s), expOut = ld32(s + 4);
  const uint32_t dataStart = le_tables(s, streamBytes, lane, rleBits, codeToCount, listed, hdr);
  const uint32_t k = blockIdx.x;
  if (dataStart == 0u || expOut != wantOut)
  {
    if (k == 0u && lane == 0u && status) atomicOr(status, RLE8M_ERR_HEADER);
    return;
  }
  if (k >= Q) return;
  const uint64_t begin64 = (uint64_t)dataStart + (uint64_t)k * G;
  if (begin64 >= (uint64_t)expIn) { if (DRY && lane == 0u) sizes[k] = 0u; return; }
  const uint32_t begin = (uint32_t)begin64;
  const uint32_t end = ((uint64_t)begin + G < (uint64_t)expIn) ? begin + G : expIn;
  const bool lastPiece = end == expIn;
  uint64_t o0 = 0, want = 0xFFFFFFFFull;
  if constexpr (!DRY)
  {
    if (outStart[Q] != (uint64_t)expOut)                                   // the dry pass's sizes do not add up to the header's: nothing is written
    {
      if (k == 0u && lane == 0u && status) atomicOr(status, RLE8M_ERR_STREAM);
      return;
    }
    o0 = outStart[k]; want = outStart[k + 1] - o0;
  }
  uint8_t *const o = out + o0;

  uint32_t op = 0, carry = carryIn[k], pendSym = (carry != 0u) ? (uint32_t)s[begin - 1u] : 0u;
  bool err = false;
  uint32_t bNext = (begin + lane < end) ? (uint32_t)s[begin + lane] : 0u;
  for (uint32_t ip = begin; ip < end; ip += 64u)
  {
    const uint32_t pos = ip + lane;
    const bool valid = pos < end;
    const uint32_t b = bNext;
    bNext = (pos + 64u < end) ? (uint32_t)s[pos + 64u] : 0u;
    const bool fl = valid && ((rleBits[b >> 5] >> (b & 31u)) & 1u);
    const uint64_t F = __builtin_amdgcn_ballot_w64(fl);
    if (F == 0ull && carry == 0u)
    {
      const uint32_t nv = (end - ip < 64u) ? end - ip : 64u;
      if ((uint64_t)nv > want - op) { err = true; break; }
      if constexpr (!DRY) { if (valid) o[op + lane] = (uint8_t)b; }
      op += nv;
      continue;
    }
    uint32_t below = 0;
    if (lane != 0u)
    {
      const uint64_t x = ~F << (64u - lane);
      below = (x == 0ull) ? lane : (uint32_t)__builtin_clzll(x);
      if (below > lane) below = lane;
    }
    const bool isCode = valid && (((below + ((below == lane) ? carry : 0u)) & 1u) != 0u);
    const bool isSym = valid && !isCode;
    const uint32_t nb = (uint32_t)__shfl_down((int)b, 1);
    uint32_t runLen = 0, v = b;
    bool bad = false;
    if (isSym)
    {
      runLen = 1u;
      if (fl)
      {
        if (pos + 1u >= end) bad = lastPiece;                             // the stream's last byte is a flagged symbol without its code; in any other piece the next one writes the repeats
        else if (lane != 63u) runLen += codeToCount[nb];
      }
    }
    else if (isCode && lane == 0u)
    {
      runLen = codeToCount[b];
      v = pendSym;
    }
    uint32_t incl = runLen;
#pragma unroll
    for (uint32_t dd = 1; dd < 64u; dd <<= 1)
    {
      const uint32_t tt = (uint32_t)__shfl_up((int)incl, dd);
      if (lane >= dd) incl += tt;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull || (uint64_t)total > want - op) { err = true; break; }
    if constexpr (!DRY)
    {
      uint8_t *q = o + op + (incl - runLen);
      uint32_t left = runLen;
      const uint32_t vv = v * 0x01010101u;
      const u32x4 line{ vv, vv, vv, vv };
      const uint64_t v8 = (uint64_t)vv * 0x0000000100000001ull;
      if (((uintptr_t)q & 1u) && left >= 1u) { *q = (uint8_t)v; q += 1; left -= 1u; }
      if (((uintptr_t)q & 2u) && left >= 2u) { st16(q, vv & 0xFFFFu); q += 2; left -= 2u; }
      if (((uintptr_t)q & 4u) && left >= 4u) { st32(q, vv); q += 4; left -= 4u; }
      if (((uintptr_t)q & 8u) && left >= 8u) { st64(q, v8); q += 8; left -= 8u; }
      for (; left >= 16u; left -= 16u, q += 16) st128(q, line);
      if (left & 8u) { st64(q, v8); q += 8; }
      if (left & 4u) { st32(q, vv); q += 4; }
      if (left & 2u) { st16(q, vv & 0xFFFFu); q += 2; }
      if (left & 1u) *q = (uint8_t)v;
    }
    op += total;
    const uint32_t top = (F == ~0ull) ? 64u : (uint32_t)__builtin_clzll(~F);
    carry = (top + ((top == 64u) ? carry : 0u)) & 1u;
    pendSym = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
  }
  if constexpr (DRY) { if (lane == 0u) sizes[k] = err ? 0u : op; }
  else if (!err && ((uint64_t)op != want || (lastPiece && carry != 0u))) err = true;
  if (err && lane == 0u && status) atomicOr(status, RLE8M_ERR_STREAM);
}

} // namespace hsrle

