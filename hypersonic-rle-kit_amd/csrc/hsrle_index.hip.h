// hsrle_index.hip.h -- entry-point index of ONE monolithic reference stream, so that the block decoder can work on it with one lane
// per 256 .. 4096 output bytes instead of one lane per stream ("index pass + expand pass").
//
// A reference stream has no random access: the position of packet N+1 is known only after packet N's header (its literal count)
// has been read (reference: src/rleX_extreme_cpu_decode.h:129-162, src/rle8_extreme_cpu.h:1849-1899, src/rleX_Xsl.h:580-760).  What
// the format does offer is that the packet chain SYNCHRONISES: a walk that starts at a wrong byte either dies within a few hops (a
// junk range field with the 4-byte flag points far outside the stream) or falls onto a true packet boundary and is the true chain
// from there on.  The index is built from that, and then PROVEN, never assumed:
//
//   k_index_walk     the stream is cut into regions of G bytes; one lane per region.  The lane guesses where the first packet of
//                    its region starts (it walks from up to M bytes in front of the region, restarting one byte further on whenever
//                    the walk dies), then walks its region from that guess: exit position (first packet start behind the region),
//                    output bytes of the packets that start in the region, and the region's effect on the decoder's symbol state
//                    (Packed: the last symbol; LUT / Short: the move-to-front list) as a K-entry transformer whose entries are
//                    "slot j of the incoming list" or "the symbol stored at stream offset x".
//   k_index_resolve  one workgroup chains the regions from the stream's first packet: region r is right iff the chain arrives
//                    exactly at its guess.  A region the chain jumps over (a literal stretch longer than a region) is skipped; a
//                    region whose guess is wrong is put on the repair list with the true entry and walked again (k_index_walk with
//                    the list), until no region is left on the list.  So a wrong guess costs time, never correctness.  The same
//                    pass scans the output sizes (output offset of every region) and composes the state transformers (symbol state
//                    at every region entry), and checks that the chain ends in the stream's last packet with exactly U output bytes.
//   k_index_records  one lane per region again, now with known entry, output offset and symbol state: for every output position
//                    b * B that falls into one of its packets it writes the decoder state at that position (stream position,
//                    literal / run bytes left, pattern phase, current symbol, move-to-front list): an ENTRY RECORD.
//   k_decode_blocks  (hsrle_decode.hip.h) takes its per-lane start state from the records instead of a block stream header.
#pragma once

#include "hsrle_common.hip.h"
#include "hsrle_decode.hip.h"   // ex32; hsrle_parse.hip.h: Pkt, parse_window
#include "hsrle_launch.h"

namespace hsrle {

constexpr uint32_t IDX_DEAD = 0xFFFFFFFFu;   // exit: the walk met a malformed packet
constexpr uint32_t IDX_END = 0xFFFFFFFEu;    // exit: the walk met the stream's last packet
constexpr uint32_t IDX_SKIP = 0xFFFFFFFDu;   // entry: no packet of the chain starts in this region
constexpr uint32_t IDX_OLD = 0x80000000u;    // transformer entry: slot (v & 15) of the incoming list; otherwise the stream offset of the symbol (< 2^31)
constexpr uint32_t IDX_INIT = 0xC0000000u;   // resolved state entry: entry (v & 15) of the list a decoder starts with

constexpr int kResolveThreads = 1024;
constexpr uint32_t HSRLE_TAIL_PAD_BYTES = 32u;   // zero bytes behind a container's payload (include/hsrle.h: HSRLE_CONTAINER_TAIL_PAD)

enum IndexStatus : uint32_t
{
  IDXS_STREAM = 1u,    // the true chain meets a malformed packet / never reaches the last packet
  IDXS_SIZE = 2u       // the chain does not produce exactly the header's uncompressed size
};

// Where a walk reads its stream from: global memory (one 16-byte window per packet hop: a memory latency per hop), or a copy of a
// piece of the payload in LDS (k_container_records: the streams of the blocks a wave walks are one contiguous piece).  LDS accesses must
// be naturally aligned to be fast (tools/ubench/lds_align.hip), so the LDS reader funnels three aligned 8-byte reads.
struct GlobalReader
{
  const uint8_t *s;
  // (a walk through global memory is bound by the NUMBER of its reads -- 64 lanes at 64 places, every read a request of its own: one
  //  16-byte read, and 8 more bytes only for the header forms that can reach beyond byte 16)
  template <bool EX>
  __device__ __forceinline__ void loadw(uint32_t p, uint64_t &lo, uint64_t &hi, uint64_t &ex) const
  {
    const u32x4 v = ld128(s + p);
    lo = (uint64_t)v.x | ((uint64_t)v.y << 32); hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    if constexpr (EX) ex = ld64(s + p + 16); else ex = 0ull;
  }
  __device__ __forceinline__ void load24(uint32_t p, uint64_t &lo, uint64_t &hi, uint64_t &ex) const { loadw<true>(p, lo, hi, ex); }
  __device__ __forceinline__ void load16(uint32_t p, uint64_t &lo, uint64_t &hi) const { lo = ld64(s + p); hi = ld64(s + p + 8); }
  __device__ __forceinline__ uint32_t load32(uint32_t p) const { return ld32(s + p); }
  __device__ __forceinline__ u32x4 load128(uint32_t p) const { return ld128(s + p); }
};
struct LdsReader
{
  const uint8_t *lds;      // 16-byte aligned
  uint32_t delta;          // LDS offset of the stream's byte 0
  __device__ __forceinline__ u32x4 load128(uint32_t p) const { return lds_read16_w8(lds, delta + p); }
  // 24 bytes from four aligned 8-byte reads (one LDS round trip) and a byte funnel
  __device__ __forceinline__ void load24(uint32_t p, uint64_t &lo, uint64_t &hi, uint64_t &ex) const
  {
    const uint32_t q = delta + p;
    const uint8_t *const src = lds + (q & ~7u);
    const uint64_t w0 = lds_ld64(src), w1 = lds_ld64(src + 8), w2 = lds_ld64(src + 16), w3 = lds_ld64(src + 24);
    const bool d1 = (q & 4u) != 0u;
    const uint32_t z0 = (uint32_t)w0, z1 = (uint32_t)(w0 >> 32), z2 = (uint32_t)w1, z3 = (uint32_t)(w1 >> 32), z4 = (uint32_t)w2, z5 = (uint32_t)(w2 >> 32), z6 = (uint32_t)w3, z7 = (uint32_t)(w3 >> 32);
    const uint32_t y0 = d1 ? z1 : z0, y1 = d1 ? z2 : z1, y2 = d1 ? z3 : z2, y3 = d1 ? z4 : z3, y4 = d1 ? z5 : z4, y5 = d1 ? z6 : z5, y6 = d1 ? z7 : z6;
    const uint32_t n = q & 3u;
    lo = (uint64_t)alignbyte(y1, y0, n) | ((uint64_t)alignbyte(y2, y1, n) << 32);
    hi = (uint64_t)alignbyte(y3, y2, n) | ((uint64_t)alignbyte(y4, y3, n) << 32);
    ex = (uint64_t)alignbyte(y5, y4, n) | ((uint64_t)alignbyte(y6, y5, n) << 32);
  }
  template <bool EX>
  __device__ __forceinline__ void loadw(uint32_t p, uint64_t &lo, uint64_t &hi, uint64_t &ex) const { load24(p, lo, hi, ex); }
  __device__ __forceinline__ void load16(uint32_t p, uint64_t &lo, uint64_t &hi) const
  {
    const u32x4 v = load128(p);
    lo = (uint64_t)v.x | ((uint64_t)v.y << 32); hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
  }
  __device__ __forceinline__ uint32_t load32(uint32_t p) const { return load128(p).x; }
};

template <int FAM, int S, int AL, typename READER>
__device__ __forceinline__ Pkt parse_packet(const READER &rd, uint32_t p, uint32_t C, bool single)
{
  if (p + 2u > C)
  {
    Pkt k;
    k.used = 1; k.lit = 0; k.run = 0; k.op = 0; k.symAt = p; k.hasSym = false; k.last = false; k.bad = true;
    return k;
  }
  uint64_t lo, hi, ex;
  rd.template loadw<header_beyond_16<FAM, S, AL>()>(p, lo, hi, ex);
  return parse_window<FAM, S, AL>(rd, lo, hi, ex, p, C, single);
}

// move-to-front bookkeeping on K u32 entries: op < K moves slot op to the front, op == K pushes v
template <int K>
__device__ __forceinline__ void state_apply(uint32_t (&t)[K > 0 ? K : 1], uint32_t op, uint32_t v)
{
  if constexpr (K > 0)
  {
    uint32_t front = v;
#pragma unroll
    for (int j = 0; j < K; j++)
      if (op == (uint32_t)j) front = t[j];
    const uint32_t limit = (op >= (uint32_t)K) ? (uint32_t)K - 1u : op;
#pragma unroll
    for (int j = K - 1; j >= 1; j--)
      if ((uint32_t)j <= limit) t[j] = t[j - 1];
    t[0] = front;
  }
}

// ---- pass 1 (and the repair passes): one lane per region ----
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_index_walk(const uint8_t *__restrict__ s, uint32_t C, uint32_t p0, uint32_t G, uint32_t M, uint32_t R, uint32_t single,
                                                   const uint32_t *__restrict__ list, uint32_t listCount, const uint32_t *__restrict__ fix,
                                                   uint32_t *__restrict__ mark, uint32_t roundTag, uint32_t extMax,
                                                   uint32_t *__restrict__ gOut, uint32_t *__restrict__ eOut, uint64_t *__restrict__ olenOut, uint32_t *__restrict__ tOut)
{
  constexpr int KE = IndexState<FAM>::KE;
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  uint32_t r;
  if (list != nullptr) { if (i >= listCount) return; r = list[i]; }
  else { if (i >= R) return; r = i; }

  uint32_t start = p0 + r * G, endr = start + G;
  const bool sgl = single != 0u;

  // ONE loop, one packet per trip, whatever a lane is doing (guessing or walking its region): nested loops would make the whole wave
  // wait for the lane with the most restarts in every restart round.
  //   guess: the first walk from <= M bytes in front of the region that survives until the region starts.  A walk that begins at the
  //   stream's first packet IS the chain; any other one is taken for dead when it claims a literal stretch of M bytes or more (a
  //   true one of that size makes the chain jump, and then k_index_resolve hands this region its entry anyway); a dead walk is
  //   started again one byte further on.
  const bool fromStart = start - p0 <= M;
  bool guessing = (list == nullptr) && r != 0u;
  uint32_t t = fromStart ? p0 : start - M;       // where the current guess walk started
  uint32_t hopsLeft = 4u * M + 64u;
  uint32_t x = guessing ? t : ((list != nullptr) ? fix[r] : p0);
  uint32_t q = x, ex = 0;
  uint64_t ol = 0;
  uint32_t tr[KE > 0 ? KE : 1];
#pragma unroll
  for (int j = 0; j < (KE > 0 ? KE : 1); j++) tr[j] = IDX_OLD | (uint32_t)j;

  // A REPAIR walk (list != nullptr) goes on into the regions behind its own while the chain does not arrive at their recorded guesses:
  // wrong guesses come in streaks, and a streak would otherwise cost one round per region.  It stops in front of a region that is on this
  // round's list itself or that another repair walk has claimed (mark == roundTag: one writer per record -- two could tear it).
  for (uint32_t ext = 0;; ext++)
  {
  for (;;)
  {
    if (guessing && x >= start) { guessing = false; q = x; }
    if (!guessing && x >= endr) { ex = x; break; }
    const Pkt k = parse_packet<FAM, S, AL>(GlobalReader{ s }, x, C, sgl);      // the one parse site of the loop: guessing and walking lanes share it
    if (guessing)
    {
      const bool truth = fromStart && t == p0;
      if (hopsLeft-- == 0u || (truth && (k.bad || k.last))) { guessing = false; q = start; x = start; }      // nothing (left) to guess: any entry will do
      else if (k.bad || k.last || (k.lit >= M && !truth)) { t++; x = t; if (t >= start) { guessing = false; q = start; } }
      else x += k.used + k.lit;
    }
    else
    {
      if (k.bad) { ex = IDX_DEAD; break; }
      ol += (uint64_t)k.lit + (uint64_t)k.run;
      state_apply<KE>(tr, k.op, k.symAt);
      if (k.last) { ex = IDX_END; break; }
      x += k.used + k.lit;
    }
  }
  gOut[r] = q;
  eOut[r] = ex;
  olenOut[r] = ol;
  if constexpr (KE > 0)
  {
#pragma unroll
    for (int j = 0; j < KE; j++) tOut[(uint64_t)r * KE + j] = tr[j];
  }
  if (list == nullptr || ex >= IDX_SKIP || ext >= extMax) break;
  const uint32_t next = (ex - p0) / G;                                  // the region the chain enters next (ex >= endr: next > r)
  // claim the region for this round (its record must have ONE writer: the chain may have jumped over listed regions whose lanes walk
  // on into the same region): whoever raises mark[next] to this round's tag first owns it
  if (next >= R || atomicMax(mark + next, roundTag) >= roundTag || gOut[next] == ex) break;
  r = next; start = p0 + r * G; endr = start + G;
  x = ex; q = ex; ol = 0;
#pragma unroll
  for (int j = 0; j < (KE > 0 ? KE : 1); j++) tr[j] = IDX_OLD | (uint32_t)j;
  }
}

// ---- pass 2: chain the regions, scan sizes and states.  ONE workgroup. ----
// ctrl: [0] regions on the repair list, [1] IndexStatus bits (final only when [0] == 0), [2..3] output bytes of the chain
template <int KE>
__device__ __forceinline__ void state_compose(uint32_t (&out)[KE > 0 ? KE : 1], const uint32_t (&first)[KE > 0 ? KE : 1], const uint32_t (&then)[KE > 0 ? KE : 1])
{
#pragma unroll
  for (int j = 0; j < KE; j++)
  {
    const uint32_t v = then[j];
    uint32_t w = v;
    if ((v >> 30) == 2u)
    {
#pragma unroll
      for (int m = 0; m < KE; m++)
        if ((v & 15u) == (uint32_t)m) w = first[m];
    }
    out[j] = w;
  }
}

// ---- pass 2, the common case in parallel (round 4).  k_index_resolve below is ONE workgroup that goes through the regions 1 024 at a time:
//      243 us for the 71 096 regions of the 1 GiB config-2 stream, and proportionally more with smaller regions -- which is what kept the
//      regions at 8 KiB and the walks (latency chains of one lane per region) long.  When every guess is right -- the walk of region r
//      starts where the walk of region r - 1 ended, which only needs e[r - 1] == g[r] -- the pass is three scans: output sizes, state
//      transformers, nothing else.  Three small kernels do that for all FULL batches but the last (which holds the end of the chain and
//      stays with k_index_resolve); `fast` = [ok flag, -, carry out lo, hi, carry state x KS]; per batch: [out sum lo, hi, T x KS].
//      Any region that fails the check clears the flag: k_index_resolve then does everything, as before.
constexpr uint32_t kFastBatchWords = 2u + 8u;       // per batch: out sum (2 words) + up to 7 transformer words (+ pad)

// inclusive scan of (output bytes, transformer) over the workgroup's 1 024 regions; returns this thread's EXCLUSIVE prefix and the totals
template <int KE>
__device__ __forceinline__ void resolve_batch_scan(uint32_t tid, uint64_t v, const uint32_t (&myT)[KE > 0 ? KE : 1], uint64_t *swave, uint32_t *sWaveT,
                                                   uint64_t &exclOut, uint64_t &allOut, uint32_t (&exclT)[KE > 0 ? KE : 1], uint32_t (&allT)[KE > 0 ? KE : 1])
{
  constexpr int NT = kResolveThreads;
  constexpr int KS = KE > 0 ? KE : 1;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint64_t xsum = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint64_t y = __shfl_up(xsum, d, 64);
    if ((int)lane >= d) xsum += y;
  }
  if (lane == 63u) swave[wave] = xsum;
  uint32_t tInc[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) tInc[j] = myT[j];
  if constexpr (KE > 0)
  {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      uint32_t left[KS], tmp[KS];
#pragma unroll
      for (int j = 0; j < KE; j++) left[j] = (uint32_t)__shfl_up((int)tInc[j], d, 64);
      state_compose<KE>(tmp, left, tInc);
      if ((int)lane >= d)
      {
#pragma unroll
        for (int j = 0; j < KE; j++) tInc[j] = tmp[j];
      }
    }
    if (lane == 63u)
    {
#pragma unroll
      for (int j = 0; j < KE; j++) sWaveT[wave * KS + j] = tInc[j];
    }
  }
  __syncthreads();
  uint64_t wbase = 0, wall = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++)
  {
    const uint64_t t = swave[w];
    if ((uint32_t)w < wave) wbase += t;
    wall += t;
  }
  exclOut = wbase + xsum - v;
  allOut = wall;
  uint32_t tWaves[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) { tWaves[j] = IDX_OLD | (uint32_t)j; allT[j] = IDX_OLD | (uint32_t)j; exclT[j] = IDX_OLD | (uint32_t)j; }
  if constexpr (KE > 0)
  {
    uint32_t wt[KS];
#pragma unroll
    for (int j = 0; j < KE; j++) wt[j] = (lane < (uint32_t)(NT / 64)) ? sWaveT[lane * KS + j] : (IDX_OLD | (uint32_t)j);
#pragma unroll
    for (int d = 1; d < NT / 64; d <<= 1)
    {
      uint32_t left[KS], tmp[KS];
#pragma unroll
      for (int j = 0; j < KE; j++) left[j] = (uint32_t)__shfl_up((int)wt[j], d, 64);
      state_compose<KE>(tmp, left, wt);
      if ((int)lane >= d)
      {
#pragma unroll
        for (int j = 0; j < KE; j++) wt[j] = tmp[j];
      }
    }
#pragma unroll
    for (int j = 0; j < KE; j++)
    {
      const uint32_t before = (uint32_t)__shfl((int)wt[j], (int)(wave > 0u ? wave - 1u : 0u), 64);
      tWaves[j] = (wave > 0u) ? before : (IDX_OLD | (uint32_t)j);
      allT[j] = (uint32_t)__shfl((int)wt[j], NT / 64 - 1, 64);
    }
    uint32_t tLane[KS];
#pragma unroll
    for (int j = 0; j < KE; j++)
    {
      const uint32_t prevLane = (uint32_t)__shfl_up((int)tInc[j], 1, 64);
      tLane[j] = (lane == 0u) ? (IDX_OLD | (uint32_t)j) : prevLane;
    }
    state_compose<KE>(exclT, tWaves, tLane);
  }
  __syncthreads();
}

// A: one workgroup per full batch: the parallel check and the batch's totals
template <int KE>
__global__ __launch_bounds__(kResolveThreads) void k_resolve_fast_totals(const uint32_t *__restrict__ g, const uint32_t *__restrict__ e, const uint64_t *__restrict__ olen,
                                                                         const uint32_t *__restrict__ tIn, uint32_t p0, uint32_t G, uint32_t *__restrict__ fast, uint32_t *__restrict__ batch)
{
  constexpr int NT = kResolveThreads;
  constexpr int KS = KE > 0 ? KE : 1;
  __shared__ uint64_t swave[NT / 64];
  __shared__ uint32_t sWaveT[(NT / 64) * KS];
  const uint32_t tid = threadIdx.x, r = blockIdx.x * (uint32_t)NT + tid;
  const uint32_t gg = g[r], ee = e[r];
  const uint32_t prev = (r == 0u) ? p0 : e[r - 1u];
  const bool ok = prev == gg && gg < p0 + (r + 1u) * G && ee < IDX_SKIP;
  const int allok = __syncthreads_and(ok ? 1 : 0);
  if (!allok) { if (tid == 0u) atomicAnd(fast, 0u); return; }
  uint32_t myT[KS], exclT[KS], allT[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) myT[j] = (KE > 0) ? tIn[(uint64_t)r * KS + j] : (IDX_OLD | (uint32_t)j);
  uint64_t exclOut, allOut;
  resolve_batch_scan<KE>(tid, olen[r], myT, swave, sWaveT, exclOut, allOut, exclT, allT);
  if (tid == 0u)
  {
    uint32_t *w = batch + (uint64_t)blockIdx.x * kFastBatchWords;
    w[0] = (uint32_t)allOut; w[1] = (uint32_t)(allOut >> 32);
#pragma unroll
    for (int j = 0; j < KS; j++) w[2 + j] = allT[j];
  }
}

// B: ONE workgroup: exclusive scan of the batch totals -> every batch's carries (in place), the carries behind the last one into fast[2 ..]
template <int KE>
__global__ __launch_bounds__(kResolveThreads) void k_resolve_fast_carries(uint32_t *__restrict__ fast, uint32_t *__restrict__ batch, uint32_t batches)
{
  constexpr int NT = kResolveThreads;
  constexpr int KS = KE > 0 ? KE : 1;
  __shared__ uint64_t swave[NT / 64];
  __shared__ uint32_t sWaveT[(NT / 64) * KS];
  __shared__ uint64_t sCarryOut;
  __shared__ uint32_t sCarryT[KS];
  if (fast[0] != 1u) return;
  const uint32_t tid = threadIdx.x;
  if (tid == 0u)
  {
    sCarryOut = 0;
#pragma unroll
    for (int j = 0; j < KS; j++) sCarryT[j] = IDX_INIT | (uint32_t)j;
  }
  __syncthreads();
  for (uint32_t base = 0; base < batches; base += NT)
  {
    const uint32_t i = base + tid;
    const bool valid = i < batches;
    uint32_t *w = batch + (uint64_t)i * kFastBatchWords;
    const uint64_t v = valid ? ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) : 0ull;
    uint32_t myT[KS], exclT[KS], allT[KS];
#pragma unroll
    for (int j = 0; j < KS; j++) myT[j] = (KE > 0 && valid) ? w[2 + j] : (IDX_OLD | (uint32_t)j);
    uint64_t exclOut, allOut;
    resolve_batch_scan<KE>(tid, v, myT, swave, sWaveT, exclOut, allOut, exclT, allT);
    if (valid)
    {
      const uint64_t o = sCarryOut + exclOut;
      w[0] = (uint32_t)o; w[1] = (uint32_t)(o >> 32);
      if constexpr (KE > 0)
      {
        uint32_t carry[KS], st[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) carry[j] = sCarryT[j];
        state_compose<KE>(st, carry, exclT);
#pragma unroll
        for (int j = 0; j < KE; j++) w[2 + j] = st[j];
      }
    }
    __syncthreads();
    if (tid == 0u)
    {
      sCarryOut += allOut;
      if constexpr (KE > 0)
      {
        uint32_t carry[KS], st[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) carry[j] = sCarryT[j];
        state_compose<KE>(st, carry, allT);
#pragma unroll
        for (int j = 0; j < KE; j++) sCarryT[j] = st[j];
      }
    }
    __syncthreads();
  }
  if (tid == 0u)
  {
    fast[2] = (uint32_t)sCarryOut; fast[3] = (uint32_t)(sCarryOut >> 32);
#pragma unroll
    for (int j = 0; j < KS; j++) fast[4 + j] = sCarryT[j];
  }
}

// C: one workgroup per full batch: what k_index_resolve writes for a region that passed -- entry, output position, state in front of it
template <int KE>
__global__ __launch_bounds__(kResolveThreads) void k_resolve_fast_emit(const uint32_t *__restrict__ g, const uint64_t *__restrict__ olen, const uint32_t *__restrict__ tIn,
                                                                       const uint32_t *__restrict__ fast, const uint32_t *__restrict__ batch,
                                                                       uint32_t *__restrict__ entry, uint64_t *__restrict__ outStart, uint32_t *__restrict__ stateIn)
{
  constexpr int NT = kResolveThreads;
  constexpr int KS = KE > 0 ? KE : 1;
  __shared__ uint64_t swave[NT / 64];
  __shared__ uint32_t sWaveT[(NT / 64) * KS];
  if (fast[0] != 1u) return;
  const uint32_t tid = threadIdx.x, r = blockIdx.x * (uint32_t)NT + tid;
  uint32_t myT[KS], exclT[KS], allT[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) myT[j] = (KE > 0) ? tIn[(uint64_t)r * KS + j] : (IDX_OLD | (uint32_t)j);
  uint64_t exclOut, allOut;
  resolve_batch_scan<KE>(tid, olen[r], myT, swave, sWaveT, exclOut, allOut, exclT, allT);
  const uint32_t *w = batch + (uint64_t)blockIdx.x * kFastBatchWords;
  entry[r] = g[r];
  outStart[r] = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) + exclOut;
  if constexpr (KE > 0)
  {
    uint32_t carry[KS], st[KS];
#pragma unroll
    for (int j = 0; j < KE; j++) carry[j] = w[2 + j];
    state_compose<KE>(st, carry, exclT);
#pragma unroll
    for (int j = 0; j < KE; j++) stateIn[(uint64_t)r * KE + j] = st[j];
  }
}

template <int KE>
__global__ __launch_bounds__(kResolveThreads) void k_index_resolve(const uint32_t *__restrict__ g, const uint32_t *__restrict__ e, const uint64_t *__restrict__ olen,
                                                                   const uint32_t *__restrict__ tIn, uint32_t R, uint32_t p0, uint32_t G, uint64_t U,
                                                                   uint32_t *__restrict__ entry, uint64_t *__restrict__ outStart, uint32_t *__restrict__ stateIn,
                                                                   uint32_t *__restrict__ fix, uint32_t *__restrict__ list, uint32_t *__restrict__ ctrl,
                                                                   uint32_t *__restrict__ mark, uint32_t roundTag, const uint32_t *__restrict__ fast, uint32_t fastBatches)
{
  // fast != nullptr: the parallel passes (k_resolve_fast_*) may have done the first fastBatches batches -- fast[0] == 1 says they did: every
  // region of those batches passed the parallel check, their records are written, and this kernel goes on behind them from the carries
  // fast[2 ..] (output bytes, state) with the chain standing at e[fastBatches * NT - 1].  fast[0] == 0: nothing was done, start at region 0.
  constexpr int NT = kResolveThreads;
  constexpr int KS = KE > 0 ? KE : 1;
  constexpr uint8_t F_NONE = 0, F_OK = 1, F_SKIP = 2, F_DIRTY = 3;
  const uint32_t base0 = (fast != nullptr && fast[0] == 1u) ? fastBatches * (uint32_t)NT : 0u;
  __shared__ uint32_t sg[NT], se[NT];
  __shared__ uint8_t sflag[NT];
  __shared__ uint64_t sokMask[NT / 64];                                 // bit: the region passed the parallel check
  __shared__ uint64_t swave[NT / 64];
  __shared__ uint32_t sWaveT[(NT / 64) * KS];                            // per wave: composition of its regions' state transformers
  __shared__ uint32_t sCur, sEnded, sTrusted, sDirty, sStatus;
  __shared__ uint32_t sCarryState[KS];
  __shared__ uint64_t sCarryOut;

  const uint32_t tid = threadIdx.x;
  if (tid == 0)
  {
    sCur = p0; sEnded = 0; sTrusted = 1; sDirty = 0; sStatus = 0; sCarryOut = 0;
#pragma unroll
    for (int j = 0; j < KS; j++) sCarryState[j] = IDX_INIT | (uint32_t)j;
    if (base0 != 0u)
    {
      sCur = e[base0 - 1u];
      sCarryOut = (uint64_t)fast[2] | ((uint64_t)fast[3] << 32);
#pragma unroll
      for (int j = 0; j < KS; j++) sCarryState[j] = (KE > 0) ? fast[4 + j] : (IDX_INIT | (uint32_t)j);
    }
  }
  __syncthreads();

  // the next batch's walk results are requested while this batch is worked on (the pass is one workgroup: nothing else hides the latency)
  const uint32_t r0 = base0 + tid;
  uint32_t nG = (r0 < R) ? g[r0] : 0u, nE = (r0 < R) ? e[r0] : 0u;
  uint64_t nOl = (r0 < R) ? olen[r0] : 0ull;
  uint32_t nT[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) nT[j] = (KE > 0 && r0 < R) ? tIn[(uint64_t)r0 * KS + j] : (IDX_OLD | (uint32_t)j);

  for (uint32_t base = base0; base < R; base += NT)
  {
    const uint32_t r = base + tid;
    const bool valid = r < R;
    const uint32_t gg = nG, ee = nE;
    const uint64_t myOl = nOl;
    uint32_t myT[KS];
#pragma unroll
    for (int j = 0; j < KS; j++) myT[j] = nT[j];
    {
      const uint32_t rn = r + NT;
      nG = (rn < R) ? g[rn] : 0u; nE = (rn < R) ? e[rn] : 0u; nOl = (rn < R) ? olen[rn] : 0ull;
#pragma unroll
      for (int j = 0; j < KS; j++) nT[j] = (KE > 0 && rn < R) ? tIn[(uint64_t)rn * KS + j] : (IDX_OLD | (uint32_t)j);
    }
    sg[tid] = gg; se[tid] = ee;
    __syncthreads();
    const uint32_t cur0 = sCur, ended0 = sEnded;                        // (read in front of the barrier below: thread 0 writes them behind it)
    const uint32_t prev = (tid == 0) ? cur0 : se[tid - 1];
    const uint32_t endr = p0 + (r + 1u) * G;
    const bool ok = !valid || (ended0 == 0u && prev == gg && gg < endr && ee < IDX_SKIP);
    const int allok = __syncthreads_and(ok ? 1 : 0);
    const uint32_t n = (R - base < (uint32_t)NT) ? R - base : (uint32_t)NT;
    {
      const uint64_t m = __ballot(ok);
      if ((tid & 63u) == 0u) sokMask[tid >> 6] = m;
    }
    const bool allSkipped = ended0 != 0u || cur0 >= p0 + (base + n) * G;   // the chain is over, or jumps over the whole batch

    if (allok)
    {
      sflag[tid] = valid ? F_OK : F_NONE;
      if (tid == 0) sCur = se[n - 1u];
    }
    else if (allSkipped)
      sflag[tid] = valid ? F_SKIP : F_NONE;
    else
      sflag[tid] = valid ? F_OK : F_NONE;                                 // what thread 0 does not touch below is a proven link
    __syncthreads();
    if (!allok && !allSkipped && tid == 0)
    {
      // the rare path, one thread over the batch: jumps over regions, the end of the chain, wrong guesses.  While the walk is IN SYNC
      // (it stands where the previous region's walk left) the parallel check above already holds the answer: stretches of regions
      // that passed it are skipped over.
      uint32_t cur = sCur, ended = sEnded, trusted = sTrusted;
      bool insync = true;
      // (always from global memory: a choice between the LDS copy and the global array becomes a flat access that this compiler cannot encode)
      auto next_guess = [&](uint32_t k, uint32_t rr) -> uint32_t { (void)k; return g[rr + 1u]; };
      for (uint32_t k = 0; k < n; k++)
      {
        if (insync && !ended && ((sokMask[k >> 6] >> (k & 63u)) & 1ull) != 0ull)
        {
          // first region at or behind k that did NOT pass: one word of the mask at a time
          uint32_t j = n;
          for (uint32_t wd = k >> 6; wd < (uint32_t)(NT / 64); wd++)
          {
            uint64_t bad = ~sokMask[wd];
            if (wd == (k >> 6)) bad &= ~0ull << (k & 63u);
            if (bad != 0ull) { j = wd * 64u + (uint32_t)__builtin_ctzll(bad); break; }
          }
          if (j > n) j = n;
          cur = se[j - 1u];                                                // regions k .. j-1 keep F_OK
          k = j - 1u;
          continue;
        }
        insync = false;
        const uint32_t rr = base + k;
        const uint32_t endk = p0 + (rr + 1u) * G;
        if (ended || cur >= endk) { sflag[k] = F_SKIP; continue; }
        if (cur == sg[k])
        {
          sflag[k] = F_OK;
          const uint32_t x = se[k];
          if (x < IDX_SKIP) { cur = x; insync = true; }
          else if (trusted) { if (x == IDX_DEAD) sStatus |= IDXS_STREAM; ended = 1; }
          else if (rr + 1u < R) cur = next_guess(k, rr);                          // behind a wrong guess nothing is final: go on from the next guess
        }
        else
        {
          // wrong guess: this region is walked again from `cur`.  Until then nothing behind it is proven; the chain is continued
          // from the next region's own guess so that further wrong guesses are found (and repaired) in the same round.
          sflag[k] = F_DIRTY;
          fix[rr] = cur;
          mark[rr] = roundTag;
          list[sDirty++] = rr;
          trusted = 0;
          if (rr + 1u < R) cur = next_guess(k, rr);
        }
      }
      for (uint32_t k = n; k < (uint32_t)NT; k++) sflag[k] = F_NONE;
      sCur = cur; sEnded = ended; sTrusted = trusted;
    }
    __syncthreads();

    const uint8_t fl = sflag[tid];
    // exclusive scan of the output sizes
    const uint64_t v = (fl == F_OK) ? myOl : 0ull;
    uint64_t xsum = v;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint64_t y = __shfl_up(xsum, d, 64);
      if ((int)lane >= d) xsum += y;
    }
    if (lane == 63u) swave[wave] = xsum;
    // inclusive scan of the state transformers: inside the wave with shuffles, across the waves through their totals in LDS
    uint32_t tInc[KS];                                                  // composition of the regions of this wave up to and including mine
#pragma unroll
    for (int j = 0; j < KS; j++) tInc[j] = (KE > 0 && fl == F_OK) ? myT[j] : (IDX_OLD | (uint32_t)j);
    if constexpr (KE > 0)
    {
#pragma unroll
      for (int d = 1; d < 64; d <<= 1)
      {
        uint32_t left[KS], tmp[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) left[j] = (uint32_t)__shfl_up((int)tInc[j], d, 64);
        state_compose<KE>(tmp, left, tInc);
        if ((int)lane >= d)
        {
#pragma unroll
          for (int j = 0; j < KE; j++) tInc[j] = tmp[j];
        }
      }
      if (lane == 63u)
      {
#pragma unroll
        for (int j = 0; j < KE; j++) sWaveT[wave * KS + j] = tInc[j];
      }
    }
    __syncthreads();
    uint64_t wbase = 0, wall = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++)
    {
      const uint64_t t = swave[w];
      if ((uint32_t)w < wave) wbase += t;
      wall += t;
    }
    // composition of the waves in front of mine, and of all waves (every wave scans the NT / 64 wave totals itself: lane w holds wave w's)
    uint32_t tWaves[KS], tAll[KS];
#pragma unroll
    for (int j = 0; j < KS; j++) { tWaves[j] = IDX_OLD | (uint32_t)j; tAll[j] = IDX_OLD | (uint32_t)j; }
    if constexpr (KE > 0)
    {
      uint32_t wt[KS];
#pragma unroll
      for (int j = 0; j < KE; j++) wt[j] = (lane < (uint32_t)(NT / 64)) ? sWaveT[lane * KS + j] : (IDX_OLD | (uint32_t)j);
#pragma unroll
      for (int d = 1; d < NT / 64; d <<= 1)
      {
        uint32_t left[KS], tmp[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) left[j] = (uint32_t)__shfl_up((int)wt[j], d, 64);
        state_compose<KE>(tmp, left, wt);
        if ((int)lane >= d)
        {
#pragma unroll
          for (int j = 0; j < KE; j++) wt[j] = tmp[j];
        }
      }
#pragma unroll
      for (int j = 0; j < KE; j++)
      {
        const uint32_t before = (uint32_t)__shfl((int)wt[j], (int)(wave > 0u ? wave - 1u : 0u), 64);
        tWaves[j] = (wave > 0u) ? before : (IDX_OLD | (uint32_t)j);
        tAll[j] = (uint32_t)__shfl((int)wt[j], NT / 64 - 1, 64);
      }
    }
    uint32_t tExcl[KS];                                                 // my wave's lanes in front of me (all lanes take part in the shuffle)
#pragma unroll
    for (int j = 0; j < KS; j++)
    {
      const uint32_t prevLane = (uint32_t)__shfl_up((int)tInc[j], 1, 64);
      tExcl[j] = (lane == 0u) ? (IDX_OLD | (uint32_t)j) : prevLane;
    }
    if (valid)
    {
      entry[r] = (fl == F_OK) ? gg : IDX_SKIP;
      outStart[r] = sCarryOut + wbase + xsum - v;
      if constexpr (KE > 0)
      {
        // the regions in front of mine: the batches before (carry), the waves before mine, my wave's lanes before me
        uint32_t st[KS], carry[KS], upto[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) carry[j] = sCarryState[j];
        state_compose<KE>(upto, carry, tWaves);
        state_compose<KE>(st, upto, tExcl);
#pragma unroll
        for (int j = 0; j < KE; j++) stateIn[(uint64_t)r * KE + j] = st[j];
      }
    }
    __syncthreads();
    if (tid == 0)
    {
      sCarryOut += wall;
      if constexpr (KE > 0)
      {
        uint32_t st[KS], carry[KS];
#pragma unroll
        for (int j = 0; j < KE; j++) carry[j] = sCarryState[j];
        state_compose<KE>(st, carry, tAll);
#pragma unroll
        for (int j = 0; j < KE; j++) sCarryState[j] = st[j];
      }
    }
    __syncthreads();
  }

  if (tid == 0)
  {
    uint32_t status = sStatus;
    if (sDirty == 0u)
    {
      if (sEnded == 0u) status |= IDXS_STREAM;                          // the chain never reached the stream's last packet
      if (sCarryOut != U) status |= IDXS_SIZE;
    }
    ctrl[0] = sDirty;
    ctrl[1] = status;
    ctrl[2] = (uint32_t)sCarryOut;
    ctrl[3] = (uint32_t)(sCarryOut >> 32);
  }
}

// symbol of a resolved state entry as the decoder keeps it: masked to S bytes in up to four dwords
template <int S, typename READER>
__device__ __forceinline__ u32x4 index_symbol(const READER &rd, uint32_t v, bool packedInit)
{
  if ((v >> 30) == 3u)
  {
    constexpr uint32_t init[7] = { 0x00u, 0x7Fu, 0xFFu, 0x01u, 0x7Eu, 0x80u, 0xFEu };      // rleX_Xsl.h:533-543 (Packed decoders start with symbol 0)
    const uint32_t b = packedInit ? 0u : init[(v & 15u) % 7u] * 0x01010101u;
    u32x4 m = u32x4{ b, b, b, b };
    if constexpr (S == 1) m = u32x4{ b & 0xFFu, 0, 0, 0 };
    else if constexpr (S == 2) m = u32x4{ b & 0xFFFFu, 0, 0, 0 };
    else if constexpr (S == 3) m = u32x4{ b & 0xFFFFFFu, 0, 0, 0 };
    else if constexpr (S == 4) m = u32x4{ b, 0, 0, 0 };
    else if constexpr (S == 6) m = u32x4{ b, b & 0xFFFFu, 0, 0 };
    else if constexpr (S == 8) m = u32x4{ b, b, 0, 0 };
    return m;
  }
  const u32x4 x = rd.load128(v);
  if constexpr (S == 1) return u32x4{ x.x & 0xFFu, 0, 0, 0 };
  else if constexpr (S == 2) return u32x4{ x.x & 0xFFFFu, 0, 0, 0 };
  else if constexpr (S == 3) return u32x4{ x.x & 0xFFFFFFu, 0, 0, 0 };
  else if constexpr (S == 4) return u32x4{ x.x, 0, 0, 0 };
  else if constexpr (S == 6) return u32x4{ x.x, x.y & 0xFFFFu, 0, 0 };
  else if constexpr (S == 8) return u32x4{ x.x, x.y, 0, 0 };
  else return x;
}

// Walk the packets of the stream `s` (C bytes; its first byte sits at byte `sBase` of what the decoder will call `payload`) from
// packet start x while x < endr, with o = output position of that packet's first byte and st = symbol state in front of it, and write
// an entry record for every output position b * B (< limit) that falls into one of the packets.
//   record dwords: [0..1] stream position (relative to `payload`), [2] literal bytes left, [3] run bytes left, [4] phase | flags,
//   [5] bytes from that position to the stream's end, [6..9] current symbol, [10..] move-to-front list
// Returns 0 (stopped at endr), 1 (the stream's last packet was walked) or 2 (malformed packet); o = output position behind the walk.
template <int FAM, int S, int AL, typename READER>
__device__ __forceinline__ uint32_t walk_emit_records(const READER &s, uint64_t sBase, uint32_t C, uint32_t x, uint32_t endr, uint64_t &o,
                                                      uint32_t (&st)[IndexState<FAM>::KE > 0 ? IndexState<FAM>::KE : 1], bool sgl, uint32_t singleSym,
                                                      uint64_t limit, uint32_t B, uint32_t *__restrict__ rec)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int KE = IndexState<FAM>::KE;
  constexpr int SW = TR::SW;
  uint32_t curSym = IDX_INIT;      // offset / tag of the current packet's symbol (families without a list)
  // the next output position that gets a record (kept incrementally: a 64 bit division per packet hop was most of this function's time)
  uint64_t nextRec = (o + B - 1u) / B * (uint64_t)B;
  while (x < endr)
  {
    const Pkt k = parse_packet<FAM, S, AL>(s, x, C, sgl);
    if (k.bad) return 2u;
    state_apply<KE>(st, k.op, k.symAt);
    if (k.hasSym) curSym = k.symAt;
    const uint64_t outEnd = o + (uint64_t)k.lit + (uint64_t)k.run;
    const uint32_t body = x + k.used;
    for (; nextRec < outEnd && nextRec < limit; nextRec += B)
    {
      const uint64_t into = nextRec - o;
      const uint64_t b = (B & (B - 1u)) == 0u ? nextRec >> __builtin_ctz(B) : nextRec / B;
      uint32_t *const w = rec + b * (uint64_t)kEntryRecDwords;
      uint32_t rsp, rlit, rrun, phase = 0;
      if (into < (uint64_t)k.lit) { rsp = body + (uint32_t)into; rlit = k.lit - (uint32_t)into; rrun = k.run; }
      else
      {
        const uint64_t t = into - k.lit;
        rsp = body + k.lit; rlit = 0; rrun = k.run - (uint32_t)t;
        phase = (uint32_t)(t % (uint64_t)S);
      }
      u32x4 sym;
      if (sgl || TR::kShortSingle) sym = u32x4{ singleSym & 0xFFu, 0, 0, 0 };
      else if constexpr (KE > 0) sym = index_symbol<S>(s, st[0], FAM == PACKED);
      else sym = index_symbol<S>(s, curSym, true);
      const uint64_t at = sBase + rsp;
      w[0] = (uint32_t)at; w[1] = (uint32_t)(at >> 32); w[2] = rlit; w[3] = rrun;
      w[4] = phase | (k.last ? REC_LAST : 0u) | (sgl ? REC_SINGLE : 0u);
      w[5] = C - rsp;
      w[6] = sym.x; w[7] = sym.y; w[8] = sym.z; w[9] = sym.w;
      if constexpr (TR::kMtf)
      {
#pragma unroll
        for (int j = 0; j < KE; j++)
        {
          const u32x4 v = index_symbol<S>(s, st[j], false);
          w[10 + j * SW] = v.x;
          if constexpr (SW > 1) w[10 + j * SW + 1] = v.y;
        }
      }
    }
    o = outEnd;
    if (k.last) return 1u;
    x = body + k.lit;
  }
  return 0u;
}

// ---- pass 3: the decoder state at every output position b * B ----
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_index_records(const uint8_t *__restrict__ s, uint32_t C, uint32_t p0, uint32_t G, uint32_t R, uint32_t single, uint32_t singleSym,
                                                      const uint32_t *__restrict__ entry, const uint64_t *__restrict__ outStart, const uint32_t *__restrict__ stateIn,
                                                      uint64_t U, uint32_t B, uint32_t *__restrict__ rec, const uint32_t *__restrict__ gate)
{
  constexpr int KE = IndexState<FAM>::KE;
  constexpr int KS = KE > 0 ? KE : 1;
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  if (r >= R) return;
  // enqueued behind the resolve pass without the host in between: entries that did not pass the check are junk and a walk from junk is
  // unbounded work (round 4 measured 105 ms) -- the records stay zero, the decoder's lanes end at once, the host repairs and comes again
  if (gate != nullptr && (gate[0] | gate[1]) != 0u) return;
  const uint32_t x = entry[r];
  if (x == IDX_SKIP) return;
  uint64_t o = outStart[r];
  // the state as offsets / init tags (what the transformers carry); symbols are fetched when a record is written
  uint32_t st[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) st[j] = (KE > 0) ? stateIn[(uint64_t)r * KS + j] : 0u;
  (void)walk_emit_records<FAM, S, AL>(GlobalReader{ s }, 0ull, C, x, p0 + (r + 1u) * G, o, st, single != 0u, singleSym, U, B, rec);   // (cannot meet a malformed packet: k_index_walk walked this chain)
}

// A wave copies `total` bytes of the payload from w0 (16-byte aligned) into its LDS window: EIGHT 16-byte loads per lane in flight.  (Until
// round 4 this was one load, wait, LDS store per trip: 32 memory latencies in a row for a 32 KB window -- 50 of the 68 us that
// k_container_records / k_container_packets took on the 88 MB frame were this loop, not the walk.)
__device__ __forceinline__ void stage_window(uint8_t *window, const uint8_t *__restrict__ payload, uint64_t w0, uint32_t total, uint64_t payloadBytes)
{
  const uint32_t lane = threadIdx.x;
  for (uint32_t c0 = lane * 16u; c0 < total; c0 += 8u * 1024u)
  {
    u32x4 v[8];
#pragma unroll
    for (uint32_t j = 0; j < 8u; j++)
    {
      const uint32_t c = c0 + j * 1024u;
      v[j] = (c < total && w0 + c + 16u <= payloadBytes + HSRLE_TAIL_PAD_BYTES) ? ld128(payload + w0 + c) : u32x4{ 0, 0, 0, 0 };
    }
#pragma unroll
    for (uint32_t j = 0; j < 8u; j++)
      if (c0 + j * 1024u < total) lds_st128(window + c0 + j * 1024u, v[j]);
  }
}

// ---- the same for a block container: one lane per block walks the block's stream from its first packet (known: no guessing, no
//      resolve pass) and writes the decoder state at every SB output bytes, so that k_decode_blocks can put B / SB lanes on a block.
//      What the block kernel checks in its prologue (table entry, stream header, mode byte) is checked here.
//      The streams of the NB consecutive blocks a wave walks are one contiguous piece of the payload: the wave copies it into LDS first
//      (coalesced 16-byte loads) and the lanes walk their chains from there -- a packet hop then costs an LDS round trip instead of a
//      global-memory latency (the 88 MB frame: 100-130 -> ~30 us).  A piece that does not fit the LDS window is walked from global memory.
template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_container_records(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ payloadEnd,
                                                          uint64_t U, uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t SB, uint32_t allowSingle,
                                                          uint32_t NB, uint32_t ldsBytes, uint32_t *__restrict__ rec, uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int KE = IndexState<FAM>::KE;
  constexpr int KS = KE > 0 ? KE : 1;
  extern __shared__ __attribute__((aligned(16))) uint8_t window[];
  const uint32_t lane = threadIdx.x;
  const uint32_t i0 = xcd_tile(blockIdx.x, gridDim.x) * NB;
  if (i0 >= blockCount) return;
  const uint32_t cnt = (blockCount - i0 < NB) ? blockCount - i0 : NB;
  const uint64_t payloadBytes = (uint64_t)(payloadEnd - payload);

  // the wave's piece of the payload -> LDS (only if the table entries are sane and the piece fits)
  const uint64_t w0raw = offsets[firstBlock + i0], w1 = offsets[firstBlock + i0 + cnt];
  const uint64_t w0 = w0raw & ~15ull;
  const bool staged = w0raw <= w1 && w1 <= payloadBytes && w1 - w0 + 96u <= (uint64_t)ldsBytes;   // (+ what the aligned reads of a parse near the end may touch)
  if (staged)
    stage_window(window, payload, w0, (uint32_t)(w1 - w0) + 64u, payloadBytes);   // (+ 64: what a parse may read behind the last packet; beyond the payload's 32-byte tail pad: zeros)
  wave_sync();
  if (lane >= cnt) return;

  const uint32_t b = firstBlock + i0 + lane;
  uint32_t err = 0;
  const uint64_t off0 = offsets[b], off1 = offsets[b + 1];
  const uint64_t start = (uint64_t)b * B;
  const uint32_t blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  if (off0 > off1 || off1 > payloadBytes || off1 - off0 > 0xFFFFFF00ull) err = DEC_ERR_HEADER;
  else
  {
    const uint8_t *const s = payload + off0;
    const uint32_t C = (uint32_t)(off1 - off0);
    uint32_t p0 = TR::kHeaderSize, sgl = 0, sym = 0;
    if (C < TR::kHeaderSize + 2u || ld32(s) != blen || ld32(s + 4) != C) err = DEC_ERR_HEADER;
    else if constexpr (TR::kShortSingle) { sym = s[8]; p0 = 9; }
    else if constexpr (S == 1 && !TR::kLut && !TR::kShort)
    {
      const uint32_t mode = s[8];
      if (mode == 1u) { if (allowSingle) { sgl = 1; sym = s[9]; p0 = 10; } else err = DEC_ERR_MODE; }
      else if (mode != 0u) err = DEC_ERR_MODE;
    }
    if (err == 0u)
    {
      uint32_t st[KS];
#pragma unroll
      for (int j = 0; j < KS; j++) st[j] = IDX_INIT | (uint32_t)j;
      uint64_t o = start;
      uint32_t how;
      if (staged && off0 >= w0 && off1 <= w1)
        how = walk_emit_records<FAM, S, AL>(LdsReader{ window, (uint32_t)(off0 - w0) }, off0, C, p0, 0xFFFFFFF0u, o, st, sgl != 0u, sym, start + blen, SB, rec);
      else
        how = walk_emit_records<FAM, S, AL>(GlobalReader{ s }, off0, C, p0, 0xFFFFFFF0u, o, st, sgl != 0u, sym, start + blen, SB, rec);
      if (how != 1u || o != start + blen) err = DEC_ERR_STREAM;
    }
  }
  if (err != 0u && status != nullptr) atomicOr(status, err);
}

// ---- PACKET LISTS: the split decode of small containers without records and without a second chain ----
// k_container_records + k_decode_blocks walk every packet chain twice (the record walk, then the sub-block lanes), and both walks are latency
// chains with a few thousand lanes on a device that holds half a million.  Only finding the packet boundaries is sequential; what a packet
// PRODUCES is known from its header alone.  So one lane per block hops through the block's packets once (k_container_packets; the block's
// stream staged in LDS as above) and leaves, per packet, one 64-bit entry
//     output position in the block (15 bits) | literal bytes (15) | stream offset of the literals (15) | symbol field (15)
// (symbol field: stream offset of the run's symbol -- the walk tracks the move-to-front lists as offsets, state_apply -- or kPktInitField | b
// for a symbol b b b ... the decoder starts with), closed by a sentinel entry whose output position is the end of what the list covers.
// k_expand_packets (hsrle_expand.hip.h) then builds the output with one lane per 16 OUTPUT bytes: binary search in the block's list, literal
// bytes straight from the stream, run bytes from the symbol -- no chain, whole lines stored.  Blocks of up to 16 KiB (15-bit fields).
// A block with more than B / 8 packets (fewer than 8 output bytes per packet on average) does not fit its list: the lane closes the list
// where it stands and decodes the rest of the block itself, byte by byte (slow, correct, and rare: adversarial inputs).
constexpr uint32_t kPktInitField = 0x7F00u;
constexpr uint32_t kPacketListMaxBlock = 16384u;
__host__ __device__ inline uint32_t packet_list_cap(uint32_t B) { return B / 8u + 2u; }
constexpr uint32_t kSubPacketList = 1u;          // what the codec tables' sub-block launchers take as "sub-block size" for this path
inline uint64_t packet_list_counts_bytes(uint32_t blockCount) { return ((uint64_t)blockCount * 4ull + 15ull) & ~15ull; }
__device__ __forceinline__ uint64_t pkt_entry(uint32_t outStart, uint32_t lit, uint32_t body, uint32_t symf)
{
  return (uint64_t)outStart | ((uint64_t)lit << 15) | ((uint64_t)body << 30) | ((uint64_t)symf << 45);
}

// returns 1 (the stream's last packet was walked; count entries written incl. the sentinel) or 2 (malformed)
template <int FAM, int S, int AL, typename READER>
__device__ __forceinline__ uint32_t walk_emit_packets(const READER &s, uint32_t C, uint32_t x, bool sgl, uint32_t blen, uint64_t *__restrict__ list, uint32_t cap,
                                                      uint32_t &count, uint8_t *__restrict__ out)
{
  using TR = Traits<FAM, S, AL>;
  constexpr int KE = IndexState<FAM>::KE;
  constexpr int KS = KE > 0 ? KE : 1;
  constexpr uint64_t kInit7 = 0x00FE807E01FF7F00ull;                    // rleX_Xsl.h:533-543, as in index_symbol
  uint32_t st[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) st[j] = IDX_INIT | (uint32_t)j;
  uint32_t curSym = IDX_INIT;
  uint32_t o = 0, n = 0;
  bool direct = false;                                                  // the list is full: this lane writes the output itself
  // the window of the NEXT packet is requested as soon as this packet's header says where it is; booking this packet (list state, entry,
  // store) then runs while that read is in flight
  uint64_t lo, hi, ex;
  s.template loadw<header_beyond_16<FAM, S, AL>()>(umin(x, C), lo, hi, ex);
  for (;;)
  {
    const Pkt k = parse_window<FAM, S, AL>(s, lo, hi, ex, x, C, sgl);
    if (k.bad || k.lit > blen - o || k.run > blen - o - k.lit) return 2u;
    s.template loadw<header_beyond_16<FAM, S, AL>()>(umin(x + k.used + k.lit, C), lo, hi, ex);   // (k.lit <= C - x - k.used: the packet was checked; a last packet's window is not used)
    state_apply<KE>(st, k.op, k.symAt);
    if (k.hasSym) curSym = k.symAt;
    uint32_t ref;
    if (sgl) ref = 9u;
    else if constexpr (TR::kShortSingle) ref = 8u;
    else if constexpr (KE > 0) ref = st[0];
    else ref = curSym;
    const uint32_t body = x + k.used;
    if (!direct && n + 2u > cap)
    {
      list[n++] = pkt_entry(o, 0u, 0u, 0u);                              // sentinel: the list covers [0, o)
      direct = true;
    }
    if (!direct)
    {
      const uint32_t initByte = (KE > 0 && FAM != PACKED) ? (uint32_t)(kInit7 >> (8u * ((ref & 15u) % 7u))) & 0xFFu : 0u;
      const uint32_t symf = ((ref >> 30) == 3u) ? (kPktInitField | initByte) : ref;
      list[n++] = pkt_entry(o, k.lit, body, symf);
    }
    else
    {
      const u32x4 sv = index_symbol<S>(s, ref, (KE > 0) ? (FAM == PACKED) : true);
      const uint32_t sw[4] = { sv.x, sv.y, sv.z, sv.w };
      for (uint32_t j = 0; j < k.lit; j++) out[o + j] = (uint8_t)(s.load32(body + j) & 0xFFu);
      for (uint32_t j = 0; j < k.run; j++) { const uint32_t q = j % (uint32_t)S; out[o + k.lit + j] = (uint8_t)(sw[q >> 2] >> (8u * (q & 3u))); }
    }
    o += k.lit + k.run;
    if (k.last) break;
    x = body + k.lit;
  }
  if (o != blen) return 2u;
  if (!direct) list[n++] = pkt_entry(o, 0u, 0u, 0u);
  count = n;
  return 1u;
}

template <int FAM, int S, int AL>
__global__ __launch_bounds__(64) void k_container_packets(const uint8_t *__restrict__ payload, const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ payloadEnd,
                                                          uint8_t *__restrict__ out, uint64_t U, uint32_t B, uint32_t firstBlock, uint32_t blockCount, uint32_t allowSingle,
                                                          uint32_t NB, uint32_t ldsBytes, uint64_t *__restrict__ lists, uint32_t *__restrict__ counts, uint32_t *__restrict__ status)
{
  using TR = Traits<FAM, S, AL>;
  extern __shared__ __attribute__((aligned(16))) uint8_t window[];
  const uint32_t lane = threadIdx.x;
  const uint32_t i0 = xcd_tile(blockIdx.x, gridDim.x) * NB;
  if (i0 >= blockCount) return;
  const uint32_t cnt = (blockCount - i0 < NB) ? blockCount - i0 : NB;
  const uint64_t payloadBytes = (uint64_t)(payloadEnd - payload);

  // the wave's piece of the payload -> LDS (as k_container_records)
  const uint64_t w0raw = offsets[firstBlock + i0], w1 = offsets[firstBlock + i0 + cnt];
  const uint64_t w0 = w0raw & ~15ull;
  const bool staged = w0raw <= w1 && w1 <= payloadBytes && w1 - w0 + 96u <= (uint64_t)ldsBytes;
#ifdef HSRLE_PKT_STAMPS
  const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
  if (staged)
    stage_window(window, payload, w0, (uint32_t)(w1 - w0) + 64u, payloadBytes);   // (+ 64: what a parse may read behind the last packet; beyond the payload's 32-byte tail pad: zeros)
  wave_sync();
#ifdef HSRLE_PKT_STAMPS
  const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
  if (lane >= cnt) return;

  const uint32_t local = i0 + lane, b = firstBlock + local;
  const uint32_t cap = packet_list_cap(B);
  uint32_t err = 0, n = 0;
  const uint64_t off0 = offsets[b], off1 = offsets[b + 1];
  const uint64_t start = (uint64_t)b * B;
  const uint32_t blen = (uint32_t)((U - start) < (uint64_t)B ? (U - start) : (uint64_t)B);
  if (off0 > off1 || off1 > payloadBytes || off1 - off0 >= (uint64_t)kPktInitField) err = DEC_ERR_HEADER;   // (a stream whose offsets do not fit the entries' 15 bits is not a block of <= 16 KiB)
  else
  {
    const uint8_t *const s = payload + off0;
    const uint32_t C = (uint32_t)(off1 - off0);
    uint32_t p0 = TR::kHeaderSize, sgl = 0;
    if (C < TR::kHeaderSize + 2u || ld32(s) != blen || ld32(s + 4) != C) err = DEC_ERR_HEADER;
    else if constexpr (TR::kShortSingle) { p0 = 9; }
    else if constexpr (S == 1 && !TR::kLut && !TR::kShort)
    {
      const uint32_t mode = s[8];
      if (mode == 1u) { if (allowSingle) { sgl = 1; p0 = 10; } else err = DEC_ERR_MODE; }
      else if (mode != 0u) err = DEC_ERR_MODE;
    }
    if (err == 0u)
    {
      uint64_t *const list = lists + (uint64_t)local * cap;
      uint32_t how;
      if (staged && off0 >= w0 && off1 <= w1)
        how = walk_emit_packets<FAM, S, AL>(LdsReader{ window, (uint32_t)(off0 - w0) }, C, p0, sgl != 0u, blen, list, cap, n, out + start);
      else
        how = walk_emit_packets<FAM, S, AL>(GlobalReader{ s }, C, p0, sgl != 0u, blen, list, cap, n, out + start);
      if (how != 1u) { err = DEC_ERR_STREAM; n = 0; }
    }
  }
  counts[local] = n;                                                     // (0: nothing of this block is expanded)
  if (err != 0u && status != nullptr) atomicOr(status, err);
#ifdef HSRLE_PKT_STAMPS
  // diagnostic build: per wave (its first lane) cycles of staging and of the walk, and the packets of the lane -> lists area of block 0 is NOT used: status + 4..
  if (status != nullptr)
  {
    const unsigned long long ts2 = __builtin_readcyclecounter();
    const unsigned long long am = __ballot(true);
    if (lane == (uint32_t)__builtin_ctzll(am)) { atomicAdd((unsigned long long *)(status + 2), ts1 - ts0); atomicAdd((unsigned long long *)(status + 4), ts2 - ts1); atomicAdd(status + 6, 1u); }
    atomicMax(status + 7, n);
    atomicAdd(status + 8, n);
  }
#endif
}

// host side: walk (records == 0) or record pass (records != 0) of one codec grammar
template <int FAM, int S, int AL>
inline hipError_t launch_index(const IndexArgs &a, int records, hipStream_t st)
{
  if (records)
    hipLaunchKernelGGL((k_index_records<FAM, S, AL>), dim3((a.R + 63u) / 64u), dim3(64), 0, st, a.stream, a.C, a.p0, a.G, a.R, a.single, a.singleSym, a.entry, a.outStart, a.stateIn,
                       a.U, a.B, a.rec, a.gate);
  else
  {
    const uint32_t n = a.list ? a.listCount : a.R;
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL((k_index_walk<FAM, S, AL>), dim3((n + 63u) / 64u), dim3(64), 0, st, a.stream, a.C, a.p0, a.G, a.M, a.R, a.single, a.list, a.listCount, a.fix, a.mark, a.roundTag, a.extMax,
                       a.g, a.e, a.olen, a.t);
  }
  return hipGetLastError();
}

template <int FAM, int S, int AL>
inline hipError_t launch_container_records(const DecodeArgs &a, uint32_t SB, uint32_t allowSingle, uint32_t *rec, hipStream_t st)
{
  // blocks per wave and LDS window: the average block stream x 1.5 per block, at most 48 KB (three waves per CU)
  const uint64_t payloadBytes = (uint64_t)(a.payloadEnd - a.payload);
  const uint64_t total = a.firstBlock + (uint64_t)a.blockCount;          // (the average over the whole container is what the host knows)
  const uint64_t avg = payloadBytes / (total ? total : 1u) + 16u;
  uint32_t NB = 64u;
  while (NB > 8u && (uint64_t)NB * avg * 3u / 2u > 49152ull) NB /= 2u;
  uint64_t lds = (uint64_t)NB * avg * 3u / 2u + 128u;
  lds = lds > 49152ull ? 49152ull : (lds < 4096ull ? 4096ull : lds);
  lds = (lds + 255ull) & ~255ull;
  hipLaunchKernelGGL((k_container_records<FAM, S, AL>), dim3((a.blockCount + NB - 1u) / NB), dim3(64), (uint32_t)lds, st, a.payload, a.offsets, a.payloadEnd, a.U, a.B, a.firstBlock,
                     a.blockCount, SB, allowSingle, NB, (uint32_t)lds, rec, a.status);
  return hipGetLastError();
}

// lists: blockCount x packet_list_cap(B) entries; counts: blockCount words (the caller's workspace)
template <int FAM, int S, int AL>
inline hipError_t launch_container_packets(const DecodeArgs &a, uint32_t allowSingle, uint64_t *lists, uint32_t *counts, hipStream_t st)
{
  const uint64_t payloadBytes = (uint64_t)(a.payloadEnd - a.payload);
  const uint64_t total = a.firstBlock + (uint64_t)a.blockCount;
  const uint64_t avg = payloadBytes / (total ? total : 1u) + 16u;
  uint32_t NB = 64u;
  while (NB > 8u && (uint64_t)NB * avg * 3u / 2u > 49152ull) NB /= 2u;
  uint64_t lds = (uint64_t)NB * avg * 3u / 2u + 128u;
  lds = lds > 49152ull ? 49152ull : (lds < 4096ull ? 4096ull : lds);
  lds = (lds + 255ull) & ~255ull;
  // The LDS window only pays while ALL waves are resident at once (3 per CU at 48 KB) and carry at least 32 chains each: beyond that the
  // walk runs in rounds of a latency chain, and 64 chains per wave straight from global memory (the lines of a stream are hit ~9 times
  // in a row) are faster -- 88 MB frame, rle64_3symlut_byte (21 MB payload): 103 us staged / 112 global; the same size run-distributed,
  // rle8_packed (48 MB): 171 / 98; 256 MiB: 283 / 183 and 418 / 186 (experiments/r04/call25.sh).
  if (NB < 32u || payloadBytes > (24ull << 20)) { NB = 64u; lds = 0; }
  hipLaunchKernelGGL((k_container_packets<FAM, S, AL>), dim3((a.blockCount + NB - 1u) / NB), dim3(64), (uint32_t)lds, st, a.payload, a.offsets, a.payloadEnd, a.out, a.U, a.B,
                     a.firstBlock, a.blockCount, allowSingle, NB, (uint32_t)lds, lists, counts, a.status);
  return hipGetLastError();
}

} // namespace hsrle

#ifdef HSRLE_EXPERIMENTS
#include "experiments/hsrle_decode_wave.hip.h"   // launch_decode_wave: one wave per block (measured slower than the split decode; experiment builds only)
#endif

namespace hsrle {

// what the codec tables hold as their "sub-block" entry: the record walk of the split decode; SB == 0 selects the wave-per-block decoder
// of the experiment builds (rec is not used)
template <int FAM, int S, int AL>
inline hipError_t launch_sub_or_wave(const DecodeArgs &a, uint32_t SB, uint32_t allowSingle, uint32_t *rec, hipStream_t st)
{
#ifdef HSRLE_EXPERIMENTS
  if (SB == 0u) return launch_decode_wave<FAM, S, AL>(a, allowSingle, st);
#else
  if (SB == 0u) return hipErrorNotSupported;                        // not in the shipped build
#endif
  if (SB == kSubPacketList)                                           // rec = [counts: blockCount words, padded to 16 bytes | lists]
    return launch_container_packets<FAM, S, AL>(a, allowSingle, (uint64_t *)((uint8_t *)rec + packet_list_counts_bytes(a.blockCount)), rec, st);
  return launch_container_records<FAM, S, AL>(a, SB, allowSingle, rec, st);
}

} // namespace hsrle
