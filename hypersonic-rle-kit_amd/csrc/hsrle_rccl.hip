// hsrle_rccl.hip -- the one exchange step of the block-sharded path as C ABI (include/hsrle.h section 4):
// one container on the root from the containers the W ranks produced for their own block ranges (gatherv of [offset table | payload]
// over RCCL), and its inverse.  Replaces nothing in the reference (it is single device); the closest precedent is its sub-section
// container for rle8m (src/rle8_low_entropy_cpu.c:131-191).
//
//   exchange 1   ncclAllGather of (codec, blockSize, blockCount, payloadSize) per rank  ->  every rank knows every offset
//   exchange 2   grouped ncclSend (ranks) / ncclRecv (root) of the table and payload segments STRAIGHT into their final places; xGMI is
//                point to point, so the root receives on its W-1 links at once -- a direct gatherv, not a ring
//   fix-up       the offsets of a segment were relative to its rank's payload: one small kernel per rank adds the payload prefix
//
// RCCL is loaded on first use (dlopen of librccl.so.1): the codec entry points do not depend on it.
#include "../../include/hsrle.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>
#include <vector>

namespace {

struct Rccl
{
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclCommCount) commCount = nullptr;
  decltype(&ncclCommUserRank) commUserRank = nullptr;
  decltype(&ncclAllGather) allGather = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rcclOnce;

bool rccl_ready()
{
  std::call_once(g_rcclOnce, [] {
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
      if ((g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!g_rccl.lib) return;
#define HS_SYM(field, sym) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, #sym)
    HS_SYM(getUniqueId, ncclGetUniqueId); HS_SYM(commInitRank, ncclCommInitRank); HS_SYM(commDestroy, ncclCommDestroy); HS_SYM(commCount, ncclCommCount);
    HS_SYM(commUserRank, ncclCommUserRank); HS_SYM(allGather, ncclAllGather); HS_SYM(broadcast, ncclBroadcast); HS_SYM(send, ncclSend); HS_SYM(recv, ncclRecv);
    HS_SYM(groupStart, ncclGroupStart); HS_SYM(groupEnd, ncclGroupEnd);
#undef HS_SYM
    g_rccl.ok = g_rccl.getUniqueId && g_rccl.commInitRank && g_rccl.commDestroy && g_rccl.commCount && g_rccl.commUserRank && g_rccl.allGather && g_rccl.broadcast && g_rccl.send &&
                g_rccl.recv && g_rccl.groupStart && g_rccl.groupEnd;
  });
  return g_rccl.ok;
}

struct Header   // the 64-byte container header (include/hsrle.h)
{
  char magic[8];
  uint32_t version, codec;
  uint64_t uncompressedSize;
  uint32_t blockSize, blockCount;
  uint64_t payloadSize, totalSize;
  uint8_t reserved[16];
};
static_assert(sizeof(Header) == HSRLE_CONTAINER_HEADER_SIZE, "container header is 64 bytes");

Header make_header(uint32_t codec, uint64_t U, uint32_t B, uint32_t nBlocks, uint64_t payload)
{
  Header h;
  memcpy(h.magic, "HSRLEKIT", 8);
  h.version = 1; h.codec = codec; h.uncompressedSize = U; h.blockSize = B; h.blockCount = nBlocks; h.payloadSize = payload;
  h.totalSize = HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)nBlocks + 1ull) + payload + HSRLE_CONTAINER_TAIL_PAD;
  memset(h.reserved, 0, sizeof(h.reserved));
  return h;
}

// the same consistency rules as the decode path's check_info (hsrle_capi.hip): a header that passes describes a container whose offset
// table and payload lie inside `size` bytes
int check_header(const Header &h, uint64_t size)
{
  if (memcmp(h.magic, "HSRLEKIT", 8) != 0 || h.version != 1) return HSRLE_ERR_FORMAT;
  if (h.blockSize < HSRLE_MIN_BLOCK_SIZE || h.blockSize > HSRLE_MAX_BLOCK_SIZE || (h.blockSize % 128u) != 0 || h.uncompressedSize == 0) return HSRLE_ERR_FORMAT;
  if ((uint64_t)h.blockCount != (h.uncompressedSize + h.blockSize - 1) / h.blockSize) return HSRLE_ERR_FORMAT;
  const uint64_t payloadStart = HSRLE_CONTAINER_HEADER_SIZE + 8ull * ((uint64_t)h.blockCount + 1ull);
  if (h.totalSize != payloadStart + h.payloadSize + HSRLE_CONTAINER_TAIL_PAD || h.totalSize > size) return HSRLE_ERR_FORMAT;
  return HSRLE_OK;
}

// table[i] += delta for i in [0, n); optionally table[n] = last
__global__ void k_rebase_offsets(uint64_t *table, uint64_t n, uint64_t delta, int writeLast, uint64_t last)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) table[i] += delta;
  if (writeLast && i == 0) table[n] = last;
}

constexpr uint64_t kPiece = 1ull << 30;   // bytes per point-to-point message

bool p2p(bool sending, const void *buf, uint64_t bytes, int peer, ncclComm_t comm, hipStream_t st)
{
  for (uint64_t at = 0; at < bytes; at += kPiece)
  {
    const uint64_t n = bytes - at < kPiece ? bytes - at : kPiece;
    const ncclResult_t r = sending ? g_rccl.send((const uint8_t *)buf + at, n, ncclUint8, peer, comm, st) : g_rccl.recv((uint8_t *)buf + at, n, ncclUint8, peer, comm, st);
    if (r != ncclSuccess) return false;
  }
  return true;
}

void shard_blocks(uint64_t blockCount, int world, int rank, uint64_t *first, uint64_t *count)
{
  *first = (uint64_t)rank * blockCount / (uint64_t)world;
  *count = (uint64_t)(rank + 1) * blockCount / (uint64_t)world - *first;
}

} // namespace

extern "C" {

int hsrle_rccl_unique_id(void *id128)
{
  if (!id128) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.getUniqueId(&id) != ncclSuccess) return HSRLE_ERR_DEVICE;
  static_assert(sizeof(id) == HSRLE_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof(id));
  return HSRLE_OK;
}

int hsrle_rccl_comm_create(const void *id128, int worldSize, int rank, void **pComm)
{
  if (!id128 || !pComm || worldSize < 1 || rank < 0 || rank >= worldSize) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  if (g_rccl.commInitRank(&comm, worldSize, id, rank) != ncclSuccess) return HSRLE_ERR_DEVICE;
  *pComm = (void *)comm;
  return HSRLE_OK;
}

int hsrle_rccl_comm_ranks(void *comm, int *pWorldSize, int *pRank)
{
  if (!comm || !pWorldSize || !pRank) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  if (g_rccl.commCount((ncclComm_t)comm, pWorldSize) != ncclSuccess || g_rccl.commUserRank((ncclComm_t)comm, pRank) != ncclSuccess) return HSRLE_ERR_DEVICE;
  return HSRLE_OK;
}

int hsrle_rccl_comm_destroy(void *comm)
{
  if (!comm) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  return g_rccl.commDestroy((ncclComm_t)comm) == ncclSuccess ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

int hsrle_gather_container_rccl(void *pComm, int root, const void *dLocal, uint64_t localSize, uint64_t totalUncompressedSize, void *dOut, uint64_t outCapacity,
                                uint64_t *pTotalSize, void *stream)
{
  if (!pComm) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  ncclComm_t comm = (ncclComm_t)pComm;
  hipStream_t st = (hipStream_t)stream;
  int world = 0, rank = 0;
  if (g_rccl.commCount(comm, &world) != ncclSuccess || g_rccl.commUserRank(comm, &rank) != ncclSuccess || root < 0 || root >= world) return HSRLE_ERR_ARGUMENT;

  // this rank's (codec, blockSize, blockCount, payloadSize, what is wrong with its container, the room it offers as root); a rank without
  // blocks says blockCount 0.  NOTHING returns between here and the end of exchange 1 on one rank alone: every rank must reach the
  // all-gather, and every rank then draws the SAME verdict from the gathered words before any point-to-point transfer is posted -- a rank
  // that bailed out on its own (bad local header, root without room) used to leave its peers' streams waiting for ever
  constexpr int kWords = 6;
  uint64_t mine[kWords] = { ~0ull, 0, 0, 0, 0, (rank == root && dOut != nullptr) ? outCapacity : 0ull };
  if (dLocal != nullptr && localSize >= HSRLE_CONTAINER_HEADER_SIZE)
  {
    Header h;
    if (hipMemcpyAsync(&h, dLocal, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) mine[4] = (uint64_t)HSRLE_ERR_DEVICE;
    else if (check_header(h, localSize) != HSRLE_OK) mine[4] = (uint64_t)HSRLE_ERR_FORMAT;
    else { mine[0] = h.codec; mine[1] = h.blockSize; mine[2] = h.blockCount; mine[3] = h.payloadSize; }
  }
  else if (dLocal != nullptr)
    mine[4] = (uint64_t)HSRLE_ERR_FORMAT;

  // exchange 1
  uint64_t *dInfo = nullptr;
  const bool haveInfo = hipMallocAsync((void **)&dInfo, 8ull * kWords * (uint64_t)(world + 1), st) == hipSuccess;
  if (!haveInfo) { dInfo = nullptr; }
  std::vector<uint64_t> all(kWords * (size_t)world);
  bool ok = haveInfo && hipMemcpyAsync(dInfo, mine, 8 * kWords, hipMemcpyHostToDevice, st) == hipSuccess &&
            g_rccl.allGather(dInfo, dInfo + kWords, kWords, ncclUint64, comm, st) == ncclSuccess &&
            hipMemcpyAsync(all.data(), dInfo + kWords, 8ull * kWords * (uint64_t)world, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (dInfo) (void)hipFreeAsync(dInfo, st);
  if (!ok) return HSRLE_ERR_DEVICE;            // (a local allocation / collective failure: nothing a peer could be told about)

  // the verdict: the same on every rank
  uint64_t codec = ~0ull, blockSize = 0, nBlocks = 0, payload = 0;
  int verdict = HSRLE_OK;
  for (int r = 0; r < world; r++)
  {
    const uint64_t *v = &all[kWords * (size_t)r];
    if (v[4] != 0 && verdict == HSRLE_OK) verdict = (int)v[4];
    if (v[2] == 0) continue;
    if (codec == ~0ull) { codec = v[0]; blockSize = v[1]; }
    else if ((codec != v[0] || blockSize != v[1]) && verdict == HSRLE_OK) verdict = HSRLE_ERR_FORMAT;     // the ranks do not agree on codec / block size
    nBlocks += v[2]; payload += v[3];
  }
  if (verdict == HSRLE_OK && (codec == ~0ull || nBlocks > 0xFFFFFFF0ull)) verdict = HSRLE_ERR_FORMAT;
  if (verdict != HSRLE_OK) return verdict;
  const Header out = make_header((uint32_t)codec, totalUncompressedSize, (uint32_t)blockSize, (uint32_t)nBlocks, payload);
  if (pTotalSize) *pTotalSize = out.totalSize;
  if (all[kWords * (size_t)root + 5] < out.totalSize) return HSRLE_ERR_CAPACITY;       // the root's room (0 without an output buffer): known to every rank

  const uint64_t myCount = all[kWords * (size_t)rank + 2], myPayload = all[kWords * (size_t)rank + 3];
  const uint8_t *local = (const uint8_t *)dLocal;
  if (rank != root)
  {
    if (myCount == 0) return HSRLE_OK;
    if (g_rccl.groupStart() != ncclSuccess) return HSRLE_ERR_DEVICE;
    ok = p2p(true, local + HSRLE_CONTAINER_HEADER_SIZE, 8ull * myCount, root, comm, st) &&
         p2p(true, local + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (myCount + 1ull), myPayload, root, comm, st);
    return (g_rccl.groupEnd() == ncclSuccess && ok) ? HSRLE_OK : HSRLE_ERR_DEVICE;
  }

  uint8_t *o = (uint8_t *)dOut;
  uint64_t *table = (uint64_t *)(o + HSRLE_CONTAINER_HEADER_SIZE);
  uint8_t *pay = o + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (nBlocks + 1ull);
  if (hipMemcpyAsync(o, &out, sizeof(out), hipMemcpyHostToDevice, st) != hipSuccess || hipMemsetAsync(pay + payload, 0, HSRLE_CONTAINER_TAIL_PAD, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;

  // exchange 2: every segment straight into its final place
  if (g_rccl.groupStart() != ncclSuccess) return HSRLE_ERR_DEVICE;
  uint64_t blk = 0, at = 0;
  ok = true;
  for (int r = 0; r < world && ok; r++)
  {
    const uint64_t c = all[kWords * (size_t)r + 2], p = all[kWords * (size_t)r + 3];
    if (c != 0 && r != root)
      ok = p2p(false, table + blk, 8ull * c, r, comm, st) && p2p(false, pay + at, p, r, comm, st);
    blk += c; at += p;
  }
  if (g_rccl.groupEnd() != ncclSuccess || !ok) return HSRLE_ERR_DEVICE;

  blk = 0; at = 0;
  for (int r = 0; r < world; r++)
  {
    const uint64_t c = all[kWords * (size_t)r + 2], p = all[kWords * (size_t)r + 3];
    if (c != 0)
    {
      if (r == root)
      {
        if (hipMemcpyAsync(table + blk, local + HSRLE_CONTAINER_HEADER_SIZE, 8ull * c, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(pay + at, local + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (c + 1ull), p, hipMemcpyDeviceToDevice, st) != hipSuccess)
          return HSRLE_ERR_DEVICE;
      }
      // offsets were relative to rank r's payload: add the payload prefix of the ranks in front of it
      hipLaunchKernelGGL(k_rebase_offsets, dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, st, table + blk, c, at, 0, 0ull);
    }
    blk += c; at += p;
  }
  hipLaunchKernelGGL(k_rebase_offsets, dim3(1), dim3(64), 0, st, table, nBlocks, 0ull, 1, payload);
  // (`out` lives on this stack frame: the header copy above must have read it before we return)
  return (hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess) ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

int hsrle_scatter_container_rccl(void *pComm, int root, const void *dContainer, uint64_t containerSize, void *dLocal, uint64_t localCapacity, uint64_t *pLocalSize, void *stream)
{
  if (!pComm) return HSRLE_ERR_ARGUMENT;
  if (!rccl_ready()) return HSRLE_ERR_UNSUPPORTED;
  ncclComm_t comm = (ncclComm_t)pComm;
  hipStream_t st = (hipStream_t)stream;
  int world = 0, rank = 0;
  if (g_rccl.commCount(comm, &world) != ncclSuccess || g_rccl.commUserRank(comm, &rank) != ncclSuccess || root < 0 || root >= world) return HSRLE_ERR_ARGUMENT;

  // the root tells every rank the container's shape and the payload range of its blocks: meta = codec, blockSize, blockCount, U, the root's
  // verdict on the container, then the payload offset of every rank's first block and of the block behind the last rank's last one
  // (world + 1 values).  As in the gather, no rank returns on its own between here and the go / no-go exchange below.
  const size_t metaCount = 5 + (size_t)world + 1;
  std::vector<uint64_t> meta(metaCount, 0);
  const uint8_t *src = (const uint8_t *)dContainer;
  if (rank == root)
  {
    Header h;
    int bad = HSRLE_OK;
    if (!dContainer || containerSize < HSRLE_CONTAINER_HEADER_SIZE) bad = HSRLE_ERR_ARGUMENT;
    else if (hipMemcpyAsync(&h, dContainer, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) bad = HSRLE_ERR_DEVICE;
    else bad = check_header(h, containerSize);
    if (bad == HSRLE_OK)
    {
      meta[0] = h.codec; meta[1] = h.blockSize; meta[2] = h.blockCount; meta[3] = h.uncompressedSize;
      for (int r = 0; r <= world && bad == HSRLE_OK; r++)
      {
        uint64_t first, count;
        shard_blocks(h.blockCount, world, r < world ? r : world - 1, &first, &count);
        const uint64_t idx = r < world ? first : h.blockCount;
        if (hipMemcpyAsync(&meta[5 + (size_t)r], src + HSRLE_CONTAINER_HEADER_SIZE + 8ull * idx, 8, hipMemcpyDeviceToHost, st) != hipSuccess) bad = HSRLE_ERR_DEVICE;
      }
      if (bad == HSRLE_OK && hipStreamSynchronize(st) != hipSuccess) bad = HSRLE_ERR_DEVICE;
      // the table entries the shards are cut at: ascending and inside the payload (p1 - p0 below must not wrap)
      for (int r = 0; r <= world && bad == HSRLE_OK; r++)
        if (meta[5 + (size_t)r] > h.payloadSize || (r > 0 && meta[5 + (size_t)r] < meta[5 + (size_t)r - 1])) bad = HSRLE_ERR_FORMAT;
    }
    meta[4] = (uint64_t)bad;
  }
  uint64_t *dMeta = nullptr;
  if (hipMallocAsync((void **)&dMeta, 8ull * (metaCount + (size_t)world + 1), st) != hipSuccess) return HSRLE_ERR_DEVICE;
  bool ok = (rank != root || hipMemcpyAsync(dMeta, meta.data(), 8ull * metaCount, hipMemcpyHostToDevice, st) == hipSuccess) &&
            g_rccl.broadcast(dMeta, dMeta, metaCount, ncclUint64, root, comm, st) == ncclSuccess &&
            hipMemcpyAsync(meta.data(), dMeta, 8ull * metaCount, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (!ok) { (void)hipFreeAsync(dMeta, st); return HSRLE_ERR_DEVICE; }
  if (meta[4] != 0) { (void)hipFreeAsync(dMeta, st); return (int)meta[4]; }       // the root's verdict, the same on every rank

  const uint64_t nBlocks = meta[2], U = meta[3], B = meta[1];
  uint64_t first, count;
  shard_blocks(nBlocks, world, rank, &first, &count);
  const uint64_t p0 = meta[5 + (size_t)rank], p1 = meta[5 + (size_t)rank + 1];
  const uint64_t lo = first * B, hi = ((first + count) * B < U) ? (first + count) * B : U;
  const Header mineH = make_header((uint32_t)meta[0], hi > lo ? hi - lo : 0, (uint32_t)B, (uint32_t)count, p1 - p0);
  if (pLocalSize) *pLocalSize = count ? mineH.totalSize : 0;
  uint8_t *dst = (uint8_t *)dLocal;
  // go / no-go: has every rank the room for its part?  (all-gather of one word; a rank that returned HSRLE_ERR_CAPACITY on its own left
  // the root's send to it waiting for ever)
  {
    const uint64_t go = (count == 0 || (dLocal != nullptr && localCapacity >= mineH.totalSize)) ? 1ull : 0ull;
    std::vector<uint64_t> gos((size_t)world, 0);
    uint64_t *dGo = dMeta + metaCount;
    ok = hipMemcpyAsync(dGo, &go, 8, hipMemcpyHostToDevice, st) == hipSuccess && g_rccl.allGather(dGo, dGo + 1, 1, ncclUint64, comm, st) == ncclSuccess &&
         hipMemcpyAsync(gos.data(), dGo + 1, 8ull * (uint64_t)world, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFreeAsync(dMeta, st);
    if (!ok) return HSRLE_ERR_DEVICE;
    for (int r = 0; r < world; r++)
      if (gos[(size_t)r] == 0) return HSRLE_ERR_CAPACITY;
  }

  if (g_rccl.groupStart() != ncclSuccess) return HSRLE_ERR_DEVICE;
  ok = true;
  if (rank == root)
  {
    const uint8_t *table = src + HSRLE_CONTAINER_HEADER_SIZE, *pay = table + 8ull * (nBlocks + 1ull);
    for (int r = 0; r < world && ok; r++)
    {
      uint64_t f, c;
      shard_blocks(nBlocks, world, r, &f, &c);
      if (c == 0 || r == root) continue;
      ok = p2p(true, table + 8ull * f, 8ull * (c + 1ull), r, comm, st) && p2p(true, pay + meta[5 + (size_t)r], meta[5 + (size_t)r + 1] - meta[5 + (size_t)r], r, comm, st);
    }
  }
  else if (count != 0)
    ok = p2p(false, dst + HSRLE_CONTAINER_HEADER_SIZE, 8ull * (count + 1ull), root, comm, st) &&
         p2p(false, dst + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (count + 1ull), p1 - p0, root, comm, st);
  if (g_rccl.groupEnd() != ncclSuccess || !ok) return HSRLE_ERR_DEVICE;
  if (count == 0) return HSRLE_OK;

  if (rank == root)
  {
    const uint8_t *table = src + HSRLE_CONTAINER_HEADER_SIZE, *pay = table + 8ull * (nBlocks + 1ull);
    if (hipMemcpyAsync(dst + HSRLE_CONTAINER_HEADER_SIZE, table + 8ull * first, 8ull * (count + 1ull), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dst + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (count + 1ull), pay + p0, p1 - p0, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return HSRLE_ERR_DEVICE;
  }
  if (hipMemcpyAsync(dst, &mineH, sizeof(mineH), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(dst + HSRLE_CONTAINER_HEADER_SIZE + 8ull * (count + 1ull) + (p1 - p0), 0, HSRLE_CONTAINER_TAIL_PAD, st) != hipSuccess)
    return HSRLE_ERR_DEVICE;
  // renumber: offsets relative to this rank's own payload
  hipLaunchKernelGGL(k_rebase_offsets, dim3((uint32_t)((count + 1 + 255) / 256)), dim3(256), 0, st, (uint64_t *)(dst + HSRLE_CONTAINER_HEADER_SIZE), count + 1ull, 0ull - p0, 0, 0ull);
  // (mineH lives on this stack frame: the copy above must have read it before we return)
  return (hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess) ? HSRLE_OK : HSRLE_ERR_DEVICE;
}

} // extern "C"
