"""Multi-GPU sharding of the block container: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).

The path shards by INDEPENDENT UNITS: every block is a complete reference stream, so rank r simply owns the contiguous block
range [r*n/W, (r+1)*n/W) -- encode and decode need no data-path collective at all.  The only exchange step is assembling ONE
container from the per-rank segments (or cutting one container into per-rank segments):

    gather_container():  all_gather of the per-rank payload sizes (W x int64)  ->  every rank knows every payload offset
                         point-to-point send (ranks) / recv (root) of [offset table | payload]  =  gatherv over xGMI:
                         the root receives on its W-1 distinct links at the same time, which is why a direct gatherv
                         beats a ring for this one-to-one-root pattern
    scatter_container(): the inverse (root -> ranks), so that a container produced elsewhere can be decoded by W GPUs

Both are codec agnostic: they move bytes and fix up the offset table.  The reference has nothing comparable (it is single
threaded and single device; its only sharded format, rle8m, is decoded by one OpenCL device: src/rle8_ocl.c:265-404).
"""
import struct

import torch
import torch.distributed as dist

HEADER_SIZE = 64
TAIL_PAD = 32
MAGIC = b"HSRLEKIT"


def shard_blocks(block_count, world_size, rank):
    """Contiguous block range of `rank`: (first, count)."""
    first = rank * block_count // world_size
    last = (rank + 1) * block_count // world_size
    return first, last - first


def shard_bytes(total_size, block_size, world_size, rank):
    """Byte range of the uncompressed buffer that `rank` owns: (offset, size)."""
    nb = (total_size + block_size - 1) // block_size
    first, count = shard_blocks(nb, world_size, rank)
    lo = first * block_size
    hi = min((first + count) * block_size, total_size)
    return lo, max(hi - lo, 0)


def pack_header(codec, uncompressed_size, block_size, block_count, payload_size):
    total = HEADER_SIZE + 8 * (block_count + 1) + payload_size + TAIL_PAD
    return MAGIC + struct.pack("<IIQIIQQ", 1, codec, uncompressed_size, block_size, block_count, payload_size, total) + bytes(16)


def unpack_header(raw):
    raw = bytes(raw[:HEADER_SIZE])
    if raw[:8] != MAGIC:
        raise ValueError("not an hsrle container")
    version, codec, usize, bsize, bcount, psize, total = struct.unpack_from("<IIQIIQQ", raw, 8)
    return {"version": version, "codec": codec, "uncompressedSize": usize, "blockSize": bsize, "blockCount": bcount, "payloadSize": psize, "totalSize": total}


def _header_of(container):
    return unpack_header(container[:HEADER_SIZE].cpu().numpy().tobytes())


P2P_PIECE = 1 << 30  # bytes per point-to-point message: keeps every message count below 2^31 whatever the transport does with it


def _pieces(op, tensor, peer, group):
    """One P2POp per <= 1 GiB piece of a 1-D uint8 tensor (sender and receiver cut identically: both know the size)."""
    n = tensor.numel()
    return [dist.P2POp(op, tensor[at : min(at + P2P_PIECE, n)], peer, group) for at in range(0, n, P2P_PIECE)]


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def gather_container(local_container, total_uncompressed_size, root=0, group=None):
    """Assemble ONE container on `root` from the containers every rank produced for its own block range.

    local_container: uint8 tensor (device of the backend) holding this rank's container (may be None / empty if the rank
    owns no blocks).  Returns the assembled container on root, None on the other ranks.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = local_container.device if local_container is not None else torch.device("cpu")

    if local_container is not None and local_container.numel() >= HEADER_SIZE:
        h = _header_of(local_container)
        mine = [h["codec"], h["blockSize"], h["blockCount"], h["payloadSize"]]
    else:
        h, mine = None, [-1, 0, 0, 0]

    # exchange step 1: everybody learns everybody's block count and payload size
    info = torch.tensor(mine, dtype=torch.int64, device=dev)
    infos = [torch.zeros_like(info) for _ in range(world)]
    dist.all_gather(infos, info, group=group)
    infos = [t.cpu().tolist() for t in infos]
    codec = max(i[0] for i in infos)
    block_size = max(i[1] for i in infos)
    if any(i[2] > 0 and (i[0] != codec or i[1] != block_size) for i in infos):
        raise ValueError("gather_container: the ranks do not agree on codec / block size")
    counts = [i[2] for i in infos]
    psizes = [i[3] for i in infos]
    nblocks, payload = sum(counts), sum(psizes)
    table_at, payload_at = HEADER_SIZE, HEADER_SIZE + 8 * (nblocks + 1)

    if rank != root:
        if h is not None and counts[rank] > 0:
            lt = HEADER_SIZE
            lp = HEADER_SIZE + 8 * (counts[rank] + 1)
            _p2p(_pieces(dist.isend, local_container[lt : lt + 8 * counts[rank]], root, group)
                 + _pieces(dist.isend, local_container[lp : lp + psizes[rank]], root, group))
        return None

    out = torch.zeros(payload_at + payload + TAIL_PAD, dtype=torch.uint8, device=dev)
    out[:HEADER_SIZE] = torch.frombuffer(bytearray(pack_header(codec, total_uncompressed_size, block_size, nblocks, payload)), dtype=torch.uint8).to(dev)

    # exchange step 2: gatherv of [offset table | payload] straight into their final places
    ops, blk, pay = [], 0, 0
    for r in range(world):
        if counts[r] > 0:
            tdst = out[table_at + 8 * blk : table_at + 8 * (blk + counts[r])]
            pdst = out[payload_at + pay : payload_at + pay + psizes[r]]
            if r == root:
                lt, lp = HEADER_SIZE, HEADER_SIZE + 8 * (counts[r] + 1)
                tdst.copy_(local_container[lt : lt + 8 * counts[r]])
                pdst.copy_(local_container[lp : lp + psizes[r]])
            else:
                ops += _pieces(dist.irecv, tdst, r, group) + _pieces(dist.irecv, pdst, r, group)
        blk += counts[r]
        pay += psizes[r]
    _p2p(ops)

    # offsets were relative to each rank's payload: add the payload prefix of the owning rank
    table = out[table_at : table_at + 8 * (nblocks + 1)].view(torch.int64)
    blk, pay = 0, 0
    for r in range(world):
        if counts[r] > 0:
            table[blk : blk + counts[r]] += pay
        blk += counts[r]
        pay += psizes[r]
    table[nblocks] = payload
    return out


def scatter_container(container, root=0, device=None, group=None):
    """Inverse of gather_container: cut the root's container into one container per rank (contiguous block ranges).

    Returns this rank's container (its blocks renumbered from 0, uncompressedSize = its byte range), or None if it owns no block.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank == root:
        dev = container.device
        h = _header_of(container)
        meta = torch.tensor([h["codec"], h["blockSize"], h["blockCount"], h["uncompressedSize"]], dtype=torch.int64, device=dev)
    else:
        dev = device if device is not None else torch.device("cpu")
        meta = torch.zeros(4, dtype=torch.int64, device=dev)
    dist.broadcast(meta, root, group=group)
    codec, block_size, nblocks, usize = meta.cpu().tolist()
    table_at, payload_at = HEADER_SIZE, HEADER_SIZE + 8 * (nblocks + 1)

    # the root tells every rank the payload range of its blocks
    ranges = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    if rank == root:
        table = container[table_at : table_at + 8 * (nblocks + 1)].view(torch.int64).cpu()
        for r in range(world):
            first, count = shard_blocks(nblocks, world, r)
            ranges[2 * r] = int(table[first])
            ranges[2 * r + 1] = int(table[first + count])
    dist.broadcast(ranges, root, group=group)
    ranges = ranges.cpu().tolist()

    first, count = shard_blocks(nblocks, world, rank)
    p0, p1 = ranges[2 * rank], ranges[2 * rank + 1]
    lo, size = shard_bytes(usize, block_size, world, rank)
    local = None
    if count > 0:
        local = torch.zeros(HEADER_SIZE + 8 * (count + 1) + (p1 - p0) + TAIL_PAD, dtype=torch.uint8, device=dev)
        local[:HEADER_SIZE] = torch.frombuffer(bytearray(pack_header(codec, size, block_size, count, p1 - p0)), dtype=torch.uint8).to(dev)

    ops = []
    if rank == root:
        for r in range(world):
            f, c = shard_blocks(nblocks, world, r)
            if c == 0:
                continue
            tsrc = container[table_at + 8 * f : table_at + 8 * (f + c + 1)]
            psrc = container[payload_at + ranges[2 * r] : payload_at + ranges[2 * r + 1]]
            if r == root:
                local[HEADER_SIZE : HEADER_SIZE + 8 * (c + 1)].copy_(tsrc)
                local[HEADER_SIZE + 8 * (c + 1) : HEADER_SIZE + 8 * (c + 1) + (p1 - p0)].copy_(psrc)
            else:
                ops += _pieces(dist.isend, tsrc, r, group) + _pieces(dist.isend, psrc, r, group)
    elif count > 0:
        ops += _pieces(dist.irecv, local[HEADER_SIZE : HEADER_SIZE + 8 * (count + 1)], root, group)
        ops += _pieces(dist.irecv, local[HEADER_SIZE + 8 * (count + 1) : HEADER_SIZE + 8 * (count + 1) + (p1 - p0)], root, group)
    _p2p(ops)

    if local is not None:
        t = local[HEADER_SIZE : HEADER_SIZE + 8 * (count + 1)].view(torch.int64)
        t -= p0
    return local


# ----------------------------------------------------------------------------------------------------------------------
# the same two operations through the library's C ABI (include/hsrle.h section 4: hsrle_gather_container_rccl /
# hsrle_scatter_container_rccl over a communicator the library creates with ncclCommInitRank).  torch.distributed only carries the
# 128-byte unique id to the ranks.  bench.py takes this path for its gather step with HSRLE_DIST_C=1; the torch path above is the one the
# CPU tests (gloo) cover.

_C_COMMS = {}


def c_comm(group=None):
    """ncclComm_t (as int) of the library for `group`, created on first use on the current CUDA device."""
    import ctypes

    import hsrle

    key = id(group)
    if key in _C_COMMS:
        return _C_COMMS[key]
    L = hsrle.lib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ident = (ctypes.c_uint8 * 128)()
    if rank == 0:
        rc = L.hsrle_rccl_unique_id(ident)
        if rc != 0:
            raise hsrle.HsrleError(rc, "hsrle_rccl_unique_id")
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor(list(ident), dtype=torch.uint8, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)   # (src is a GLOBAL rank: the group's first member)
    ident = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
    comm = ctypes.c_void_p()
    rc = L.hsrle_rccl_comm_create(ident, world, rank, ctypes.byref(comm))
    if rc != 0:
        raise hsrle.HsrleError(rc, "hsrle_rccl_comm_create")
    _C_COMMS[key] = comm
    if len(_C_COMMS) == 1:
        import atexit

        atexit.register(destroy_c_comms)
    return comm


def c_comm_ranks(group=None):
    """(ranks, this rank) as the library's communicator itself reports them (ncclCommCount / ncclCommUserRank through hsrle_rccl_comm_ranks)."""
    import ctypes

    import hsrle

    L = hsrle.lib()
    L.hsrle_rccl_comm_ranks.restype = ctypes.c_int
    L.hsrle_rccl_comm_ranks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    world, rank = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.hsrle_rccl_comm_ranks(c_comm(group), ctypes.byref(world), ctypes.byref(rank))
    if rc != 0:
        raise hsrle.HsrleError(rc, "hsrle_rccl_comm_ranks")
    return world.value, rank.value


def destroy_c_comms():
    """hsrle_rccl_comm_destroy for every communicator c_comm() made (call before destroy_process_group; also runs at exit)."""
    import hsrle

    while _C_COMMS:
        _, comm = _C_COMMS.popitem()
        try:
            hsrle.lib().hsrle_rccl_comm_destroy(comm)
        except Exception:
            pass


def gather_container_c(local_container, total_uncompressed_size, root=0, group=None):
    """gather_container through hsrle_gather_container_rccl.  Returns the assembled container on root, None elsewhere."""
    import ctypes

    import hsrle

    L = hsrle.lib()
    L.hsrle_gather_container_rccl.restype = ctypes.c_int
    L.hsrle_gather_container_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    comm = c_comm(group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    n = local_container.numel() if local_container is not None else 0
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    cap = sum(int(x.item()) for x in sizes)                               # the parts' headers / tail pads make this an upper bound
    out = torch.empty(cap, dtype=torch.uint8, device=dev) if rank == root else None
    total = ctypes.c_uint64(0)
    rc = L.hsrle_gather_container_rccl(comm, root, ctypes.c_void_p(local_container.data_ptr()) if n else None, n, total_uncompressed_size,
                                       ctypes.c_void_p(out.data_ptr()) if out is not None else None, cap, ctypes.byref(total),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise hsrle.HsrleError(rc, "hsrle_gather_container_rccl")
    return out[: total.value] if out is not None else None


def scatter_container_c(container, shard_capacity, root=0, group=None):
    """scatter_container through hsrle_scatter_container_rccl; shard_capacity >= hsrle.container_bound(shard bytes, block size)."""
    import ctypes

    import hsrle

    L = hsrle.lib()
    L.hsrle_scatter_container_rccl.restype = ctypes.c_int
    L.hsrle_scatter_container_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                               ctypes.c_void_p]
    comm = c_comm(group)
    rank = dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    local = torch.empty(shard_capacity, dtype=torch.uint8, device=dev)
    size = ctypes.c_uint64(0)
    rc = L.hsrle_scatter_container_rccl(comm, root, ctypes.c_void_p(container.data_ptr()) if rank == root else None, container.numel() if rank == root else 0,
                                        ctypes.c_void_p(local.data_ptr()), shard_capacity, ctypes.byref(size), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise hsrle.HsrleError(rc, "hsrle_scatter_container_rccl")
    return local[: size.value] if size.value else None
