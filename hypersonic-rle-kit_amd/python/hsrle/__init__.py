"""hsrle -- thin ctypes binding of libhsrle_hip.so (include/hsrle.h) for tests, bench.py and the multi-GPU driver.

PyTorch is used only as plumbing: device allocations (uint8 tensors), streams and torch.distributed.  Every codec call goes
through the C ABI into the hand-written gfx950 kernels; there is no Python or CPU implementation behind this module -- if the
shared library is missing or no GPU is present the calls raise.

The codec table mirrors the reference's plugin table (reference: src/codec_funcs.h:262-410): `CODECS[i]` is the name of the
(compress, decompress) pair with id i, e.g. "rle8_packed_multi", "rle64_3symlut_byte".
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get("HSRLE_LIB", os.path.join(PKG_DIR, "libhsrle_hip.so"))  # HSRLE_LIB: developer override for A/B builds

OK, ERR_ARGUMENT, ERR_CAPACITY, ERR_FORMAT, ERR_DEVICE, ERR_UNSUPPORTED = range(6)
SYNTH_RUNS, SYNTH_VIDEO = 0, 1
DEFAULT_BLOCK_SIZE = 4096
HEADER_SIZE = 64
TAIL_PAD = 32


class HsrleError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: {_lib().hsrle_status_string(status).decode()} ({status})")


class ContainerInfo(ctypes.Structure):
    _fields_ = [
        ("version", ctypes.c_uint32),
        ("codec", ctypes.c_uint32),
        ("uncompressedSize", ctypes.c_uint64),
        ("blockSize", ctypes.c_uint32),
        ("blockCount", ctypes.c_uint32),
        ("payloadSize", ctypes.c_uint64),
        ("totalSize", ctypes.c_uint64),
    ]

    @property
    def payload_start(self):
        return HEADER_SIZE + 8 * (self.blockCount + 1)


_LIB = None


def _lib():
    """Load libhsrle_hip.so (built by `make -C hypersonic-rle-kit_amd` / __graft_entry__.build()).  Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        # torch ships its own HIP runtime; load it first so that this library binds to the SAME runtime instance
        # (two runtimes in one process do not see each other's allocations)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR}` (hipcc, gfx950); there is no fallback codec")
    L = ctypes.CDLL(LIB_PATH)
    u8p, u32, u64, vp, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
    L.hsrle_codec_from_name.restype = ci
    L.hsrle_codec_from_name.argtypes = [ctypes.c_char_p]
    L.hsrle_codec_name.restype = ctypes.c_char_p
    L.hsrle_codec_name.argtypes = [ci]
    L.hsrle_status_string.restype = ctypes.c_char_p
    L.hsrle_status_string.argtypes = [ci]
    L.hsrle_version.restype = ctypes.c_char_p
    L.hsrle_device_count.restype = ci
    L.hsrle_experiments_enabled.restype = ci
    L.hsrle_encode_path.restype = ci
    L.hsrle_encode_path.argtypes = [ci, ctypes.c_uint64, ctypes.c_uint32]
    L.hsrle_build_id.restype = ctypes.c_char_p
    L.hsrle_suggest_block_size.restype = u32
    L.hsrle_suggest_block_size.argtypes = [u64]
    L.hsrle_kernel_waves_per_cu.restype = ci
    L.hsrle_kernel_waves_per_cu.argtypes = [ci, ci]
    L.rle_compress_bounds.restype = u32
    L.rle_compress_bounds.argtypes = [u32]
    L.rle_decompress_additional_size.restype = u32
    for nm in ("hsrle_compress_mono", "hsrle_decompress_mono"):
        f = getattr(L, nm)
        f.restype = u32
        f.argtypes = [ci, u8p, u32, u8p, u32]
    L.hsrle_container_bound.restype = u64
    L.hsrle_container_bound.argtypes = [u64, u32]
    L.hsrle_compress_workspace_size.restype = u64
    L.hsrle_compress_workspace_size.argtypes = [u64, u32]
    L.hsrle_compress_workspace_size_codec.restype = u64
    L.hsrle_compress_workspace_size_codec.argtypes = [ctypes.c_int, u64, u32]
    L.hsrle_compress_dev_async.restype = ci
    L.hsrle_compress_dev_async.argtypes = [ci, vp, u64, vp, u64, u32, vp, u64, vp]
    L.hsrle_compress_dev.restype = ci
    L.hsrle_compress_dev.argtypes = [ci, vp, u64, vp, u64, u32, ctypes.POINTER(u64), vp]
    L.hsrle_container_info_dev.restype = ci
    L.hsrle_container_info_dev.argtypes = [vp, u64, ctypes.POINTER(ContainerInfo), vp]
    L.hsrle_container_info_host.restype = ci
    L.hsrle_container_info_host.argtypes = [vp, u64, ctypes.POINTER(ContainerInfo)]
    L.hsrle_decompress_dev_async.restype = ci
    L.hsrle_decompress_dev_async.argtypes = [vp, ctypes.POINTER(ContainerInfo), vp, u64, vp, vp]
    L.hsrle_decompress_blocks_dev_async.restype = ci
    L.hsrle_decompress_blocks_dev_async.argtypes = [vp, ctypes.POINTER(ContainerInfo), u32, u32, vp, u64, vp, vp]
    L.hsrle_decompress_dev.restype = ci
    L.hsrle_decompress_dev.argtypes = [vp, u64, vp, u64, ctypes.POINTER(u64), vp]
    L.hsrle_compress_host.restype = ci
    L.hsrle_compress_host.argtypes = [ci, vp, u64, vp, u64, u32, ctypes.POINTER(u64)]
    L.hsrle_decompress_host.restype = ci
    L.hsrle_decompress_host.argtypes = [vp, u64, vp, u64, ctypes.POINTER(u64)]
    L.hsrle_decompress_mono_workspace_size.restype = u64
    L.hsrle_decompress_mono_workspace_size.argtypes = [ci, u32, u32]
    L.hsrle_decompress_mono_dev.restype = ci
    L.hsrle_decompress_mono_dev.argtypes = [ci, vp, u32, vp, u64, vp, u64, ctypes.POINTER(u32), ctypes.POINTER(u32), vp]
    L.hsrle_compress_mono_workspace_size.restype = u64
    L.hsrle_compress_mono_workspace_size.argtypes = [ci, u32]
    L.hsrle_decompress_mono_dev_async.restype = ci
    L.hsrle_decompress_mono_dev_async.argtypes = [ci, vp, ctypes.c_char_p, u32, vp, u64, vp, u64, ctypes.POINTER(u32), vp, vp]
    L.hsrle_compress_mono_dev.restype = ci
    L.hsrle_compress_mono_dev.argtypes = [ci, vp, u32, vp, u64, vp, u64, ctypes.POINTER(u32), ctypes.POINTER(u32), vp]
    L.hsrle_compress_mono_dev_async.restype = ci
    L.hsrle_compress_mono_dev_async.argtypes = [ci, vp, u32, vp, u64, vp, u64, vp, vp]
    L.hsrle_mono_tuning.restype = None
    L.hsrle_mono_tuning.argtypes = [u32, u32, u32]
    L.hsrle_mono_encode_stats.restype = None
    L.hsrle_mono_encode_stats.argtypes = [ctypes.POINTER(u32)]
    L.hsrle_split_sub_block_size.restype = u32
    L.hsrle_split_sub_block_size.argtypes = [ctypes.POINTER(ContainerInfo), u32]
    L.hsrle_decompress_split_workspace_size.restype = u64
    L.hsrle_decompress_split_workspace_size.argtypes = [ctypes.POINTER(ContainerInfo), u32, u32]
    L.hsrle_decompress_split_dev_async.restype = ci
    L.hsrle_decompress_split_dev_async.argtypes = [vp, ctypes.POINTER(ContainerInfo), u32, u32, vp, u64, vp, vp, u64, u32, vp]
    if hasattr(L, "hsrle_decompress_wave_dev_async"):                    # experiment builds only
        L.hsrle_decompress_wave_dev_async.restype = ci
        L.hsrle_decompress_wave_dev_async.argtypes = [vp, ctypes.POINTER(ContainerInfo), u32, u32, vp, u64, vp, vp]
    L.hsrle_hash_blocks_dev_async.restype = ci
    L.hsrle_hash_blocks_dev_async.argtypes = [vp, ctypes.POINTER(ContainerInfo), u32, u32, vp, vp]
    L.hsrle_synth_dev_async.restype = ci
    L.hsrle_synth_dev_async.argtypes = [ci, ci, u64, vp, u64, vp]
    _LIB = L
    return L


def lib():
    return _lib()


CODEC_COUNT = 110  # 50 extreme codecs + 44 Short + 15 Greedy encoders + rle8_single_short (include/hsrle.h)


def codec_names():
    L = _lib()
    return [L.hsrle_codec_name(i).decode() for i in range(CODEC_COUNT)]


def codec_id(name_or_id):
    if isinstance(name_or_id, int):
        return name_or_id
    cid = _lib().hsrle_codec_from_name(name_or_id.encode())
    if cid < 0:
        raise KeyError(name_or_id)
    return cid


def suggest_block_size(n):
    """Block size that gives a buffer of n bytes enough blocks to fill the GPU (hsrle_suggest_block_size)."""
    return int(_lib().hsrle_suggest_block_size(n))


def build_id():
    """The library's build id: a hash of its sources and build flags (Makefile); profiles/*_traffic.json is stamped with it."""
    return _lib().hsrle_build_id().decode()


PATH_RING, PATH_SPLIT, PATH_RUN_LIST, PATH_POSITION_PARALLEL = 0, 1, 2, 3


def encode_path(codec, size, block_size):
    """Which encoder `compress` uses for `size` bytes in blocks of `block_size` (include/hsrle.h: hsrle_encode_path; needs no device): PATH_RING
    (one lane per block), PATH_SPLIT (small containers, chunks inside the blocks) or PATH_RUN_LIST (small containers, a wave per block)."""
    cid = codec if isinstance(codec, int) else codec_id(codec)
    r = int(_lib().hsrle_encode_path(cid, size, block_size))
    if r < 0:
        raise ValueError("hsrle_encode_path: bad codec, size or block size")
    return r


def experiments_enabled():
    """True for -DHSRLE_EXPERIMENTS builds (developer knobs / kernels measured slower); the shipped library returns False."""
    return bool(_lib().hsrle_experiments_enabled())


def kernel_waves_per_cu(codec, decode=True):
    """Wavefronts of the codec's decode / encode kernel resident on one CU (from the HIP runtime's occupancy calculation)."""
    return int(_lib().hsrle_kernel_waves_per_cu(codec_id(codec), 1 if decode else 0))


def compress_bounds(n):
    return _lib().rle_compress_bounds(n)


def container_bound(n, block_size=DEFAULT_BLOCK_SIZE):
    return _lib().hsrle_container_bound(n, block_size)


def workspace_size(n, block_size=DEFAULT_BLOCK_SIZE, codec=None):
    """Bytes of workspace for compress_async; with a codec (key or id): hsrle_compress_workspace_size_codec (8 bit Single / 128 bit: small containers
    then take the split encode)."""
    if codec is None:
        return _lib().hsrle_compress_workspace_size(n, block_size)
    return _lib().hsrle_compress_workspace_size_codec(codec_id(codec), n, block_size)


# ----------------------------------------------------------------------------------------------------------------------
# drop-in (host pointer, monolithic stream) path


def mono_compress(codec, data):
    """Reference-compatible single stream (what `<codec>_compress` of rle.h returns).  bytes -> bytes | None (failure)."""
    data = bytes(data)
    n = len(data)
    cap = compress_bounds(n) if n else 0
    out = ctypes.create_string_buffer(max(cap, 1))
    size = _lib().hsrle_compress_mono(codec_id(codec), data, n, out, cap)
    return out.raw[:size] if size else None


def mono_decompress(codec, stream, out_size=None):
    stream = bytes(stream)
    if out_size is None:
        out_size = int.from_bytes(stream[:4], "little")
    out = ctypes.create_string_buffer(max(out_size, 1))
    size = _lib().hsrle_decompress_mono(codec_id(codec), stream, len(stream), out, out_size)
    return out.raw[:size] if size else None


def mono_tuning(block=0, region=0, lookback=0):
    """Knobs of the monolithic decode (hsrle_mono_tuning): any values give the same output; 0 = the library's choice."""
    _lib().hsrle_mono_tuning(block, region, lookback)


def mono_compress_dev(codec, src, dst=None, workspace=None, return_chunks=False):
    """ONE monolithic reference stream of the CUDA uint8 tensor `src`, written by many lanes (hsrle_compress_mono_dev; the multi-symbol
    codecs of every width and the 8 bit Single codecs: not Greedy, not rle8_single_short).  Returns the stream tensor."""
    import torch

    _check_u8_cuda(src, "src")
    cid = codec_id(codec)
    n = src.numel()
    if dst is None:
        dst = torch.empty(compress_bounds(n) + 64, dtype=torch.uint8, device=src.device)
    need = _lib().hsrle_compress_mono_workspace_size(cid, n)
    if need == 0:
        raise HsrleError(ERR_UNSUPPORTED, "hsrle_compress_mono_dev")
    if workspace is None or workspace.numel() < need:
        workspace = _scratch(need, src.device)
    size, chunks = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = _lib().hsrle_compress_mono_dev(cid, ctypes.c_void_p(src.data_ptr()), n, ctypes.c_void_p(dst.data_ptr()), dst.numel(), ctypes.c_void_p(workspace.data_ptr()),
                                        workspace.numel(), ctypes.byref(size), ctypes.byref(chunks), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_compress_mono_dev")
    return (dst[: size.value], chunks.value) if return_chunks else dst[: size.value]


def mono_compress_dev_async(codec, src, dst, workspace, size_out=None):
    """hsrle_compress_mono_dev_async (rle8_multi / rle8_packed_multi): enqueue the encode of ONE monolithic reference stream on the current stream; nothing
    synchronises (can be captured in a HIP graph).  dst (>= compress_bounds(n) bytes), workspace (>= hsrle_compress_mono_workspace_size) and size_out
    (uint32[1] or None) are CUDA tensors the caller owns; the stream's size is also in bytes 4 .. 7 of dst once the stream has run."""
    _check_u8_cuda(src, "src")
    rc = _lib().hsrle_compress_mono_dev_async(codec_id(codec), ctypes.c_void_p(src.data_ptr()), src.numel(), ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                              ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), ctypes.c_void_p(size_out.data_ptr() if size_out is not None else None), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_compress_mono_dev_async")


def mono_encode_stats():
    """Of this thread's last monolithic encode with a move-to-front-list codec: (repair rounds, wrong list guesses in round 0, 1, later)."""
    stats = (ctypes.c_uint32 * 4)()
    _lib().hsrle_mono_encode_stats(stats)
    return tuple(int(v) for v in stats)


def mono_decompress_dev(codec, stream_tensor, dst=None, workspace=None, return_stats=False):
    """Decode ONE monolithic reference stream that lives in device memory (uint8 CUDA tensor with >= 64 bytes of slack behind the
    stream's last byte, 128-byte aligned) into device memory.  Returns the output tensor (and (regions, rounds, rewalked, lookback))."""
    import torch

    _check_u8_cuda(stream_tensor, "stream")
    head = stream_tensor[:8].cpu().numpy().tobytes()
    usize, csize = int.from_bytes(head[:4], "little"), int.from_bytes(head[4:8], "little")
    if dst is None:
        dst = torch.empty(max(usize, 1), dtype=torch.uint8, device=stream_tensor.device)
    cid = codec_id(codec)
    need = _lib().hsrle_decompress_mono_workspace_size(cid, usize, csize)
    if workspace is None or workspace.numel() < need:
        workspace = _scratch(max(need, 256), stream_tensor.device)
    n = ctypes.c_uint32(0)
    stats = (ctypes.c_uint32 * 4)()
    rc = _lib().hsrle_decompress_mono_dev(cid, ctypes.c_void_p(stream_tensor.data_ptr()), csize, ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                          ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), ctypes.byref(n), stats, _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_mono_dev")
    return (dst[: n.value], tuple(stats)) if return_stats else dst[: n.value]


MONO_DONE, MONO_MALFORMED, MONO_NEEDS_REPAIR = 0, 1, 2


def mono_decompress_dev_async(codec, stream_tensor, header16, dst, workspace, status, stream_size=None):
    """hsrle_decompress_mono_dev_async: enqueue the decode of ONE monolithic reference stream on the current stream, nothing synchronises
    (can be captured in a HIP graph).  header16: the stream's first 16 bytes (host bytes); dst / workspace / status (uint32[1]) are CUDA
    tensors the caller owns.  Returns the uncompressed size; `status` holds MONO_DONE / MONO_MALFORMED / MONO_NEEDS_REPAIR once the stream
    has run."""
    _check_u8_cuda(stream_tensor, "stream")
    header16 = bytes(header16[:16]).ljust(16, b"\0")
    csize = int.from_bytes(header16[4:8], "little") if stream_size is None else int(stream_size)
    n = ctypes.c_uint32(0)
    rc = _lib().hsrle_decompress_mono_dev_async(codec_id(codec), ctypes.c_void_p(stream_tensor.data_ptr()), header16, csize, ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                                ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), ctypes.byref(n), ctypes.c_void_p(status.data_ptr()), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_mono_dev_async")
    return n.value


def mono_decompress_workspace_size(codec, usize, csize):
    return int(_lib().hsrle_decompress_mono_workspace_size(codec_id(codec), usize, csize))


def call_dropin(name, data, out_cap):
    """Call one of the rle.h-named exports directly, e.g. call_dropin("rle8_packed_multi_compress", data, cap)."""
    f = getattr(_lib(), name)
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    data = bytes(data)
    out = ctypes.create_string_buffer(max(out_cap, 1))
    size = f(data, len(data), out, out_cap)
    return size, out.raw[:size]


# ----------------------------------------------------------------------------------------------------------------------
# device-resident container path (torch tensors as device memory)


def _stream_ptr(stream=None):
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _scratch(n, device):
    """A workspace tensor.  HSRLE_POISON_WORKSPACE=1 (the test suite sets it) fills it with garbage first: the library must not rely on
    a workspace that happens to be zero."""
    import torch

    if os.environ.get("HSRLE_POISON_WORKSPACE") == "1":
        return torch.full((n,), 0xC3, dtype=torch.uint8, device=device)
    return torch.empty(n, dtype=torch.uint8, device=device)


def _check_u8_cuda(t, what):
    import torch

    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise TypeError(f"{what} must be a contiguous CUDA uint8 tensor")


def compress_async(codec, src, dst, block_size=DEFAULT_BLOCK_SIZE, workspace=None, stream=None):
    """Enqueue compression of `src` into the container buffer `dst` (capacity >= container_bound)."""
    _check_u8_cuda(src, "src")
    _check_u8_cuda(dst, "dst")
    wptr, wsize = (None, 0)
    if workspace is not None:
        _check_u8_cuda(workspace, "workspace")
        wptr, wsize = ctypes.c_void_p(workspace.data_ptr()), workspace.numel()
    rc = _lib().hsrle_compress_dev_async(codec_id(codec), ctypes.c_void_p(src.data_ptr()), src.numel(), ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                         block_size, wptr, wsize, _stream_ptr(stream))
    if rc != OK:
        raise HsrleError(rc, "hsrle_compress_dev_async")


def compress(codec, src, block_size=DEFAULT_BLOCK_SIZE, dst=None):
    """Compress a CUDA uint8 tensor; returns (container tensor trimmed to its size, ContainerInfo)."""
    import torch

    _check_u8_cuda(src, "src")
    if dst is None:
        dst = torch.empty(container_bound(src.numel(), block_size), dtype=torch.uint8, device=src.device)
    total = ctypes.c_uint64(0)
    rc = _lib().hsrle_compress_dev(codec_id(codec), ctypes.c_void_p(src.data_ptr()), src.numel(), ctypes.c_void_p(dst.data_ptr()), dst.numel(), block_size,
                                   ctypes.byref(total), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_compress_dev")
    container = dst[: total.value]
    return container, container_info(container)


def container_info(container):
    info = ContainerInfo()
    if hasattr(container, "is_cuda"):
        if container.is_cuda:
            rc = _lib().hsrle_container_info_dev(ctypes.c_void_p(container.data_ptr()), container.numel(), ctypes.byref(info), _stream_ptr())
        else:
            rc = _lib().hsrle_container_info_host(ctypes.c_void_p(container.data_ptr()), container.numel(), ctypes.byref(info))
    else:
        b = bytes(container)
        rc = _lib().hsrle_container_info_host(b, len(b), ctypes.byref(info))
    if rc != OK:
        raise HsrleError(rc, "hsrle_container_info")
    return info


def decompress_async(container, info, dst, status=None, first_block=0, block_count=None, stream=None):
    """Enqueue decompression (of a block range) of a device container into `dst`; `status` is an optional int32/uint32 CUDA tensor."""
    _check_u8_cuda(container, "container")
    _check_u8_cuda(dst, "dst")
    if block_count is None:
        block_count = info.blockCount - first_block
    sp = ctypes.c_void_p(status.data_ptr()) if status is not None else None
    rc = _lib().hsrle_decompress_blocks_dev_async(ctypes.c_void_p(container.data_ptr()), ctypes.byref(info), first_block, block_count,
                                                  ctypes.c_void_p(dst.data_ptr()), dst.numel(), sp, _stream_ptr(stream))
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_blocks_dev_async")


def hash_blocks(container, info, first_block=0, block_count=None):
    """64 bit hash of every block stream (hsrle_hash_blocks_dev_async) as an int64 CUDA tensor (bit pattern of the uint64 values)."""
    import torch

    _check_u8_cuda(container, "container")
    if block_count is None:
        block_count = info.blockCount - first_block
    out = torch.empty(block_count, dtype=torch.int64, device=container.device)
    rc = _lib().hsrle_hash_blocks_dev_async(ctypes.c_void_p(container.data_ptr()), ctypes.byref(info), first_block, block_count, ctypes.c_void_p(out.data_ptr()), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_hash_blocks_dev_async")
    return out


def split_workspace_size(info, block_count=None, sub_block=0):
    return int(_lib().hsrle_decompress_split_workspace_size(ctypes.byref(info), info.blockCount if block_count is None else block_count, sub_block))


def decompress_split_async(container, info, dst, workspace, status=None, sub_block=0, first_block=0, block_count=None, stream=None):
    """Split decode (hsrle_decompress_split_dev_async): one decode lane per `sub_block` output bytes instead of per block."""
    _check_u8_cuda(container, "container")
    _check_u8_cuda(dst, "dst")
    if block_count is None:
        block_count = info.blockCount - first_block
    sp = ctypes.c_void_p(status.data_ptr()) if status is not None else None
    wp, wn = (ctypes.c_void_p(workspace.data_ptr()), workspace.numel()) if workspace is not None else (None, 0)
    rc = _lib().hsrle_decompress_split_dev_async(ctypes.c_void_p(container.data_ptr()), ctypes.byref(info), first_block, block_count, ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                                 sp, wp, wn, sub_block, _stream_ptr(stream))
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_split_dev_async")


def decompress_wave_async(container, info, dst, status=None, first_block=0, block_count=None, stream=None):
    """Wave decode (hsrle_decompress_wave_dev_async): one wave per block.  Experiment builds only."""
    if not hasattr(_lib(), "hsrle_decompress_wave_dev_async"):
        raise HsrleError(ERR_UNSUPPORTED, "hsrle_decompress_wave_dev_async (not in the shipped build)")
    _check_u8_cuda(container, "container")
    _check_u8_cuda(dst, "dst")
    if block_count is None:
        block_count = info.blockCount - first_block
    sp = ctypes.c_void_p(status.data_ptr()) if status is not None else None
    rc = _lib().hsrle_decompress_wave_dev_async(ctypes.c_void_p(container.data_ptr()), ctypes.byref(info), first_block, block_count, ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                                sp, _stream_ptr(stream))
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_wave_dev_async")


def decompress(container, dst=None):
    import torch

    _check_u8_cuda(container, "container")
    info = container_info(container)
    if dst is None:
        dst = torch.empty(info.uncompressedSize, dtype=torch.uint8, device=container.device)
    n = ctypes.c_uint64(0)
    rc = _lib().hsrle_decompress_dev(ctypes.c_void_p(container.data_ptr()), container.numel(), ctypes.c_void_p(dst.data_ptr()), dst.numel(), ctypes.byref(n), _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_decompress_dev")
    return dst[: n.value]


def synth(kind, symbol_bytes, seed, size, device="cuda", out=None):
    """Deterministic synthetic workload generated on the device; same bytes as the C/python generators."""
    import torch

    if out is None:
        out = torch.empty(size, dtype=torch.uint8, device=device)
    rc = _lib().hsrle_synth_dev_async(kind, symbol_bytes, seed, ctypes.c_void_p(out.data_ptr()), size, _stream_ptr())
    if rc != OK:
        raise HsrleError(rc, "hsrle_synth_dev_async")
    return out


def split_container(container_bytes):
    """Host-side view of a container: (ContainerInfo, [block stream bytes ...])."""
    b = bytes(container_bytes)
    info = container_info(b)
    import struct

    offs = struct.unpack_from(f"<{info.blockCount + 1}Q", b, HEADER_SIZE)
    p0 = info.payload_start
    return info, [b[p0 + offs[i] : p0 + offs[i + 1]] for i in range(info.blockCount)]


# ----------------------------------------------------------------------------------------------------------------------
# rle8m: the reference's own GPU decode path (include/hsrle.h)


class Rle8mInfo(ctypes.Structure):
    _fields_ = [("compressedSize", ctypes.c_uint32), ("uncompressedSize", ctypes.c_uint32), ("sections", ctypes.c_uint32)]


def rle8m_info(stream_tensor):
    """Header fields of a device-resident rle8m stream (synchronises)."""
    _check_u8_cuda(stream_tensor, "stream")
    info = Rle8mInfo()
    L = _lib()
    L.hsrle_rle8m_info_dev.restype = ctypes.c_int
    L.hsrle_rle8m_info_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.hsrle_rle8m_info_dev(ctypes.c_void_p(stream_tensor.data_ptr()), stream_tensor.numel(), ctypes.byref(info), _stream_ptr())
    if rc != 0:
        raise HsrleError(rc, "hsrle_rle8m_info_dev")
    return info


def rle8m_decompress_async(stream_tensor, info, dst, status=None, stream=None):
    """Enqueue the decode of a device-resident rle8m stream into `dst` (uint8 CUDA tensor); `status` optional int32 tensor."""
    _check_u8_cuda(stream_tensor, "stream")
    _check_u8_cuda(dst, "dst")
    L = _lib()
    L.hsrle_rle8m_decompress_dev_async.restype = ctypes.c_int
    L.hsrle_rle8m_decompress_dev_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    sp = ctypes.c_void_p(status.data_ptr()) if status is not None else None
    rc = L.hsrle_rle8m_decompress_dev_async(ctypes.c_void_p(stream_tensor.data_ptr()), ctypes.byref(info), ctypes.c_void_p(dst.data_ptr()), dst.numel(), sp, _stream_ptr(stream))
    if rc != 0:
        raise HsrleError(rc, "hsrle_rle8m_decompress_dev_async")


def rle8m_compress_dropin(sections, data):
    """rle8m_compress(subSections, pIn, inSize, pOut, outSize) of the library (host pointers); returns the stream or None (the reference's 0)."""
    L = _lib()
    L.rle8m_compress_bounds.restype = ctypes.c_uint32
    L.rle8m_compress.restype = ctypes.c_uint32
    data = bytes(data)
    cap = L.rle8m_compress_bounds(ctypes.c_uint32(sections), ctypes.c_uint32(len(data)))
    out = ctypes.create_string_buffer(cap + 64)
    size = L.rle8m_compress(ctypes.c_uint32(sections), data, ctypes.c_uint32(len(data)), out, ctypes.c_uint32(cap))
    return out.raw[:size] if size else None


def rle8m_compress_async(src, sections, dst, workspace, status=None, stream=None):
    """Enqueue the rle8m encode of the uint8 CUDA tensor `src` into `dst` (capacity >= rle8m_compress_bounds) with `sections` sub-sections."""
    _check_u8_cuda(src, "src")
    _check_u8_cuda(dst, "dst")
    _check_u8_cuda(workspace, "workspace")
    L = _lib()
    L.hsrle_rle8m_compress_dev_async.restype = ctypes.c_int
    L.hsrle_rle8m_compress_dev_async.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    sp = ctypes.c_void_p(status.data_ptr()) if status is not None else None
    rc = L.hsrle_rle8m_compress_dev_async(ctypes.c_void_p(src.data_ptr()), src.numel(), sections, ctypes.c_void_p(dst.data_ptr()), dst.numel(),
                                          ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), sp, _stream_ptr(stream))
    if rc != 0:
        raise HsrleError(rc, "hsrle_rle8m_compress_dev_async")


def rle8m_workspace_size(in_size, sections):
    L = _lib()
    L.hsrle_rle8m_compress_workspace_size.restype = ctypes.c_uint64
    return int(L.hsrle_rle8m_compress_workspace_size(ctypes.c_uint32(in_size), ctypes.c_uint32(sections)))


def rle8m_bounds(sections, in_size):
    L = _lib()
    L.rle8m_compress_bounds.restype = ctypes.c_uint32
    return int(L.rle8m_compress_bounds(ctypes.c_uint32(sections), ctypes.c_uint32(in_size)))
