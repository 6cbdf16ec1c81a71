"""Device-resident throughput of the unsectioned low-entropy codecs, the reference CPU codec beside it:
    python tools/low_entropy_bench.py [size_mib]
One line per variant and data kind: ratio, GPU encode / decode GiB/s (HIP events, stream and buffers in HBM), CPU encode / decode GiB/s
(compiled reference when present, one core, first 64 MiB), stream == the CPU's on that prefix."""
import ctypes, os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import Oracle, Reference, REF_SO
size = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
L = hsrle.lib()
L.hsrle_low_entropy_workspace_size.restype = ctypes.c_uint64
L.hsrle_low_entropy_decompress_workspace_size.restype = ctypes.c_uint64
L.hsrle_low_entropy_decompress_workspace_size.argtypes = [ctypes.c_uint64]
vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
L.hsrle_low_entropy_compress_dev_async.argtypes = [vp, u32, ctypes.c_int, vp, u64, vp, u64, vp, vp]
L.hsrle_low_entropy_decompress_dev.argtypes = [vp, u64, vp, u64, vp, u64, ctypes.POINTER(u32), vp]
cpu = Reference() if os.path.exists(REF_SO) else Oracle()
names = ("rle8_low_entropy", "rle8_low_entropy_short", "rle8_low_entropy_only_max_frequency", "rle8_low_entropy_short_only_max_frequency")
print("library build", hsrle.build_id(), "| %d MiB, device resident | CPU: %s, one core, first 64 MiB" % (size >> 20, "compiled reference" if isinstance(cpu, Reference) else "oracle"))
for kind, kname in ((1, "video"), (0, "runs"), (2, "zeros")):   # zeros: ONE run of a flagged symbol -- cut inside the run since round 6 (k_le_cuts)
    src = hsrle.synth(kind, 1, 2, size) if kind < 2 else torch.zeros(size, dtype=torch.uint8, device="cuda")
    dst = torch.empty(size + 297, dtype=torch.uint8, device="cuda")
    ws = torch.empty(L.hsrle_low_entropy_workspace_size(size), dtype=torch.uint8, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = torch.empty(size, dtype=torch.uint8, device="cuda")
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sample = src[: 64 << 20].cpu().numpy().tobytes()
    for variant in range(4):
        enc = lambda: L.hsrle_low_entropy_compress_dev_async(src.data_ptr(), size, variant, dst.data_ptr(), dst.numel(), ws.data_ptr(), ws.numel(), status.data_ptr(), sp)
        assert enc() == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): enc()
        e1.record(); torch.cuda.synchronize()
        enc_ms = e0.elapsed_time(e1) / 5
        csize = int(dst[:4].view(torch.int32).item())
        dws = torch.empty(L.hsrle_low_entropy_decompress_workspace_size(csize), dtype=torch.uint8, device="cuda")
        got = u32(0)
        dec = lambda: L.hsrle_low_entropy_decompress_dev(dst.data_ptr(), csize, out.data_ptr(), out.numel(), dws.data_ptr(), dws.numel(), ctypes.byref(got), sp)
        assert dec() == 0
        t0 = time.perf_counter()
        for _ in range(5): dec()
        dec_ms = (time.perf_counter() - t0) / 5 * 1e3
        ok = int(status[0].item()) == 0 and got.value == size and torch.equal(out, src)
        t0 = time.perf_counter(); cst = cpu.low_entropy_compress(variant, sample); t1 = time.perf_counter()
        back = cpu.low_entropy_decompress(variant & 1, cst, len(sample)) if isinstance(cpu, Reference) else cpu.low_entropy_decompress(cst, len(sample)); t2 = time.perf_counter()
        # the GPU stream of the same prefix
        s2 = torch.frombuffer(bytearray(sample), dtype=torch.uint8).cuda()
        d2 = torch.empty(len(sample) + 297, dtype=torch.uint8, device="cuda")
        assert L.hsrle_low_entropy_compress_dev_async(s2.data_ptr(), len(sample), variant, d2.data_ptr(), d2.numel(), ws.data_ptr(), ws.numel(), status.data_ptr(), sp) == 0
        torch.cuda.synchronize()
        same = d2[: int(d2[:4].view(torch.int32).item())].cpu().numpy().tobytes() == cst
        print("%-42s %-5s ratio %.4f | GPU enc %7.1f dec %7.1f GiB/s | CPU enc %5.2f dec %5.2f GiB/s | round trip %s, stream == CPU %s" % (
            names[variant], kname, csize / size, size / 2**30 / (enc_ms * 1e-3), size / 2**30 / (dec_ms * 1e-3), (64 / 1024) / (t1 - t0), (64 / 1024) / (t2 - t1), ok, same and back == sample), flush=True)
        del dws
