"""hsrle.dist over RCCL (backend "nccl") on the hardware that is there: a world of ONE rank on a 1-GPU box.
It runs in a child process so that the process group never meets pytest's own process.  Covered: RCCL init, the all_gather of the
per-rank (blocks, payload) pairs, gather == the local container, scatter(gather(x)) == x, decode of the scattered container, and the
point-to-point piece path (a transfer above 1 GiB is cut into <= 1 GiB messages: hsrle/dist.py `_pieces`) as a self send/recv."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, socket, sys, time
sys.path.insert(0, os.path.join(%(repo)r, "hypersonic-rle-kit_amd", "python"))
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
import hsrle
from hsrle import dist as hd
res = {}
size = (1 << 30) + (1 << 28) + 4096 * 3 + 77          # payload of the incompressible buffer > 1 GiB: two pieces
for name, kind in (("runs", hsrle.SYNTH_RUNS), ("noise", None)):
    if kind is None:
        g = torch.Generator(device=dev); g.manual_seed(7)
        src = torch.randint(0, 256, (size,), dtype=torch.uint8, device=dev, generator=g)
    else:
        src = hsrle.synth(kind, 1, 101, size, device=dev)
    container, info = hsrle.compress("rle8_packed_multi", src, block_size=4096)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    full = hd.gather_container(container, size, root=0)
    torch.cuda.synchronize(); res["gather_ms_" + name] = (time.perf_counter() - t0) * 1e3
    assert torch.equal(full, container[: full.numel()]), "gather of one rank differs from its own container"
    back = hd.scatter_container(full, root=0)
    assert torch.equal(back, full), "scatter(gather(x)) != x"
    out = hsrle.decompress(back)
    assert torch.equal(out, src)
    res["payload_" + name] = int(info.payloadSize)
    # the point-to-point path on one device: a self send/recv of the whole payload, cut into <= 1 GiB pieces like gather/scatter do
    p0 = info.payload_start
    pay = container[p0 : p0 + info.payloadSize]
    dst = torch.zeros_like(pay)
    ops = hd._pieces(dist.isend, pay, 0, None) + hd._pieces(dist.irecv, dst, 0, None)
    res["pieces_" + name] = len(ops) // 2
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hd._p2p(ops)
    torch.cuda.synchronize(); res["self_p2p_ms_" + name] = (time.perf_counter() - t0) * 1e3
    assert torch.equal(dst, pay), "self send/recv changed the payload"
    # the same through the library's C ABI (hsrle_gather_container_rccl / hsrle_scatter_container_rccl over its own communicator)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fullc = hd.gather_container_c(container, size, root=0)
    torch.cuda.synchronize(); res["c_gather_ms_" + name] = (time.perf_counter() - t0) * 1e3
    assert torch.equal(fullc, full), "C-ABI gather differs from the torch gather"
    backc = hd.scatter_container_c(fullc, fullc.numel(), root=0)
    torch.cuda.synchronize()
    assert torch.equal(backc, full), "C-ABI scatter(gather(x)) != x"
    # error paths of the C ABI (round 3: every rank draws its verdict from exchanged words BEFORE any point-to-point transfer is posted, so
    # an error is an error code on every rank, never a hang): root without room, shard without room, a header that lies
    import ctypes
    L = hsrle.lib(); comm = hd.c_comm(); stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.hsrle_gather_container_rccl.restype = ctypes.c_int
    L.hsrle_gather_container_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    L.hsrle_scatter_container_rccl.restype = ctypes.c_int
    L.hsrle_scatter_container_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    small = torch.empty(1024, dtype=torch.uint8, device=dev); total = ctypes.c_uint64(0)
    rc = L.hsrle_gather_container_rccl(comm, 0, ctypes.c_void_p(container.data_ptr()), container.numel(), size, ctypes.c_void_p(small.data_ptr()), small.numel(), ctypes.byref(total), stream)
    assert rc == 2 and total.value == full.numel(), ("gather into a root without room", rc)            # HSRLE_ERR_CAPACITY, and the size it needs
    rc = L.hsrle_gather_container_rccl(comm, 0, ctypes.c_void_p(container.data_ptr()), container.numel(), size, None, 0, ctypes.byref(total), stream)
    assert rc == 2, ("gather without an output buffer", rc)
    rc = L.hsrle_scatter_container_rccl(comm, 0, ctypes.c_void_p(full.data_ptr()), full.numel(), ctypes.c_void_p(small.data_ptr()), small.numel(), ctypes.byref(total), stream)
    assert rc == 2, ("scatter into a shard without room", rc)
    bad = full.clone(); bad[40:48] = 0xFF                                  # totalSize: the header no longer describes the container
    big = torch.empty(full.numel() + 4096, dtype=torch.uint8, device=dev)
    rc = L.hsrle_scatter_container_rccl(comm, 0, ctypes.c_void_p(bad.data_ptr()), bad.numel(), ctypes.c_void_p(big.data_ptr()), big.numel(), ctypes.byref(total), stream)
    assert rc == 3, ("scatter of a container whose header lies", rc)                                   # HSRLE_ERR_FORMAT
    rc = L.hsrle_gather_container_rccl(comm, 0, ctypes.c_void_p(bad.data_ptr()), bad.numel(), size, ctypes.c_void_p(big.data_ptr()), big.numel(), ctypes.byref(total), stream)
    assert rc == 3, ("gather of a local container whose header lies", rc)
    res["c_error_paths_" + name] = "ok"
    del src, container, full, back, out, dst, fullc, backc, bad, big, small
hd.destroy_c_comms()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""


@pytest.mark.gpu
def test_rccl_world_of_one_gather_scatter_and_pieces():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res["pieces_noise"] == 2 and res["payload_noise"] > (1 << 30)   # the > 1 GiB transfer really was cut
    out = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_world1.json"), "w") as f:
            json.dump(res, f, indent=1)


@pytest.mark.gpu
@pytest.mark.parametrize("c_abi", [True, False], ids=["c_abi_gather", "torch_gather"])
def test_bench_distributed_leg_runs_as_a_world_of_one(c_abi):
    """bench.py's own sharded leg (HSRLE_FORCE_DIST=1: RCCL process group, seeds 100 + rank, the gather of the compressed segments through the
    library's C ABI or through torch.distributed) on the hardware that is there -- so the code an 8-GPU driver run executes has run before.
    The line must carry the gather figures (ms, bytes, GB/s, value with the gather), the CPU baseline and a bit-exact verdict."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HSRLE_FORCE_DIST="1", HSRLE_DIST_C="1" if c_abi else "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--size-gib", "1", "--no-extras", "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["bit_exact"] is True and j["scaling"] == "weak"
    assert j["gather_ms"] > 0 and j["gather_bytes"] == 0 and "gather_GBps" in j and j["value_with_gather"] > 0   # a world of one: nothing crosses a link
    assert j["value_with_gather"] < j["value"] and len(j["per_rank_ms_per_step"]) == 1
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["matches_gpu_input"] and j["encode"]["cpu_baseline"]["streams_match_gpu"]
    assert j["roofline"]["frac"] > 0 and j["config"]["sharding"] == "blocks x1"
    # the line says what the communicator spans, and what every rank found and ran
    assert j["rccl_ranks"] == 1 and ("ncclCommCount" in j["rccl_ranks_source"]) == bool(c_abi)
    assert j["per_rank_bit_exact"] == [True] and j["per_rank_library_build_id"] == [j["library_build_id"]]
