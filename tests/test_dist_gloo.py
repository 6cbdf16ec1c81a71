"""The multi-GPU sharding logic (hsrle.dist) on CPU: world_size 2, gloo backend.  Ranks encode their contiguous block
ranges (with the oracle standing in for the device codec -- the sharding code is codec agnostic), the containers are gathered
into one, compared with the single-process container, scattered again and decoded."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "hypersonic-rle-kit_amd", "python"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, codec_key, total, block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsrle import dist as hd

        hd.P2P_PIECE = 1000  # force the multi-piece path of the point-to-point transfers (1 GiB pieces in production)
        from hsrle_testlib import CODECS, CODEC_BY_KEY, Oracle, build_container

        ora = Oracle()
        codec = CODEC_BY_KEY[codec_key]
        cid = CODECS.index(codec)
        data = ora.synth(0, codec.S, 11, total)
        lo, size = hd.shard_bytes(total, block, world, rank)
        shard = data[lo : lo + size]
        local = None
        if size > 0:
            streams = ora.compress_blocks(codec, shard, block)
            local = torch.frombuffer(bytearray(build_container(cid, size, block, streams)), dtype=torch.uint8)
        else:
            local = torch.zeros(0, dtype=torch.uint8)

        full = hd.gather_container(local, total, root=0)
        if rank == 0:
            expect = build_container(cid, total, block, ora.compress_blocks(codec, data, block))
            assert full.numpy().tobytes() == expect, "gathered container differs from the single-process container"

        back = hd.scatter_container(full if rank == 0 else None, root=0)
        if size > 0:
            assert back.numpy().tobytes() == local.numpy().tobytes(), "scatter(gather(x)) != x"
            # decode the shard from the scattered container with the oracle
            h = hd.unpack_header(back.numpy().tobytes())
            raw = back.numpy().tobytes()
            offs = np.frombuffer(raw, dtype=np.uint64, count=h["blockCount"] + 1, offset=64)
            p0 = 64 + 8 * (h["blockCount"] + 1)
            out = b"".join(ora.decompress(codec, raw[p0 + int(offs[i]) : p0 + int(offs[i + 1])]) for i in range(h["blockCount"]))
            assert out == shard.tobytes()
        else:
            assert back is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("codec_key,total,block", [("rle8_packed_multi", 300000, 4096), ("rle64_3symlut_byte", 70001, 1024), ("rle8_packed_multi", 3000, 4096)])
def test_gather_scatter_two_ranks(codec_key, total, block):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, codec_key, total, block, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_shard_ranges_cover_everything():
    from hsrle import dist as hd

    for total, block, world in ((1 << 20, 4096, 8), (1000, 128, 8), (123457, 4096, 3), (1, 128, 2)):
        nb = (total + block - 1) // block
        covered = 0
        for r in range(world):
            f, c = hd.shard_blocks(nb, world, r)
            lo, size = hd.shard_bytes(total, block, world, r)
            assert lo == f * block or size == 0
            covered += size
        assert covered == total
