"""Split decode of block containers (hsrle_decompress_split_dev_async): one lane per block walks the block's packets and leaves the
decoder state every SB output bytes, then one lane per SUB-block decodes -- or (sub-block size 1 = HSRLE_SPLIT_PACKET_LIST, the library's
choice for blocks of up to 16 KiB) leaves one entry per packet, from which a second kernel builds the output 16 bytes per lane.  Bar: the same bytes as the plain block decode (= the input),
for every codec and every legal sub-block size; malformed blocks are reported, nothing outside the output is written."""
import random
import struct

import pytest

from hsrle_testlib import CODECS, CODEC_BY_KEY, SYNTH_VIDEO, mixed_runs, single_symbol_mix, fuzz_sections, FUZZ_LENGTHS

pytestmark = pytest.mark.gpu

PACKET_LIST = 1          # include/hsrle.h: HSRLE_SPLIT_PACKET_LIST


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()
    return hsrle


def _data(seed, size):
    rng = random.Random(seed)
    parts, n = [], 0
    while n < size:
        k = rng.randrange(4)
        d = (fuzz_sections(rng) if k == 0 else mixed_runs(rng, rng.choice([100, 1000, 3000])) if k == 1 else single_symbol_mix(rng, rng.choice([64, 333, 3000]))
             if k == 2 else bytes(rng.randrange(256) for _ in range(rng.choice([3, 130, 700, 5000]))))
        parts.append(d)
        n += len(d)
    return b"".join(parts)[:size]


def _split(hs, container, info, n, sub, first=0, count=None, guard=0):
    import torch

    count = info.blockCount - first if count is None else count
    ws = torch.full((max(hs.split_workspace_size(info, count, sub), 16),), 0xC3, dtype=torch.uint8, device="cuda")   # (garbage: nothing may rely on a zeroed workspace)
    out = torch.full((n + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    hs.decompress_split_async(container, info, out[:n], ws, status, sub_block=sub, first_block=first, block_count=count)
    torch.cuda.synchronize()
    return out, int(status.item())


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: c.key)
def test_split_decode_equals_the_input(hs, codec):
    import torch

    data = _data(99 + CODECS.index(codec), 150000 + 77)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    for block, subs in ((1024, (128, 256, 512, PACKET_LIST)), (4096, (512, 0, PACKET_LIST)), (1536, (128, 768, PACKET_LIST)), (16384, (PACKET_LIST, 0)), (256, (PACKET_LIST,))):
        container, info = hs.compress(codec.key, src, block_size=block)
        for sub in subs:
            out, status = _split(hs, container, info, len(data), sub, guard=512)
            assert status == 0 and out[: len(data)].cpu().numpy().tobytes() == data, f"{codec.key} block {block} sub {sub}"
            assert bool((out[len(data):] == 0xA5).all())


def test_split_decode_of_a_block_range(hs):
    import torch

    data = _data(5, 300000)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    container, info = hs.compress("rle8_packed_multi", src, block_size=2048)
    first, count = 17, 50
    for sub in (256, PACKET_LIST):
        out, status = _split(hs, container, info, len(data), sub, first=first, count=count)
        host = out.cpu().numpy().tobytes()
        assert status == 0 and host[first * 2048 : (first + count) * 2048] == data[first * 2048 : (first + count) * 2048]
        assert set(host[: first * 2048]) == {0xA5} and set(host[(first + count) * 2048 :]) == {0xA5}
    # the last blocks, the last one partial
    first = info.blockCount - 3
    out, status = _split(hs, container, info, len(data), PACKET_LIST, first=first, count=3)
    host = out.cpu().numpy().tobytes()
    assert status == 0 and host[first * 2048 : len(data)] == data[first * 2048 :] and set(host[: first * 2048]) == {0xA5}


def test_split_decode_into_an_output_that_is_not_16_byte_aligned(hs):
    """The expand kernel stores 16 bytes per lane; nothing says the caller's output is aligned."""
    import torch

    data = _data(6, 200000)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    for key in ("rle8_packed_multi", "rle24_3symlut_byte", "rle64_sym_short"):
        container, info = hs.compress(key, src, block_size=4096)
        ws = torch.full((max(hs.split_workspace_size(info, None, PACKET_LIST), 16),), 0xC3, dtype=torch.uint8, device="cuda")
        for shift in (1, 7, 8):
            out = torch.full((len(data) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            status = torch.zeros(1, dtype=torch.int32, device="cuda")
            hs.decompress_split_async(container, info, out[shift : shift + len(data)], ws, status, sub_block=PACKET_LIST)
            torch.cuda.synchronize()
            host = out.cpu().numpy().tobytes()
            assert int(status.item()) == 0 and host[shift : shift + len(data)] == data and set(host[:shift]) == {0xA5} and set(host[shift + len(data):]) == {0xA5}, f"{key} shift {shift}"


def test_packet_list_of_a_block_with_more_packets_than_its_list(hs):
    """A block whose packets produce fewer than 8 output bytes on average does not fit its list (blockSize / 8 + 2 entries): the walking lane
    closes the list where it stands and writes the rest of the block itself.  Runs of three and four bytes back to back: ~1 200 packets per 4 KiB."""
    import torch

    rng = random.Random(12)
    dense = b"".join(bytes([rng.randrange(256)]) * rng.choice((3, 4, 5)) for _ in range(30000))
    wide = b"".join(bytes(rng.randrange(256) for _ in range(8)) * rng.choice((2, 3)) for _ in range(9000))
    for key, data in (("rle8_packed_multi", dense), ("rle8_multi_short", dense), ("rle8_3symlut", dense), ("rle8_packed_single", b"".join(b"\x07" * rng.choice((3, 4)) + bytes([rng.randrange(1, 7)]) for _ in range(30000))),
                      ("rle64_byte_packed", wide), ("rle64_7symlut_byte_short", wide), ("rle24_sym_packed", b"".join(bytes(rng.randrange(256) for _ in range(3)) * rng.choice((3, 4)) for _ in range(20000)))):
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        for block in (4096, 1024, 16384):
            container, info = hs.compress(key, src, block_size=block)
            out, status = _split(hs, container, info, len(data), PACKET_LIST, guard=512)
            assert status == 0 and out[: len(data)].cpu().numpy().tobytes() == data, f"{key} block {block}"
            assert bool((out[len(data):] == 0xA5).all())


def test_split_decode_reports_malformed_blocks(hs):
    import torch

    rng = random.Random(8)
    data = mixed_runs(rng, 200000)
    for key in ("rle8_packed_multi", "rle8_3symlut", "rle32_byte", "rle64_7symlut_byte_short"):
        container, info = hs.compress(key, torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda(), block_size=1024)
        host = bytearray(container.cpu().numpy().tobytes())
        p0 = info.payload_start
        table = struct.unpack_from(f"<{info.blockCount + 1}Q", host, 64)
        for i in range(0, info.blockCount, 3):
            a, b = p0 + table[i], p0 + table[i + 1]
            for j in range(a + 10, b):
                host[j] = rng.randrange(256)
        bad = torch.frombuffer(host, dtype=torch.uint8).cuda()
        host2 = bytearray(container.cpu().numpy().tobytes())
        host2[64 + 8 * 3 : 64 + 8 * 4] = struct.pack("<Q", 1 << 40)          # table entry outside the payload
        bad2 = torch.frombuffer(host2, dtype=torch.uint8).cuda()
        for sub in (256, PACKET_LIST):
            out, status = _split(hs, bad, info, len(data), sub, guard=4096)
            assert status != 0 and bool((out[len(data):] == 0xA5).all())
            # the blocks that were left alone still decode
            good = [i for i in range(info.blockCount) if i % 3 != 0]
            host_out = out.cpu().numpy().tobytes()
            assert all(host_out[i * 1024 : (i + 1) * 1024] == data[i * 1024 : (i + 1) * 1024] for i in good[:40]), f"{key} sub {sub}"
            out, status = _split(hs, bad2, info, len(data), sub, guard=4096)
            assert status != 0 and bool((out[len(data):] == 0xA5).all())


def test_config3_frame_split_decode(hs, oracle):
    """BASELINE config 3 at its own size and block size: 88 473 600 bytes video-shaped, rle64_3symlut_byte, 4 KiB blocks."""
    import torch

    size = 88473600
    src = hs.synth(SYNTH_VIDEO, 8, 3, size, device="cuda")
    container, info = hs.compress("rle64_3symlut_byte", src, block_size=4096)
    assert hs.lib().hsrle_split_sub_block_size(__import__("ctypes").byref(info), 0) == PACKET_LIST
    for sub in (0, PACKET_LIST, 1024, 256):
        out, status = _split(hs, container, info, size, sub)
        assert status == 0 and torch.equal(out[:size], src), f"sub {sub}"


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: c.key)
def test_wave_decode_every_codec(hs, codec):
    """One wave per block (hsrle_decompress_wave_dev_async, csrc/hsrle_decode_wave.hip.h): every grammar, block sizes 128 .. 16 KiB, blocks with
    more packets than a descriptor batch holds, a range of blocks, the bytes behind the output untouched."""
    if not hs.experiments_enabled():
        pytest.skip("the wave-per-block decoder is not part of the shipped build (measured slower than the split decode: DESIGN.md 8)")
    import torch

    rng = random.Random(31 + CODECS.index(codec))
    data = b"".join(d for d in (mixed_runs(rng, rng.choice([1, 17, 333, 3000, 9000])) for _ in range(12)) if d) + bytes(rng.randrange(4) for _ in range(40000))
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    for block_size in (128, 4096, 16384):
        container, info = hs.compress(codec.key, src, block_size=block_size)
        out = torch.full((len(data) + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        hs.decompress_wave_async(container, info, out[: len(data)], status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        assert out[: len(data)].cpu().numpy().tobytes() == data, f"{codec.key}: wave decode differs (block size {block_size})"
        assert bool((out[len(data):] == 0xA5).all())
    if info.blockCount > 2:
        out.fill_(0xA5)
        hs.decompress_wave_async(container, info, out[: len(data)], status, first_block=1, block_count=1)
        torch.cuda.synchronize()
        bs = info.blockSize
        assert out[bs : 2 * bs].cpu().numpy().tobytes()[: len(data) - bs] == data[bs : 2 * bs] and bool((out[:bs] == 0xA5).all()) and bool((out[2 * bs : len(data)] == 0xA5).all())
