"""Thread safety of the C ABI (include/hsrle.h "Threads and streams"): several threads compress (without a caller
workspace: the library allocates stream-ordered scratch per call), decompress and call the host-pointer drop-in functions at the same
time, each on its own stream, with buffers of different sizes so that a shared or re-allocated workspace would corrupt somebody."""
import random
import threading

import pytest

from hsrle_testlib import CODEC_BY_KEY, mixed_runs

pytestmark = pytest.mark.gpu


def test_concurrent_calls_from_several_threads():
    import torch

    assert torch.cuda.is_available()
    import hsrle

    hsrle.lib()
    errors = []

    def worker(tid):
        try:
            rng = random.Random(tid)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(12):
                    n = rng.choice([70000, 300000, 1500000, 5000000]) + tid
                    key = ["rle8_packed_multi", "rle8_3symlut", "rle16_sym", "rle64_7symlut_byte"][(tid + it) % 4]
                    data = mixed_runs(rng, min(n, 200000)) * (n // 200000 + 1)
                    data = data[:n]
                    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                    container, info = hsrle.compress(key, src, block_size=rng.choice([1024, 4096]))
                    out = hsrle.decompress(container)
                    assert torch.equal(out, src), f"thread {tid} iteration {it}: device round trip"
                    if it % 3 == 0:                                          # host-pointer drop-in calls (per-device lock)
                        small = data[:30000]
                        codec = CODEC_BY_KEY[key]
                        size, s = hsrle.call_dropin(codec.cname, small, hsrle.compress_bounds(len(small)))
                        size2, d = hsrle.call_dropin(codec.dname, s, len(small))
                        assert size2 == len(small) and d == small, f"thread {tid} iteration {it}: drop-in round trip"
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
