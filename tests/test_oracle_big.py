"""The oracle's restatement against the BIG-CONFIG manifests (tests/golden/big/: per-256-block roll-ups of the stream hashes and the
monolithic stream's sha256, minted from the COMPILED REFERENCE over the deterministic synthetic buffers of BASELINE configs 2 / 3 and
the 8 GiB headline buffer).  CPU only; bounded prefixes so that the whole file runs in well under a minute."""
import ctypes
import hashlib

import numpy as np
import pytest

from hsrle_testlib import CODEC_BY_KEY, big_manifest, big_case, rollups


def _oracle_rollups(oracle, codec, data, block):
    L = oracle.lib
    L.hso_hash_blocks.restype = ctypes.c_uint64
    L.hso_hash_blocks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    nb = (data.size + block - 1) // block
    hashes = np.empty(nb, dtype=np.uint64)
    sizes = np.empty(nb, dtype=np.uint32)
    assert L.hso_hash_blocks(codec.family, codec.S, codec.aligned, data.ctypes.data, data.size, block, hashes.ctypes.data, sizes.ctypes.data) == nb
    return hashes, sizes


@pytest.mark.skipif(big_manifest() is None, reason="big manifests not minted")
def test_rollup_helper_matches_the_c_definition(oracle):
    rng = np.random.default_rng(1)
    h = rng.integers(0, 1 << 63, size=1000, dtype=np.uint64)
    out = np.empty(4, dtype=np.uint64)
    oracle.lib.hso_rollups.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    oracle.lib.hso_rollups(h.ctypes.data, 1000, 256, out.ctypes.data)
    assert (rollups(h) == out).all()


@pytest.mark.skipif(big_manifest() is None, reason="big manifests not minted")
@pytest.mark.parametrize("name,prefix", [("config2_1GiB", 64 << 20), ("headline_8GiB", 32 << 20), ("config4_shard5", 16 << 20), ("config3_video", None)] +
                         [(k, 8 << 20) for k in sorted((big_manifest() or {"cases": {}})["cases"]) if k.startswith("config5_")])   # every width x {Packed, 3LUT} (+ Single), seed 5
def test_oracle_block_streams_hash_like_the_reference(oracle, name, prefix):
    e = big_manifest()["cases"][name]
    codec = CODEC_BY_KEY[e["codec"]]
    _, _, want = big_case(e["codec"], e["kind"], e["seed"], e["size"], e["block"])
    size = e["size"] if prefix is None else prefix
    data = oracle.synth(e["kind"], codec.S, e["seed"], size)
    hashes, sizes = _oracle_rollups(oracle, codec, data, e["block"])
    got = rollups(hashes)
    full = size // e["block"] // 256                                      # complete groups of the prefix
    assert full > 0 and (got[:full] == want[:full]).all()
    if prefix is None:
        assert (got == want).all() and int(sizes.sum()) == e["payload_size"]
        oracle.lib.hso_hash64.restype = ctypes.c_uint64
        oracle.lib.hso_hash64.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        assert "%016x" % oracle.lib.hso_hash64(sizes.ctypes.data, 4 * sizes.size) == e["sizes_hash"]


@pytest.mark.skipif(big_manifest() is None, reason="big manifests not minted")
def test_oracle_monolithic_stream_of_the_config3_frame(oracle):
    e = big_manifest()["cases"]["config3_video"]
    codec = CODEC_BY_KEY[e["codec"]]
    data = oracle.synth(e["kind"], codec.S, e["seed"], e["size"])
    assert hashlib.sha256(data.tobytes()).hexdigest() == e["input_sha256"]
    stream = oracle.compress(codec, data.tobytes())
    assert len(stream) == e["mono"]["size"] and hashlib.sha256(stream).hexdigest() == e["mono"]["sha256"]
