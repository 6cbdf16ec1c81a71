# which tables make rle8_low_entropy_decompress_with_info fail?
import sys, ctypes, random
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/hypersonic-rle-kit_amd/python")
import hsrle
from hsrle_testlib import mixed_runs, Reference
import test_gpu_low_entropy_helpers as T
lib, ref = T._bind(hsrle.lib()), T._bind(Reference().lib)
rng = random.Random(3301)
data = mixed_runs(rng, 200000, alphabet=3)
a = T.CompressInfo(); lib.rle8_low_entropy_get_compress_info(data, len(data), ctypes.byref(a))
print("flags", [i for i in range(256) if a.rle[i]], "count", a.symbolCount, "order", list(a.symbolsByProb)[:8])
for extra in ([], [7], [200], [0x41, 0x42]):
  for swap in (None, (0, 1), (3, 40)):
    info = T.CompressInfo.from_buffer_copy(bytes(a))
    for s in extra: info.rle[s] = 1
    if swap:
        p = list(info.symbolsByProb); p[swap[0]], p[swap[1]] = p[swap[1]], p[swap[0]]
        for k in range(256): info.symbolsByProb[k] = p[k]
    body = T._body(ref, "rle8_low_entropy_compress_with_info", data, info)
    mine = T._body(lib, "rle8_low_entropy_compress_with_info", data, info)
    head = ctypes.create_string_buffer(600); hs_ = ref.rle8_low_entropy_write_compress_info(ctypes.byref(info), head, 600)
    d = T.DecompressInfo(); ref.rle8_low_entropy_read_decompress_info(head.raw, hs_, ctypes.byref(d))
    got = T._decode(lib, "rle8_low_entropy_decompress_with_info", body, d, len(data))
    refgot = T._decode(ref, "rle8_low_entropy_decompress_with_info", body, d, len(data))
    print(extra, swap, "bodies equal", mine == body, "mine decodes", got == data, "reference decodes", refgot == data, len(body))
