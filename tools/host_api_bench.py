# PCIe-inclusive rate of the host-pointer container API (hsrle_compress_host / hsrle_decompress_host): python tools/host_api_bench.py [size_mib]
import sys, time, ctypes
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import numpy as np, torch, hsrle
size=(int(sys.argv[1]) if len(sys.argv)>1 else 1024)<<20
L=hsrle.lib()
src=hsrle.synth(0,1,2,size).cpu().numpy()
cap=hsrle.container_bound(size,4096)
dst=np.empty(cap,dtype=np.uint8); out=np.empty(size,dtype=np.uint8)
csz=ctypes.c_uint64(0); usz=ctypes.c_uint64(0)
codec=L.hsrle_codec_from_name(b'rle8_packed_multi')
def enc(): return L.hsrle_compress_host(codec, src.ctypes.data, size, dst.ctypes.data, cap, 4096, ctypes.byref(csz))
def dec(): return L.hsrle_decompress_host(dst.ctypes.data, csz.value, out.ctypes.data, size, ctypes.byref(usz))
assert enc()==0 and dec()==0
te=min(timeit for timeit in [ (lambda: (time.perf_counter(), enc(), time.perf_counter()))() for _ in range(3)] for timeit in [timeit[2]-timeit[0]])
td=min(t[2]-t[0] for t in [ (lambda: (time.perf_counter(), dec(), time.perf_counter()))() for _ in range(3)])
print('host-pointer container API, %d MiB rle8_packed_multi (pageable host memory, PCIe inclusive): compress %.2f GiB/s, decompress %.2f GiB/s, round trip %s'%(size>>20, size/2**30/te, size/2**30/td, 'ok' if (out==src).all() else 'FAIL'))
