#!/usr/bin/env python
"""Does the K2 ring config keep HBM saturated when every tile also costs N cycles of (emulated) compute?"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import _lib
L = ctypes.CDLL(_lib.LIB_PATH)
f = L.murcl_debug_stream_probe
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
nbytes = 256 << 20
src = torch.empty(nbytes, dtype=torch.uint8, device="cuda").random_()
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(nw, tk, ns, wpc, reads, sleep):
    for _ in range(2): f(src.data_ptr(), out.data_ptr(), nbytes, nw, tk, ns, wpc, reads, sleep, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f(src.data_ptr(), out.data_ptr(), nbytes, nw, tk, ns, wpc, reads, sleep, st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
for cfg in [(4, 16, 4, 2), (8, 32, 4, 1), (4, 32, 2, 2)]:
    for reads in (0, 16):
        for sleep in (0, 1, 2, 3, 4, 6):
            ms = run(*cfg, reads, sleep)
            print(f"NW={cfg[0]} tile={cfg[1]}KB slots={cfg[2]} wg/cu={cfg[3]} lds_reads={reads:2d} sleep={sleep} (~{sleep*512} cyc/tile): {ms*1e3:7.1f} us {nbytes/ms/1e9:6.2f} TB/s", flush=True)
