#!/usr/bin/env python
"""Sweep the LDS-DMA ring skeleton (murcl_debug_stream_probe) to find the HBM read ceiling of each configuration."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import _lib
import subprocess
PROBES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_abl", "lib", "probes.so")
if not os.path.exists(PROBES):
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_abl", "build_probes.py")])
L = ctypes.CDLL(PROBES)          # lab equipment: its own library, not part of libmurcl_amd.so
f = L.murcl_debug_stream_probe
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
nbytes = 512 << 20
src = torch.empty(nbytes, dtype=torch.uint8, device="cuda").random_()
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(nw, tk, ns, wpc, reads=0, sleep=0):
    rc = f(src.data_ptr(), out.data_ptr(), nbytes, nw, tk, ns, wpc, reads, sleep, st)
    if rc: return None
    for _ in range(2): f(src.data_ptr(), out.data_ptr(), nbytes, nw, tk, ns, wpc, reads, sleep, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f(src.data_ptr(), out.data_ptr(), nbytes, nw, tk, ns, wpc, reads, sleep, st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
for cfg in [(4,32,4,1),(8,32,4,1),(8,32,3,1),(4,16,4,2),(8,16,8,1),(4,16,8,1),(4,8,8,2),(4,32,2,2),(8,16,4,2),(4,16,2,4),(4,16,3,3),(4,8,4,4),(2,16,4,2),(2,8,4,4),(1,8,4,4),(1,4,4,8),(1,4,8,4),(2,8,8,2)]:
    for reads, sleep in [(0, 0), (8, 0)]:
        ms = run(*cfg, reads, sleep)
        if ms is None: print(cfg, "unsupported"); break
        print(f"NW={cfg[0]} tile={cfg[1]:2d}KB slots={cfg[2]} wg/cu={cfg[3]} in-flight/CU={(cfg[2]-1)*cfg[1]*cfg[3]:4d}KB lds_reads={reads}: {ms*1e3:7.1f} us  {nbytes/ms/1e9:7.1f} GB/s", flush=True)
