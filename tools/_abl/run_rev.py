# chain of 3 forward layers X -> H1 -> H2 -> H3 (as in the step); PG_REVERSE=0: all forward walks, 2: alternate directions
import ctypes, os, sys, math, torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
M = 128 * 2048
g = torch.Generator(device=dev); g.manual_seed(1)
X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
W = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
bias = torch.randn((512,), generator=g, device=dev) * 0.1
Hs = [torch.empty_like(X) for _ in range(3)]
bms = [torch.empty((M * 512 // 8,), dtype=torch.uint8, device=dev) for _ in range(3)]
P, I = ctypes.c_void_p, ctypes.c_int
L = ctypes.CDLL(os.path.join(here, "libpg_rev.so"))
f = L.murcl_panel_gemm
f.argtypes = [P, P, P, I, I, I, I, P, P, P, P, P, I, P, I, P]
st = torch.cuda.current_stream().cuda_stream
def chain():
    src = X
    for l in range(3):
        rc = f(src.data_ptr(), W.data_ptr(), Hs[l].data_ptr(), M, 512, 512, 0, bias.data_ptr(), bms[l].data_ptr(), None, None, None, 0, None, 0, st)
        assert rc == 0
        src = Hs[l]
for _ in range(3): chain()
torch.cuda.synchronize()
ts = []
for _ in range(15):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); chain(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(f"PG_REVERSE={os.environ.get('PG_REVERSE','0')}: 3-layer chain median {ts[7]:.1f} us  min {ts[0]:.1f} us", flush=True)
