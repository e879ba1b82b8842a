import ctypes, os, sys, math, torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
M = 128 * 2048
g = torch.Generator(device=dev); g.manual_seed(1)
X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
W = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
bias = torch.randn((512,), generator=g, device=dev) * 0.1
C = torch.empty_like(X)
bm = torch.empty((M * 512 // 8,), dtype=torch.uint8, device=dev)
cs = torch.zeros((512,), device=dev)
csws = torch.empty((256 * 512,), device=dev)
WITH_CS = os.environ.get('WITH_CS') == '1'
P, I = ctypes.c_void_p, ctypes.c_int
names = sys.argv[1:] or [f"libabl_{k}.so" for k in range(6)]
for nm in names:
    L = ctypes.CDLL(os.path.join(here, nm))
    f = L.murcl_panel_gemm
    f.argtypes = [P, P, P, I, I, I, I, P, P, P, P, P, I, P, I, P, I, P]
    st = torch.cuda.current_stream().cuda_stream
    def run(epi=0, bm_in=None):
        rc = f(X.data_ptr(), W.data_ptr(), C.data_ptr(), M, 512, 512, epi, bias.data_ptr(), bm.data_ptr() if epi == 0 else None,
               bm.data_ptr() if epi == 1 else None, None, None, 0, cs.data_ptr() if (WITH_CS and epi == 1) else None, 0, csws.data_ptr(), 0, st)
        assert rc == 0, rc
    for epi in (0, 1):
        for _ in range(3): run(epi)
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(epi); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"{nm} epi={epi}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us", flush=True)
