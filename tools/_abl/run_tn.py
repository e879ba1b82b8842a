import ctypes, os, sys, math, torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
M = 128 * 2048
g = torch.Generator(device=dev); g.manual_seed(1)
X = (torch.randn((M, 512), generator=g, device=dev) * 0.5).bfloat16()
H = (torch.randn((M, 512), generator=g, device=dev) * 0.5).bfloat16()
C = torch.zeros((512, 512), device=dev)
P, I = ctypes.c_void_p, ctypes.c_int
names = sys.argv[1:]
for rep in range(2):
  for nm in names:
    L = ctypes.CDLL(os.path.join(here, nm))
    f = L.murcl_gemm_tn
    f.argtypes = [P, P, P, I, I, I, I, I, I, I, I, P]
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = f(X.data_ptr(), H.data_ptr(), C.data_ptr(), M, 512, 512, 512, 512, 512, 1, 0, st)
        assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{nm}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us", flush=True)
