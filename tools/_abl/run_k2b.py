import ctypes, os, sys, math, torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
B, N = 128, 2048
g = torch.Generator(device=dev); g.manual_seed(1)
H = (torch.randn((B * N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
Wa = (torch.randn((128, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
ba = torch.randn((128,), generator=g, device=dev) * 0.1
wb = torch.randn((128,), generator=g, device=dev) * 0.05
bb = torch.zeros((1,), device=dev)
scores = torch.empty((B, N), device=dev); A = torch.empty((B, N), device=dev); M = torch.empty((B, 512), device=dev)
ml = torch.empty((B, 2), device=dev); ws = torch.empty((B * 64 * 514,), device=dev)
dM = torch.randn((B, 512), generator=g, device=dev)
dT = torch.empty((B * N + 32, 128), dtype=torch.bfloat16, device=dev)
pws = torch.empty(512 * 257, device=dev)
dba = torch.zeros(128, device=dev); dwb = torch.zeros(128, device=dev); dbb = torch.zeros(1, device=dev)
P, I = ctypes.c_void_p, ctypes.c_int
ref = None
for rep in range(2):
  for nm in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(here, nm))
    f = L.murcl_abmil_pool_fwd; f.argtypes = [P] * 10 + [I] * 6 + [P]
    st = torch.cuda.current_stream().cuda_stream
    f(H.data_ptr(), Wa.data_ptr(), ba.data_ptr(), wb.data_ptr(), bb.data_ptr(), scores.data_ptr(), A.data_ptr(), M.data_ptr(),
      ml.data_ptr(), ws.data_ptr(), B, N, 512, 128, 1, 0, st)
    fb = L.murcl_abmil_pool_bwd; fb.argtypes = [P] * 13 + [I] * 6 + [P]
    def run():
        rc = fb(H.data_ptr(), Wa.data_ptr(), ba.data_ptr(), wb.data_ptr(), scores.data_ptr(), ml.data_ptr(), M.data_ptr(), dM.data_ptr(),
                dT.data_ptr(), dba.data_ptr(), dwb.data_ptr(), dbb.data_ptr(), pws.data_ptr(), B, N, 512, 128, 1, 0, st)
        assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    if ref is None: ref = dT.float().clone()
    err = (dT.float() - ref)[:B * N].abs().max().item() / ref.abs().max().item()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    def runf():
        f(H.data_ptr(), Wa.data_ptr(), ba.data_ptr(), wb.data_ptr(), bb.data_ptr(), scores.data_ptr(), A.data_ptr(), M.data_ptr(),
          ml.data_ptr(), ws.data_ptr(), B, N, 512, 128, 1, 0, st)
    tf = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); runf(); b.record(); torch.cuda.synchronize(); tf.append(a.elapsed_time(b) * 1e3)
    tf.sort()
    print(f"   (fwd+combine in this harness: median {tf[10]:.1f} us)")
    print(f"{nm}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  rel diff vs first {err:.2e}", flush=True)
