#!/usr/bin/env python
"""Dev experiment: does torch.cuda.graph capture / replay work with the ctypes launches?
'step': the whole bench step (Adam's host-side step counter is frozen in the graph - timing only).  Measured: replay
1.704 ms/step vs eager 1.702 ms/step, i.e. the step is GPU-bound and a graph buys nothing here.
'ops': three launches captured alone - the replay left garbage in the split-K output on this ROCm build (the zero-fill
node?), which is why nothing in the product relies on graph capture."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if len(sys.argv) < 2 or sys.argv[1] == "ops":
    A = torch.randn(128, 512, device=dev); W = torch.randn(256, 512, device=dev); b = torch.randn(256, device=dev)
    def f():
        y = ops.gemm_nt(A, W, epi=ops.EPI_BIAS_RELU, bias=b)
        g = ops.gemm_tn(y, A)
        return y, g + 1.0
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        y, g = f()
    ref_y, ref_g = f()
    torch.cuda.synchronize()
    A.mul_(2.0)
    gph.replay(); torch.cuda.synchronize()
    y2, g2 = f()
    print("replay vs eager on new input: max|dy| %.3e (|y| %.1f)  max|dg| %.3e (|g| %.1f);  vs capture-time values: %.3e" % ((y - y2).abs().max().item(), y2.abs().max().item(), (g - g2).abs().max().item(), g2.abs().max().item(), (y - ref_y).abs().max().item()))
    print("ops stage ok", flush=True)
else:
    import bench
    model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
    views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
    step = bench.make_step(model, fc, opt, crit, views, 1)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("warm", flush=True)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        loss = step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(5): gph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): gph.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step, loss {loss.item():.5f}", flush=True)
    t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    print(f"eager:        {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step", flush=True)
