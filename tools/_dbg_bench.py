"""Dev: how many steps does the GPU need before the step time settles?  (10-step regions right after a cold start)"""
import os, sys, time, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
step = bench.make_step(model, fc, opt, crit, views, 1)
for _ in range(5): step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
out = []
for r in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): step()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 10 * 1e3)
print("10-step regions after 5 warm-up steps:", " ".join("%.3f" % x for x in out))
time.sleep(2.0)
out = []
for r in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): step()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 10 * 1e3)
print("after 2 s idle:", " ".join("%.3f" % x for x in out))
