"""Where do the rare multi-millisecond host stalls inside a step come from?  Runs the bench step N times in 20-step regions (sync
between regions, like the timed region of bench.py), records the host time of every step and every Python GC pass (gc.callbacks),
and prints the steps that took > 2 ms on the host with what the collector did meanwhile.  Dev tool."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
N = int(os.environ.get("STALL_STEPS", "4000"))
model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
step = bench.make_step(model, fc, opt, crit, views, 1)
step(); torch.cuda.synchronize()
if os.environ.get("STALL_FREEZE", "1") == "1":
    gc.collect(); gc.freeze()
if os.environ.get("STALL_GC_OFF") == "1":
    gc.disable()
events = []
def cb(phase, info):
    events.append((time.perf_counter(), phase, info.get("generation"), info.get("collected")))
gc.callbacks.append(cb)
for _ in range(30): step()
torch.cuda.synchronize()
host, regions = [], []
for i in range(N):
    if i % 20 == 0:
        torch.cuda.synchronize()
        r0 = time.perf_counter()
    t0 = time.perf_counter()
    step()
    host.append((t0, time.perf_counter() - t0))
    if i % 20 == 19:
        torch.cuda.synchronize()
        regions.append((time.perf_counter() - r0) / 20 * 1e3)
h = sorted(x[1] for x in host)
print(f"steps {N}: host enqueue median {h[len(h)//2]*1e3:.3f} ms, p99 {h[int(0.99*len(h))]*1e3:.3f}, max {h[-1]*1e3:.3f}")
rs = sorted(regions)
print(f"20-step regions {len(rs)}: ms/step median {rs[len(rs)//2]:.4f}, p90 {rs[int(0.9*len(rs))]:.4f}, max {rs[-1]:.4f}; regions > 1.1 x median: {sum(r > 1.1 * rs[len(rs)//2] for r in rs)}")
for i, (t0, dt) in enumerate(host):
    if dt > 2e-3:
        g = [(ph, gen, col, round((t - t0) * 1e3, 2)) for t, ph, gen, col in events if t0 <= t <= t0 + dt]
        print(f"step {i} (in-region {i % 20}): {dt*1e3:.2f} ms on the host; gc inside: {g}")
print("gc passes by generation:", {g: sum(1 for _, ph, gen, _ in events if ph == 'start' and gen == g) for g in (0, 1, 2)})
