"""How far can the host run ahead of the GPU?  Enqueue a long GPU job then N small launches; report host enqueue time."""
import time, sys, os
if len(sys.argv) > 2 and sys.argv[2] == "early": os.environ["HSA_KERNARG_POOL_SIZE"] = sys.argv[1]
import torch
if len(sys.argv) > 2 and sys.argv[2] == "late": os.environ["HSA_KERNARG_POOL_SIZE"] = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
y = torch.zeros(1024, device="cuda")
z = torch.zeros(64, 3, 5, 7, device="cuda")[:, :, ::2, ::3]     # strided -> big kernarg (offset calculator)
def long_job():
    for _ in range(60): torch.mm(a, a)
long_job(); torch.cuda.synchronize()
t0 = time.perf_counter(); long_job(); torch.cuda.synchronize(); print(f"long job = {1e3*(time.perf_counter()-t0):.0f} ms")
for kind in ("big-arg",):
    for n in (2000, 4000, 8000, 16000):
        torch.cuda.synchronize(); long_job(); t0 = time.perf_counter()
        if kind == "small-arg":
            for _ in range(n): y.add_(1.0)
        else:
            for _ in range(n): z.add_(1.0)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{kind}: n={n:5d} host enqueue {1e3*(t1-t0):7.1f} ms ({1e6*(t1-t0)/n:5.1f} us/launch)  gpu done at {1e3*(t2-t0):7.1f} ms")
