import torch, time
x = torch.zeros(4096, device="cuda")
def run(n, size):
    y = torch.zeros(size, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): y.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): y.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
for size in (1024, 65536, 1 << 20):
    print("graph: dependent elementwise kernels, size", size, "us/kernel", round(run(2000, size), 3))
# eager
y = torch.zeros(1024, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5000): y.add_(1.0)
torch.cuda.synchronize(); print("eager us/kernel", (time.perf_counter() - t0) / 5000 * 1e6)
