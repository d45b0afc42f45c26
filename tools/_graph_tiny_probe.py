"""Wall cost of a tiny dependent kernel inside a hipGraph, alone and behind a big streaming kernel."""
import torch, time
dev = "cuda"
big = torch.randn(32 << 20, device=dev)            # 128 MB
small = torch.zeros(64, device=dev)
def run(fn, reps=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s): fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
def tiny_only():
    for _ in range(100): small.add_(1.0)
def big_only():
    for _ in range(20): big.mul_(1.0001)
def big_tiny(k):
    def f():
        for _ in range(20):
            big.mul_(1.0001)
            for _ in range(k): small.add_(1.0)
    return f
a = run(tiny_only); b = run(big_only)
print(f"100 tiny: {a:.1f} us ({a / 100:.2f} us each);  20 big: {b:.1f} us ({b / 20:.1f} each)")
for k in (1, 3, 6):
    c = run(big_tiny(k))
    print(f"20 x (big + {k} tiny): {c:.1f} us -> {(c - b) / (20 * k):.2f} us per tiny")
