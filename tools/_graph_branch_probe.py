"""Does hipGraph run independent branches concurrently?  3 chains x 20 small kernels: one stream vs three forked streams."""
import torch, time
dev = "cuda"
def chain(x, n):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x
for numel in (1 << 14, 1 << 18, 1 << 21):
    xs = [torch.randn(numel, device=dev) for _ in range(3)]
    res = {}
    for mode in ("serial", "forked"):
        g = torch.cuda.CUDAGraph()
        side = [torch.cuda.Stream() for _ in range(2)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2): [chain(x, 20) for x in xs]
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                cur = torch.cuda.current_stream()
                if mode == "serial":
                    outs = [chain(x, 20) for x in xs]
                else:
                    outs = [None] * 3
                    for k in range(2):
                        side[k].wait_stream(cur)
                        with torch.cuda.stream(side[k]):
                            outs[k + 1] = chain(xs[k + 1], 20)
                    outs[0] = chain(xs[0], 20)
                    for k in range(2): cur.wait_stream(side[k])
                tot = outs[0] + outs[1] + outs[2]
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 50 * 1e6
    print(f"numel {numel}: serial {res['serial']:.1f} us, forked {res['forked']:.1f} us  (120 kernel launches)")
