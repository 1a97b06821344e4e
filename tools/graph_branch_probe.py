"""Do independent branches of a captured hipGraph run concurrently on this ROCm build?  Two chains of 30 tiny kernels
each: captured on one stream vs forked onto a side stream.  Prints the replay time of both graphs."""
import time
import torch
dev = torch.device('cuda:0')
a = torch.randn(4096, device=dev)
b = torch.randn(4096, device=dev)


def chain(x, n=30):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x


def capture(fork):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(a); chain(b)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if fork:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yb = chain(b)
            ya = chain(a)
            cur.wait_stream(side)
        else:
            ya = chain(a); yb = chain(b)
        out = ya + yb
    return g, out


for fork in (False, True):
    g, out = capture(fork)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print('fork' if fork else 'single', '%.1f us per replay (61 kernels)' % ((time.perf_counter() - t0) / 200 * 1e6))
