"""GPU experiment: does a replayed hipGraph run independent branches concurrently?  Two kernels that each fill a quarter of the chip
(an elementwise pass over a small tensor, repeated; 64 workgroups), captured (a) one after the other on one stream, (b) on two forked
streams joined at the end.  Prints microseconds per replay."""
import torch
dev = torch.device("cuda:0")
x = [torch.randn(64 * 256 * 4, device=dev) for _ in range(2)]


def work(t, n=40):
    for _ in range(n):
        t.mul_(1.0001)          # 64 workgroups of 256 threads x 4 elements: a quarter of the CUs


def timed(g, reps=50):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def big(t, n=10):
    for _ in range(n):
        torch.mm(t, t)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    work(x[0]); work(x[1]); torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        work(x[0]); work(x[1])
    g2 = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g2):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            work(x[1])
        work(x[0])
        torch.cuda.current_stream().wait_stream(side)
    print("elementwise  serial %.1f us   forked %.1f us" % (timed(g1), timed(g2)))
    m = [torch.randn(512, 512, device=dev).bfloat16() for _ in range(2)]
    big(m[0]); big(m[1]); torch.cuda.synchronize()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        big(m[0]); big(m[1])
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            big(m[1])
        big(m[0])
        torch.cuda.current_stream().wait_stream(side)
    print("small GEMMs  serial %.1f us   forked %.1f us" % (timed(g3), timed(g4)))
