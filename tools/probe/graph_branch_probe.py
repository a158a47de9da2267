"""Does a replayed HIP graph run its parallel branches concurrently?  Two captured streams, ten 100-us single-workgroup spin kernels each:
~1 ms per replay = concurrent, ~2 ms = one after the other.   python tools/probe/graph_branch_probe.py"""
import time, torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
CYC = 240_000          # ~100 us


def body():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(10):
            torch.cuda._sleep(CYC)
    with torch.cuda.stream(s2):
        for _ in range(10):
            torch.cuda._sleep(CYC)
    cur.wait_stream(s1); cur.wait_stream(s2)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    body(); torch.cuda.synchronize()
    print("eager, two streams: %.2f ms" % timed(body))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        body()
    print("graph replay, two captured branches: %.2f ms" % timed(g.replay))

    def serial():
        for _ in range(20):
            torch.cuda._sleep(CYC)
    print("eager, one stream (20 kernels): %.2f ms" % timed(serial))
