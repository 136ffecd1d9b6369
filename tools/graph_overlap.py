"""Do hipGraph replays on two streams overlap?  Each graph = 20 launches of a conv that fills ~1/4 of the chip."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
g = (20, 20, 8)
def mk():
    x = torch.randn(3200, 512, device="cuda"); wt = torch.randn(27, 128, 512, device="cuda") * 0.01
    wh, wl = ops.split_bf16(wt)
    return x, wh, wl
def body(a):
    for _ in range(20):
        ops.conv3d_cl_bf16x3(a[0], a[1], a[2], g, 3, 1, False)
A, B = mk(), mk()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
graphs = []
for a, s in ((A, s1), (B, s2)):
    with torch.cuda.stream(s):
        body(a); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            body(a)
        graphs.append(gr)
torch.cuda.synchronize()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def eager_one():
    with torch.cuda.stream(s1): body(A)
def eager_two():
    with torch.cuda.stream(s1): body(A)
    with torch.cuda.stream(s2): body(B)
def graph_one():
    with torch.cuda.stream(s1): graphs[0].replay()
def graph_two():
    with torch.cuda.stream(s1): graphs[0].replay()
    with torch.cuda.stream(s2): graphs[1].replay()
print(f"eager: one stream {timeit(eager_one):.3f} ms, two streams (2x work) {timeit(eager_two):.3f} ms")
print(f"graph: one stream {timeit(graph_one):.3f} ms, two streams (2x work) {timeit(graph_two):.3f} ms")
