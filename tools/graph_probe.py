"""Can one eps evaluation of the pixel DDPM UNet be captured into a HIP graph (torch.cuda.CUDAGraph around the
C-ABI call) and replayed?  Timing of eager vs replay for small batches (diagnostic)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.diffusion import Model
dev = "cuda:0"
m = Model(device=dev)
m.init_random(0)
for B in (1, 2, 8):
    x = torch.randn(B, 3, 256, 256, device=dev)
    ref = m(x, 501.0)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(x, 501.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = m(x, 501.0)
    except Exception as e:
        print("capture failed:", repr(e)[:300])
        break
    g.replay(); torch.cuda.synchronize()
    print("B", B, "replay equals eager:", torch.equal(out, ref))
    def timeit(f, n=20):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("   eager %.2f ms   graph replay %.2f ms" % (timeit(lambda: m(x, 501.0)), timeit(g.replay)))
