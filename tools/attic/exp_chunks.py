"""Experiment: C2 batch split into independent chunk chains on separate streams inside one hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def run(chunks, steps=200, warm=20, B=4096, cls=bench.Parabolic1D):
    dev = torch.device("cuda", 0)
    wls = [cls(dev, 1 + i, B=B // chunks) for i in range(chunks)]
    for w in wls:
        w.prepare(warm + 2 * steps + 8)
        for _ in range(warm):
            w.step()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    subs = [torch.cuda.Stream() for _ in range(chunks)]
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        for w, s in zip(wls, subs):
            with torch.cuda.stream(s):
                w.step(); w.i -= 1
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            for s in subs:
                s.wait_stream(cap)
            for w, s in zip(wls, subs):
                with torch.cuda.stream(s):
                    for _ in range(steps):
                        w.step()
            for s in subs:
                cap.wait_stream(s)
    torch.cuda.current_stream().wait_stream(cap)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / steps

if __name__ == "__main__":
    for cls, B in ((bench.Parabolic1D, 4096), (bench.Transport1D, 16384)):
        for c in (1, 2, 4, 1, 2):
            t = run(c, B=B, cls=cls)
            print(f"{cls.__name__} B={B} chunks={c}: {t*1e6:.2f} us/step  {B/t:.4g} env-steps/s")
