"""Experiment: the C2 batch as TWO independent chains of 2048 instances on separate streams inside one hipGraph,
the second chain delayed by a fraction of a step so that one chain's load burst / epilogue meets the other's sub-step loop.
Prints microseconds per env-step of the whole batch (4096 instances) for several delays; chunks = 1 is the shipped single launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(chunks, delay_us, steps=200, warm=20, B=4096, cls=bench.Parabolic1D):
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
                torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            for s in subs:
                s.wait_stream(cap)
            for k, (w, s) in enumerate(zip(wls, subs)):
                with torch.cuda.stream(s):
                    if k and delay_us > 0:
                        torch.cuda._sleep(max(1, int(k * delay_us / 0.143)))      # calibrated below: one _sleep unit = 0.143 us on MI355X
                    for _ in range(steps):
                        w.step()
            for s in subs:
                cap.wait_stream(s)
    torch.cuda.current_stream().wait_stream(cap)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[3] / steps


if __name__ == "__main__":
    # calibrate the sleep: how long is _sleep(100000)?
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(1000000); torch.cuda.synchronize()
    print(f"_sleep(1e6) = {(time.perf_counter() - t0) * 1e6:.0f} us")
    for c, d in ((1, 0), (2, 0), (2, 3), (2, 6), (2, 10), (2, 14), (4, 0), (4, 5), (1, 0)):
        t = run(c, d)
        print(f"Parabolic1D B=4096 chains={c} delay={d} us per chain: {t * 1e6:.2f} us per env-step of the batch", flush=True)
