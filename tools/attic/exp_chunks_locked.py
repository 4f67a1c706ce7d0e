"""Experiment (continued): two chains of 2048 C2 instances inside one hipGraph with the phase between them HELD:
a one-wave timer kernel of `delta` microseconds starts together with every step launch of chain A, and chain B's step k waits for
it (and vice versa), so B's launches start `delta` after A's and A's next launch `delta` after B's -- one chain's load burst and
epilogue always meet the other chain's sub-step loop.  Free-running chains (tools/exp_chunks_offset.py) drift in and out of that
phase (15.2 us per batch step in one run, 17.7 in the next)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

UNIT_US = 0.1465          # one torch.cuda._sleep unit on MI355X (calibrated by exp_chunks_offset.py: 1e6 units = 143-151 ms)


def run(delta_us, steps=40, warm=20, B=4096, cls=bench.Parabolic1D):
    dev = torch.device("cuda", 0)
    wls = [cls(dev, 1 + i, B=B // 2) for i in range(2)]
    for w in wls:
        w.prepare(warm + 2 * steps + 8)
        for _ in range(warm):
            w.step()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    sa, sb, ta, tb = (torch.cuda.Stream() for _ in range(4))
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    units = max(1, int(delta_us / UNIT_US))
    with torch.cuda.stream(cap):
        for w, s in zip(wls, (sa, sb)):
            with torch.cuda.stream(s):
                w.step(); w.i -= 1
        for s in (ta, tb):
            with torch.cuda.stream(s):
                torch.cuda._sleep(10)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            for s in (sa, sb, ta, tb):
                s.wait_stream(cap)
            A, Bw = wls
            # a metronome: stream ta runs a chain of timer kernels of `delta` microseconds each; A's step k waits for tick 2k, B's
            # for tick 2k + 1 (events recorded on the timer stream) -- no waits from the timer stream on the chains
            for k in range(steps):
                ev_a = torch.cuda.Event()
                ev_a.record(ta)
                sa.wait_event(ev_a)
                with torch.cuda.stream(sa):
                    A.step()
                with torch.cuda.stream(ta):
                    torch.cuda._sleep(units)
                ev_b = torch.cuda.Event()
                ev_b.record(ta)
                sb.wait_event(ev_b)
                with torch.cuda.stream(sb):
                    Bw.step()
                with torch.cuda.stream(ta):
                    torch.cuda._sleep(units)
            for s in (sa, sb, ta, tb):
                cap.wait_stream(s)
    torch.cuda.current_stream().wait_stream(cap)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[3] / steps, min(ts) / steps, max(ts) / steps


if __name__ == "__main__":
    for d in (6.5, 7, 7.5, 8, 8.5, 9, 10, 7.5, 8):
        med, lo, hi = run(d)
        print(f"Parabolic1D B=2x2048 locked phase delta={d} us: {med * 1e6:.2f} us per env-step of the batch (min {lo * 1e6:.2f}, max {hi * 1e6:.2f})", flush=True)
