#!/usr/bin/env python3
"""One-launch rollouts (pdegym_*_rollout) against T step launches in a hipGraph, on the BASELINE shapes.

    python tools/bench_rollout_kernel.py [T] [workload ...]     workloads: parabolic_c2 transport_c3 burgers_c3
Prints us per env-step of the batch for both forms (median of 5 timed regions each) and checks that they agree bitwise.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    names = sys.argv[2:] or ["parabolic_c2", "transport_c3", "burgers_c3"]
    dev = torch.device("cuda", 0)
    cls = {"parabolic_c2": bench.Parabolic1D, "transport_c3": bench.Transport1D, "burgers_c3": bench.Burgers1D}
    for name in names:
        res = {}
        for form in ("steps_in_graph", "one_launch"):
            w = cls[name](dev, 1)
            w.prepare(T)
            e, B, n = w.env, w.B, w.env.n
            obs = torch.zeros(T + 1, B, n, device=dev)
            rew = torch.zeros(T, B, device=dev)
            te = torch.zeros(T, B, dtype=torch.uint8, device=dev)
            tr = torch.zeros(T, B, dtype=torch.uint8, device=dev)
            first = e.t["obs"].clone()
            keys = [k for k in ("time_index", "bsum", "ring", "reset_count", "beta") if torch.is_tensor(e.t.get(k))]
            snap = {k: e.t[k].clone() for k in keys}

            def body():
                if form == "one_launch":
                    e.rollout(obs, w.actions, rew, te, tr)
                else:
                    e.t["obs"] = obs[0]
                    e.t["u"] = obs[0]
                    for t in range(T):
                        e.step(w.actions[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])

            def rewind():
                obs[0].copy_(first)
                for k, v in snap.items():
                    e.t[k].copy_(v)

            rewind()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
                rewind()
                with torch.cuda.graph(g, stream=s):
                    body()
            torch.cuda.current_stream().wait_stream(s)
            times = []
            for _ in range(6):
                rewind()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            times = sorted(times[1:])
            res[form] = (times[len(times) // 2] / T * 1e6, obs.clone(), rew.clone(), te.clone(), tr.clone())
        same = all(torch.equal(a, b) for a, b in zip(res["steps_in_graph"][1:], res["one_launch"][1:]))
        a, b = res["steps_in_graph"][0], res["one_launch"][0]
        print(f"{name}: T={T}  step launches {a:7.2f} us/env-step ({w.B / a:6.1f} M env-steps/s)   one launch {b:7.2f} us/env-step "
              f"({w.B / b:6.1f} M env-steps/s)   x{a / b:.2f}   bitwise equal: {same}", flush=True)


if __name__ == "__main__":
    main()
