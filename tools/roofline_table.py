#!/usr/bin/env python3
"""The ONE roofline table of DESIGN.md section 6, from tracked files only:
   python tools/roofline_table.py profiles/r06f_summary.json profiles/r06i_summary.json profiles/r06j_bench_also.json
rocprof columns (kernel, average duration, HBM traffic) come from the profile round's summary, time and fractions from the driver-style
bench run's full result (bench_also.json: HIP-event step time, SQ_INSTS_VALU / FETCH / WRITE per step / time / peak).
Compulsory bytes per env-step (the "x" column's denominator): 1D step = row in + beta in + row/obs out (3 148 B at nx = 256); rollouts = the
observation slot + command + reward + flags of every env-step (the state is carried in registers); NS = 6 fields (state 2 + p in, observation 2
+ p out; the shared reference frame comes from L2); traffic = r, y in and out + observation; tumour = row in and out + scalars."""
import json
import sys

summ = {}
for path in sys.argv[1:-1]:           # several profile rounds: a later one replaces the workloads it re-profiled
    summ.update({k: v for k, v in json.load(open(path))["workloads"].items() if v.get("avg_ns")})
full = json.load(open(sys.argv[-1]))
ROWS = [("parabolic_c2", "**C2 headline** Parabolic nx=256 B=4096 S=100 f32", 309444, 3148, 4096),
        ("transport_c3", "C3 Transport nx=512 B=16384 S=100 f32", 100 * 12 * 512 + 4 * 512 + 16, 4 * 512 * 2 + 4 * 512 + 64, 16384),
        ("ns2d_c4", "C4 NS 128² K=50 B=512 f32", 4 * 166 * 128 * 128, 4 * 6 * 128 * 128, 512),
        ("ns2d_c4_b4096", "metric: NS 128² K=50 B=4096 f32", 4 * 166 * 128 * 128, 4 * 6 * 128 * 128, 4096),
        ("ns2d_c4_f64", "C4 at f64 B=512", 8 * 166 * 128 * 128, 8 * 6 * 128 * 128, 512),
        ("ns2d_c4_f64_b4096", "metric at f64 B=4096", 8 * 166 * 128 * 128, 8 * 6 * 128 * 128, 4096),
        ("ns2d_c5", "C5 shard NS 256² K=50 B=512 f32", 4 * 166 * 256 * 256, 4 * 6 * 256 * 256, 512),
        ("ns2d_c5_f64", "C5 shard at f64", 8 * 166 * 256 * 256, 8 * 6 * 256 * 256, 512),
        ("ns2d_example", "NS 21×21 K=2000 B=8192 f64 (shipped example)", 8 * 6016 * 21 * 21, 8 * 6 * 21 * 21, 8192),
        ("parabolic_c2_s1", "C2 shape, S=1, per-step launch", 12 * 257 + 4 * 257 + 16, 3148, 4096),
        ("parabolic_c2_s1_open_loop_rollout", "C2 shape, S=1, 100 env-steps per launch", 100 * (12 * 257 + 4 * 257 + 16), 100 * (4 * 257 + 16), 4096),
        ("parabolic_c2_open_loop_rollout", "C2, 25 env-steps per launch (commands ahead)", 25 * 309444, 25 * (4 * 257 + 16), 4096),
        ("parabolic_c2_rollout", "C2, 25 env-steps per launch, 64-unit policy inside", 25 * 309444, 25 * (4 * 257 + 16), 4096),
        ("parabolic_c2_rollout_256", "C2, 25 env-steps per launch, 256-unit policy inside", 25 * 309444, 25 * (4 * 257 + 16), 4096),
        ("traffic_arz", "Traffic M=51 f64 B=16384 S=2", 2 * 32 * 51 + 16 * 51 + 48, 2 * 16 * 51 + 16 * 51 + 64, 16384),
        ("traffic_arz_rollout", "Traffic, 25 env-steps per launch", 25 * (2 * 32 * 51 + 16 * 51 + 48), 25 * (16 * 51 + 18), 16384),
        ("brain_tumor", "Tumour nx=201 f64 B=65536", 16 * 201 + 96, 16 * 201 + 96, 65536)]


def f(x, spec=".2f"):
    return "—" if x is None else format(x, spec)


print("| workload | dominant kernel | rocprof avg µs | step µs (driver-style) | env-steps/s | bound | VALU issue (f64-w.) | HBM | traffic per step (× compulsory) | algorithmic B per env-step |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, label, alg, comp, B in ROWS:
    e = full if key == "parabolic_c2" else (full.get("also") or {}).get(key)
    s = summ.get(key, {})
    if not e or "error" in e:
        continue
    rf = e["roofline"]
    if key == "parabolic_c2":
        valu, hbm, w = (rf.get("valu_issue") or {}).get("frac"), (rf.get("hbm") or {}).get("frac"), (rf.get("valu_issue") or {}).get("frac_f64_weighted")
    else:
        valu, hbm, w = rf.get("valu_issue_frac"), rf.get("hbm_frac"), rf.get("valu_issue_frac_f64_weighted")
    tr = rf.get("traffic")
    kern = (s.get("kernel") or "").split("<")[0]
    units = {"parabolic_c2_s1_open_loop_rollout": 100}.get(key, 25 if "rollout" in key else 1)
    print(f"| {label} | `{kern}` | {f(s.get('avg_ns', 0) / 1e3 if s.get('avg_ns') else None, '.1f')} | {f(rf['step_ms'] * 1e3, '.1f')} | {e['value']:.3g} | "
          f"{rf.get('bound')} | {f(valu)}{'' if not w or abs(w - (valu or 0)) < 0.005 else ' (' + f(w) + ')'} | {f(hbm)} | "
          f"{f(tr / 1e6 if tr else None, '.1f')} MB ({f(tr / (comp * B) if tr else None, '.2f')}×) | {alg // units if units > 1 else alg:,}{' × ' + str(units) if units > 1 else ''} |")
