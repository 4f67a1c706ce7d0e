#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (rocprofv3 CSVs written by tools/profile_round.sh) into the small files committed under
profiles/: per-workload kernel stats, per-launch HBM traffic (PMC), and SQ counters of the dominant kernels.

HBM traffic follows MI355X_MICROARCH.md section HBM: bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 per launch
(gfx950's FETCH_SIZE tallies 128-byte fabric reads at 64 bytes, hence the factor 2 on the read side; WRITE_SIZE is
taken as is).  Check: the parabolic kernel's known compulsory traffic (row + beta in, row + obs out) is reproduced.

Run it on the tree that was PROFILED: every workload's entry of profiles/counters_latest.json is stamped with the fingerprint of the
kernel sources of the tree this script runs in (bench.kernel_stamp) -- summarising after an experiment has edited a kernel stamps the old
counters with the new sources' fingerprint (round 5 did that once; re-run after reverting)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_stamp: fingerprint of the sources a workload's kernels are compiled from)
from pdecontrolgym_amd import build as _build  # noqa: E402
DOM = {"parabolic_c2_s1": "step1d_kernel", "parabolic_c2_s1_open_loop_rollout": "rollout1d_kernel", "parabolic_c2_s1_rollout": "rollout1d_policy_kernel",
       "parabolic_c2": "step1d_kernel", "parabolic_c2_policy_loop": "step1d_kernel", "parabolic_c2_rollout": "rollout1d_policy_kernel", "parabolic_c2_policy_loop_256": "mlp_forward_kernel", "parabolic_c2_rollout_256": "rollout1d_policy_kernel", "parabolic_c2_open_loop_rollout": "rollout1d_kernel", "transport_c3": "step1d_kernel", "burgers_c3": "step1d_kernel", "ns2d_c4": "ns_tile_step",
       "ns2d_c4_b4096": "ns_tile_step", "ns2d_c5": "ns256_fused_step", "ns2d_c5_f64": "ns256_pass_f64", "ns2d_c4_f64_b4096": "ns_", "ns2d_c4_f64": "ns_", "ns2d_example": "ns_col_step", "traffic_arz": "traffic_step_kernel", "traffic_arz_rollout": "traffic_rollout_kernel",
       "brain_tumor": "tumor_step_kernel"}
if os.path.exists(os.path.join(root, "tools", "dominant_kernels.json")):
    DOM.update(json.load(open(os.path.join(root, "tools", "dominant_kernels.json"))))
summary = {"tag": tag, "units": "bytes per launch of the dominant kernel", "workloads": {}}


def counter_avg(path, kernel_sub):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


for wl, ksub in DOM.items():
    st = os.path.join(src, f"stats_{wl}", "p_kernel_stats.csv")
    entry = {}
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
        for r in csv.DictReader(open(st)):
            if ksub in r["Name"]:
                entry["kernel"] = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-80:]
                entry["calls"] = int(r["Calls"])
                entry["avg_ns"] = float(r["AverageNs"])
                entry["min_ns"] = float(r["MinNs"])
                entry["max_ns"] = float(r["MaxNs"])
                break
        bj = os.path.join(src, f"stats_{wl}.json")
        if os.path.exists(bj):
            try:
                entry["bench_line_under_profiler"] = json.loads(open(bj).read().strip().splitlines()[-1])
            except Exception:
                pass
    tr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"pmc_{wl}_{c}", "p_counter_collection.csv")
        if os.path.exists(f):
            avg, n = counter_avg(f, ksub)
            if c in avg:
                tr[c + "_KB"] = avg[c]
                tr["launches_" + c] = n[c]
    if "FETCH_SIZE_KB" in tr and "WRITE_SIZE_KB" in tr:
        tr["hbm_read_bytes"] = tr["FETCH_SIZE_KB"] * 1024 * 2
        tr["hbm_write_bytes"] = tr["WRITE_SIZE_KB"] * 1024
        tr["hbm_bytes"] = tr["hbm_read_bytes"] + tr["hbm_write_bytes"]
    entry["traffic"] = tr
    summary["workloads"][wl] = entry
# SQ counters of EVERY kernel of every workload (per-launch averages), so that each roofline fraction can be recomputed from
# tracked files: valu_issue_frac = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * duration * clock)
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][-80:]


def counters_by_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = short(r["Kernel_Name"])
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: dict({c: sum(v) / len(v) for c, v in d.items()}, launches=max(len(v) for v in d.values())) for k, d in agg.items()}


for wl in DOM:
    merged = {}
    for grp in ("sq1", "sq2", "sq3"):
        f = os.path.join(src, f"pmc_{grp}_{wl}", "p_counter_collection.csv")
        if os.path.exists(f):
            for k, d in counters_by_kernel(f).items():
                merged.setdefault(k, {}).update(d)
    if merged:
        summary.setdefault("sq_counters", {})[wl] = merged
# ---- per-STEP totals for bench.py's roofline block (profiles/counters_latest.json) -------------------------------------
def totals_by_kernel(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return tot, cnt


def is_step_kernel(name):
    return not ("at::native" in name or "reset" in name or "Functor" in name or "elementwise" in name)


latest = {"source": "profiles/<round>_summary.json of each workload's `round` (tools/profile_round.sh -> tools/summarize_profiles.py): "
                    "rocprofv3 --pmc, separate passes for FETCH_SIZE, WRITE_SIZE and the SQ group; "
                    "HBM bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md: gfx950 tallies 128-byte reads at 64 bytes); "
                    "per-step = total over all launches of the step's kernels / number of steps",
          "workloads": {}}
for wl in DOM:
    per = {}
    kernels = {}
    for label, sub, counter, scale in (("valu_insts_per_step", f"pmc_sq1_{wl}", "SQ_INSTS_VALU", 1.0),
                                       ("_fetch", f"pmc_{wl}_FETCH_SIZE", "FETCH_SIZE", 2048.0),
                                       ("_write", f"pmc_{wl}_WRITE_SIZE", "WRITE_SIZE", 1024.0)):
        f = os.path.join(src, sub, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        tot, cnt = totals_by_kernel(f)
        step_k = {k: v for k, v in tot.items() if is_step_kernel(k) and cnt[k].get(counter, 0) >= 10}
        if not step_k:
            continue
        n_steps = min(cnt[k][counter] for k in step_k)
        per[label] = sum(v.get(counter, 0.0) for v in step_k.values()) * scale / n_steps
        for k, v in step_k.items():
            e = kernels.setdefault(k, {})
            e["launches_per_step"] = cnt[k][counter] / n_steps
            e[counter + "_per_step"] = v.get(counter, 0.0) * (scale if counter != "SQ_INSTS_VALU" else 1.0) / n_steps
    # float64 arithmetic wave-instructions (add / mul / fma / transcendental): each occupies the SIMD for 4 cycles instead of 2
    f3 = os.path.join(src, f"pmc_sq3_{wl}", "p_counter_collection.csv")
    if os.path.exists(f3) and "valu_insts_per_step" in per:
        tot, cnt = totals_by_kernel(f3)
        names = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")
        step_k = {k: v for k, v in tot.items() if is_step_kernel(k) and cnt[k].get(names[0], 0) >= 10}
        if step_k:
            n_steps = min(cnt[k][names[0]] for k in step_k)
            per["f64_insts_per_step"] = sum(v.get(c, 0.0) for v in step_k.values() for c in names) / n_steps
            per["mfma_f32_insts_per_step"] = sum(v.get("SQ_INSTS_VALU_MFMA_F32", 0.0) for v in step_k.values()) / n_steps
            for k, v in step_k.items():
                e = kernels.setdefault(k, {})
                e["F64_INSTS_per_step"] = sum(v.get(c, 0.0) for c in names) / n_steps
    if "_fetch" in per and "_write" in per:
        per["hbm_bytes_per_step"] = per.pop("_fetch") + per.pop("_write")
    per = {k: v for k, v in per.items() if not k.startswith("_")}
    st = os.path.join(src, f"stats_{wl}", "p_kernel_stats.csv")
    if os.path.exists(st):
        for r in csv.DictReader(open(st)):
            k = short(r["Name"])
            if k in kernels:
                kernels[k]["avg_ns"] = float(r["AverageNs"])
    if per:
        per["kernels"] = kernels
        per["round"] = tag
        per["kernel_stamp"] = bench.kernel_stamp(wl)          # sources these kernels were compiled from (bench.KERNEL_SOURCES)
        per["library_stamp"] = _build.library_stamp()[:16]
        latest["workloads"][wl] = per
if latest["workloads"]:
    prev_path = os.path.join(dst, "counters_latest.json")
    if os.path.exists(prev_path):       # keep workloads that this round did not re-profile
        try:
            prev = json.load(open(prev_path))
            for k, v in prev.get("workloads", {}).items():      # a workload keeps the round it was last profiled in
                latest["workloads"].setdefault(k, v)
        except Exception:
            pass
    with open(prev_path, "w") as fh:
        json.dump(latest, fh, indent=1)
with open(os.path.join(dst, f"{tag}_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({wl: {"avg_us": e.get("avg_ns", 0) / 1e3, "hbm_MB": (e.get("traffic", {}).get("hbm_bytes") or 0) / 1e6}
                  for wl, e in summary["workloads"].items()}, indent=1))
