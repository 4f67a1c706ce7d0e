"""bench.py's LAST stdout line is the driver's measurement channel: it must stay small enough to be parsed (the 23 KB
line of round 4 gave BENCH_r04.json `parsed: null`).  Built here from a real full result (profiles/r04p_bench_steps20.json, the
line round 4 printed) and from a synthetic worst case -- no GPU."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _round4_result():
    with open(os.path.join(ROOT, "profiles", "r04p_bench_steps20.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_final_line_from_the_round4_result_is_small_and_complete():
    full = _round4_result()
    assert len(json.dumps(full)) > 20000                      # the line that was not parsed
    line = bench.final_line(full)
    assert "\n" not in line and len(line) < bench.MAX_LINE == 4096
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] == bench._r(full["value"]) and d["config"]["workload"] == full["config"]["workload"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_step", "kernel_avg_ns", "step_event_ns",
              "counters_stale", "counters_live", "valu_wave_insts_per_step", "hbm_frac", "valu_issue_frac", "hbm_copy_measured_GBps"):
        assert k in rf, k
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-3 and rf["bound"] == full["roofline"]["bound"]
    assert abs(rf["step_event_ns"] - full["roofline"]["step_ms"] * 1e6) < 1.0          # the bench's own HIP-event time per step
    # `frac` can be recomputed from the line alone: wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz x step time)
    if rf["bound"] == "valu_issue":
        again = rf["valu_wave_insts_per_step"] * 2 / (1024 * 2.4e9 * rf["step_event_ns"] * 1e-9)
        assert abs(again - rf["frac"]) < 2e-3
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["all_cores"]["logical_cpus"] == full["cpu_baseline"]["all_cores"]["logical_cpus"]
    names = [s["name"] for s in d["secondary"]]
    for must in ("ns2d_c4_b4096", "ns2d_c4_f64_b4096", "transport_c3", "ns2d_c4_f64", "ns2d_c5_f64"):
        assert must in names
    for s in d["secondary"]:
        assert set(s) == {"name", "value", "ms_per_step", "dtype", "bound", "frac"}
        assert abs(s["value"] / full["also"][s["name"]]["value"] - 1) < 1e-5
    assert "also" not in d and "timed_regions_s" not in d


def test_final_line_stays_under_the_cap_when_names_and_notes_grow():
    full = _round4_result()
    full["config"]["workload"] = "W" * 3000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for k in list(full["also"]):
        full["also"][k + "_" + "x" * 200] = full["also"][k]
    line = bench.final_line(full)
    assert len(line) < bench.MAX_LINE
    d = json.loads(line)
    assert d["value"] == bench._r(full["value"]) and d["roofline"]["bound"] and d["cpu_baseline"]["value"]


def test_final_line_multi_gpu_fields_fit():
    full = _round4_result()
    full.pop("also")
    full["n_gpus"] = 8
    full["per_rank_env_steps_per_s"] = [2.0912345678e8 + i for i in range(8)]
    full["n1_equivalent"] = {"value": 2.09e8, "note": "x" * 300}
    full["process_group"] = "nccl"
    line = bench.final_line(full)
    d = json.loads(line)
    assert len(line) < bench.MAX_LINE and len(d["per_rank_env_steps_per_s"]) == 8 and d["process_group"] == "nccl"
    assert d["n1_equivalent"] == 2.09e8 and "secondary" not in d


def test_final_line_survives_an_error_in_a_secondary_workload_and_missing_counters():
    full = _round4_result()
    full["also"]["ns2d_c5_f64"] = {"error": "RuntimeError('boom')" * 20}
    full["roofline"] = {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "step_ms": 0.02}
    d = json.loads(bench.final_line(full))
    assert any(s["name"] == "ns2d_c5_f64" and "error" in s for s in d["secondary"])
    assert d["roofline"]["frac"] is None and d["roofline"]["traffic"] is None


def test_hbm_probe_library_builds_and_exports_its_entry_point():
    """tools/hbm_probe.hip is built by __graft_entry__.build() into its own small library (not part of the product ABI); no GPU is
    needed to load it and find the symbol."""
    import ctypes
    from pdecontrolgym_amd import build
    lib = ctypes.CDLL(build.build_probe())
    assert hasattr(lib, "pdegym_probe_hbm")
    # ... and it does not touch the product library's fingerprint (it lives outside csrc/)
    assert not os.path.exists(os.path.join(build.CSRC, "hbm_probe.hip"))


def test_final_line_carries_live_counters_and_rocprof_kernel_average():
    """A run whose rocprofv3 children delivered counters (bench.live_counters): counters_live is true, kernel_avg_ns is rocprof's
    average of the dominant kernel (not the bench's event time) and frac follows from the line's own numbers."""
    full = _round4_result()
    live = {"collected": "rocprofv3 children", "kernel": "step1d_kernel<4, true, false, false, false, false, true>", "calls": 77,
            "kernel_avg_ns": 16335.0, "valu_insts_per_step": 11141120.0, "hbm_bytes_per_step": 14.28e6}

    class WL:
        def algorithmic_bytes_per_step(self):
            return 1267482624

        def compulsory_bytes_per_step(self):
            return 12_800_000
    rf = bench.roofline_block(WL(), "parabolic_c2", 0.018010, True, live)
    assert rf["counters_live"] is True and rf["kernel_avg_ns"] == 16335.0 and rf["traffic"] == 14.28e6 and rf["bound"] == "valu_issue"
    assert abs(rf["frac"] - 11141120.0 * 2 / (1024 * 2.4e9 * 18010e-9)) < 1e-6
    full["roofline"] = rf
    d = json.loads(bench.final_line(full))["roofline"]
    assert d["counters_live"] is True and d["kernel_avg_ns"] == 16335.0 and abs(d["step_event_ns"] - 18010.0) < 1.0
    assert abs(d["valu_wave_insts_per_step"] * 2 / (1024 * 2.4e9 * d["step_event_ns"] * 1e-9) - d["frac"]) < 2e-3
    # without live counters the committed ones serve and the line says so
    rf2 = bench.roofline_block(WL(), "parabolic_c2", 0.018010, True, None)
    assert rf2["counters_live"] is False


def test_final_line_single_env_block_and_multi_gpu_secondary():
    full = _round4_result()
    with open(os.path.join(ROOT, "profiles", "r06a_single_env_after.json")) as fh:
        full["also"]["single_env"] = json.load(fh)
    line = bench.final_line(full)
    d = json.loads(line)
    assert len(line) < bench.MAX_LINE
    se = d["single_env_us_per_step"]
    assert set(se) == {"transport_c1", "parabolic_example", "transport_s1", "ns2d_example", "traffic_example", "tumor_example"}
    for gpu, gpu_nohist, cpu in se.values():
        assert gpu > 0 and gpu_nohist > 0 and cpu > 0
    assert se["transport_c1"][0] < se["transport_c1"][2] and se["parabolic_example"][0] < se["parabolic_example"][2]
    assert d["single_env_gpu_wins_above_substeps"] > 0
    # an N > 1 run: the C5 shard workloads carry the node total and every rank's own rate
    multi = _round4_result()
    multi["n_gpus"] = 8
    multi["also"] = {n: {"value": 8e6, "ms_per_step": 0.5, "dtype": dt, "per_rank_env_steps_per_s": [1.01e6 + i for i in range(8)],
                         "instances_per_gpu": 512, "roofline": {"bound": None, "frac": None}}
                     for n, dt in (("ns2d_c5", "f32"), ("ns2d_c5_f64", "f64"))}
    line = bench.final_line(multi)
    sec = {e["name"]: e for e in json.loads(line)["secondary"]}
    assert len(line) < bench.MAX_LINE and set(sec) == {"ns2d_c5", "ns2d_c5_f64"}
    assert all(len(e["per_rank"]) == 8 and e["instances_per_gpu"] == 512 for e in sec.values())


def test_live_counters_parse_rocprof_csvs_and_fail_soft(tmp_path, monkeypatch):
    """bench.live_counters_from_csvs on synthetic rocprofv3 tables (the layout tools/summarize_profiles.py reads): per-launch averages
    of the named kernel only, the gfx950 read-side factor 2, None when a pass saw too few launches; and live_counters() itself returns
    None -- the committed counters then serve -- when the profiler child fails."""
    k = "void (anonymous namespace)::step1d_kernel<4, true, false, false, false, false, true>(pdegym_params1d, pdegym_bufs1d, int)"
    other = "void at::native::elementwise_kernel<...>(int)"
    stats = tmp_path / "p_kernel_stats.csv"
    stats.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                     f'"{other}",5,500,100.0,1.0,90,110,1.0\n"{k}",47,752100,16002.1,99.0,15800,17000,20.0\n')

    def pmc(name, counter, values):
        f = tmp_path / f"{name}.csv"
        rows = ['"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
        for i, v in enumerate(values):
            rows.append(f'{i},{i},1,1,1,1,1048576,3,"{k}",256,0,0,40,0,96,"{counter}",{v},0,1')
        rows.append(f'99,99,1,1,1,1,64,4,"{other}",64,0,0,8,0,16,"{counter}",123456789,0,1')
        f.write_text("\n".join(rows) + "\n")
        return str(f)
    paths = {"stats": str(stats), "SQ_INSTS_VALU": pmc("valu", "SQ_INSTS_VALU", [11141120] * 12),
             "FETCH_SIZE": pmc("fetch", "FETCH_SIZE", [4700.0] * 12), "WRITE_SIZE": pmc("write", "WRITE_SIZE", [4540.0] * 12)}
    got = bench.live_counters_from_csvs(paths, "step1d_kernel")
    assert got["kernel"].startswith("step1d_kernel<4, true") and got["calls"] == 47 and got["kernel_avg_ns"] == 16002.1
    assert got["valu_insts_per_step"] == 11141120.0
    assert got["hbm_bytes_per_step"] == 4700.0 * 2048 + 4540.0 * 1024
    assert bench.live_counters_from_csvs(dict(paths, WRITE_SIZE=pmc("few", "WRITE_SIZE", [1.0] * 3)), "step1d_kernel") is None
    assert bench.live_counters_from_csvs(paths, "no_such_kernel") is None
    # the collection itself: a failing profiler child must not take the bench run down
    import subprocess

    import sys
    real_popen = subprocess.Popen
    seen = []

    def failing_child(cmd, **kw):            # stands in for rocprofv3: exits 3 at once
        seen.append((cmd, kw))
        return real_popen([sys.executable, "-c", "import sys; sys.exit(3)"], **kw)
    monkeypatch.setattr(subprocess, "Popen", failing_child)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: True)
    assert bench.live_counters("parabolic_c2") is None
    cmd, kw = seen[0]
    assert kw.get("start_new_session") is True                       # a timeout can take the whole group down
    i = cmd.index("--")
    assert cmd[i + 1] == "python3" and "--pmc" not in cmd[:i]         # the program itself directly after `--`; first pass: stats only
    assert "--no-live-counters" in cmd and "--kernel-trace" in cmd and "--sys-trace" not in cmd

    def hanging_child(cmd, **kw):            # ... and one that never finishes: killed with its group, None returned
        return real_popen([sys.executable, "-c", "import time; time.sleep(60)"], **kw)
    monkeypatch.setattr(subprocess, "Popen", hanging_child)
    t0 = bench.time.perf_counter()
    assert bench.live_counters("parabolic_c2", budget_s=7.0, first_timeout_s=1.0) is None
    assert bench.time.perf_counter() - t0 < 10
    assert bench.live_counters("ns2d_c4") is None             # only the headline has a live path
