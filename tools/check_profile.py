#!/usr/bin/env python3
"""Hold a rocprofv3 kernel summary against the figure bench.py printed in the SAME run (the roofline
must be reproducible from profiles/ alone).

  encoder: python tools/check_profile.py encoder <kernel_stats.csv> <bench.json> <out.json>
      sum(calls x avg) over the `gemm_pp_kernel` rows / sum(calls) of the stats CSV  vs  roofline.avg_launch_ms  (bench.py's
      HIP events inside the timed region).  The run must be the headline-only command
      (`bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-extra-legs --no-ceiling`) so every gemm_pp launch has the B = 32 shapes.
  decode:  python tools/check_profile.py decode <kernel_trace.csv> <bench.json> <out.json>
      per-dispatch trace of `bench.py --workload decode`: kernels are grouped by (name, grid, workgroup); the groups whose call
      count is a positive multiple of (warmup + steps) are the decode step; their total time / (warmup + steps)  vs  ms_per_step.
Writes a small JSON summary (what gets committed under profiles/) and exits non-zero beyond 3 % (encoder) / 6 % (decode: the
step also holds ~170 launch gaps that kernel durations do not include)."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:100]


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def encoder(stats_csv, bench_json, out):
    b = last_json_line(bench_json)
    rows = []
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            if "gemm_pp_kernel" in r["Name"]:
                rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"])))
    calls = sum(r[1] for r in rows)
    tot = sum(r[3] for r in rows)
    prof_ms = tot / calls / 1e6
    bench_ms = b["roofline"]["avg_launch_ms"]
    gflop = b["roofline"]["avg_launch_gflop"]
    res = {"what": "gemm_pp_kernel: rocprofv3 --kernel-trace --stats vs bench.py's HIP events, same run",
           "rows": [{"kernel": n, "calls": c, "avg_us": a / 1e3} for n, c, a, _ in rows],
           "profile_launches": calls, "profile_avg_launch_ms": prof_ms, "bench_avg_launch_ms": bench_ms, "bench_timed_launches": b["roofline"]["launches"],
           "ratio_profile_over_bench": prof_ms / bench_ms, "avg_launch_gflop": gflop,
           "tflops_from_profile": gflop / prof_ms, "frac_of_2500_from_profile": gflop / prof_ms / 2500.0,
           "bench_frac": b["roofline"]["frac"], "ms_per_step": b["ms_per_step"], "value_audio_s_per_s": b["value"]}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    return 0 if abs(prof_ms / bench_ms - 1.0) <= 0.03 else 1


def decode(trace_csv, bench_json, out):
    b = last_json_line(bench_json)
    n_steps = b["steps"] + b["warmup"]
    grp = collections.defaultdict(lambda: [0, 0.0])
    with open(trace_csv) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
            g = grp[key]
            g[0] += 1
            g[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    step_rows, other = [], 0.0
    for (name, grid, wg), (calls, ns) in grp.items():
        if calls >= n_steps and calls % n_steps == 0:
            step_rows.append({"kernel": name, "grid": grid, "wg": wg, "per_step": calls // n_steps, "avg_us": ns / calls / 1e3, "us_per_step": ns / n_steps / 1e3})
        else:
            other += ns
    step_rows.sort(key=lambda r: -r["us_per_step"])
    prof_ms = sum(r["us_per_step"] for r in step_rows) / 1e3
    bench_ms = b["ms_per_step"]
    launches = sum(r["per_step"] for r in step_rows)
    res = {"what": "decode step: sum of kernel durations per step (rocprofv3 --kernel-trace) vs bench.py --workload decode ms_per_step, same run",
           "steps_in_trace": n_steps, "kernels_per_step": launches, "rows": step_rows, "profile_kernel_ms_per_step": prof_ms, "bench_ms_per_step": bench_ms,
           "ratio_profile_over_bench": prof_ms / bench_ms, "gap_us_per_launch": (bench_ms - prof_ms) * 1e3 / max(1, launches),
           "bytes_per_step": b["roofline"]["bytes_per_step"], "GBps_from_profile": b["roofline"]["bytes_per_step"] / (prof_ms * 1e-3) / 1e9,
           "bench_frac_of_8TBps": b["roofline"]["frac"], "other_kernels_ms_total": other / 1e6}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "rows"}, indent=1))
    for r in step_rows[:12]:
        print(f"  {r['kernel'][:80]:80s} x{r['per_step']:3d}  {r['avg_us']:8.2f} us  {r['us_per_step']:9.1f} us/step")
    return 0 if 0.90 <= prof_ms / bench_ms <= 1.03 else 1


if __name__ == "__main__":
    mode = sys.argv[1]
    sys.exit({"encoder": encoder, "decode": decode}[mode](*sys.argv[2:5]))
