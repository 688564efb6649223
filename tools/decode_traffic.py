#!/usr/bin/env python3
"""HBM-side bytes per decode STEP from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV) over
`AFHIP_DECODE_GRAPH=0 python3 tools/decode_probe.py B steps ctx` with ONE weight format (PROBE_BF16_ONLY=1 / PROBE_FP8_ONLY=1).
usage: python tools/decode_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <label e.g. B8_bf16>
Per kernel of the decode step (the launches whose grid is that kernel's most frequent grid: prefill launches of a shared kernel drop out):
mean bytes per launch x launches per step (its launch count / the count of pick_kernel, one per step).  Corrections as
MI355X_MICROARCH.md prescribes: both counters in KiB, FETCH_SIZE doubled on gfx950, WRITE_SIZE as is."""
import collections
import csv
import json
import re
import sys

# the kernels of the decode step; rmsnorm / gemm_pp / rope launches in the same trace belong to the prefill
STEP_KERNELS = ("img_phase_kernel", "attn_kernel", "attn_decode128_kernel", "attn_combine_kernel", "embed_kernel", "pick_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:160]


def load(path, counter):
    rows = collections.defaultdict(list)          # kernel -> [(grid, value)]
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if not any(s in k for s in STEP_KERNELS):
                continue
            rows[k].append((r.get("Grid_Size", ""), float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        vals = [x for g, x in v if g == grid]
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    fetch, write, out, label = sys.argv[1:5]
    F, W = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    steps = max([n for k, (_, n) in F.items() if "pick_kernel" in k or "decode_update_kernel" in k] or [1])
    per_step, kernels = 0.0, {}
    for k in sorted(set(F) | set(W)):
        rd, n = F.get(k, (0.0, 0))
        wr, n2 = W.get(k, (0.0, 0))
        n = max(n, n2)
        per_launch = 2.0 * 1024.0 * rd + 1024.0 * wr
        cnt = round(n / steps)
        if cnt == 0 or n % steps != 0:            # not once (or k times) per step: a prefill launch of a kernel the step shares a name with
            continue
        kernels[k] = {"hbm_read_bytes_per_launch": 2048.0 * rd, "hbm_write_bytes_per_launch": 1024.0 * wr, "launches_per_step": cnt, "launches_sampled": n}
        per_step += per_launch * cnt
    try:
        doc = json.load(open(out))
    except (OSError, ValueError):
        doc = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches) on tools/decode_probe.py (AF3-7B shape, context ~800); "
                       "FETCH doubled per MI355X_MICROARCH.md; bytes are L2-to-fabric traffic per decode STEP = sum over the step's kernels of mean bytes per "
                       "launch x launches per step (tools/decode_traffic.py)", "per_step_bytes": {}, "kernels": {}}
    doc["per_step_bytes"][label] = per_step
    doc["kernels"][label] = kernels
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{label}: {steps} steps sampled, {per_step / 1e9:.3f} GB per step")
    for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1]["hbm_read_bytes_per_launch"] + kv[1]["hbm_write_bytes_per_launch"]) * kv[1]["launches_per_step"])[:8]:
        print(f"   {k[:90]:90s} x{v['launches_per_step']:3d}  read {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:7.2f} MB")


if __name__ == "__main__":
    main()
