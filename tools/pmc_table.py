#!/usr/bin/env python3
"""Per-kernel averages of every counter found in a set of rocprofv3 --pmc passes (CSV output).
usage: python tools/pmc_table.py <out.json> <dir or counter_collection.csv> [...]
Each pass is its own rocprofv3 run over the same command (the SQ block has 8 slots, TCC 4: MI355X_MICROARCH.md, PMC slots); values are
averaged per dispatch of a kernel (template arguments kept, so the qkv / out / fc1 / fc2 forms of gemm_pp stay apart).  SQ_* cycle
counters are in quad-cycles summed over all waves / SEs as the guide describes; derived ratios printed here use only same-unit pairs."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)E+v", name)
    if m:
        return m.group(1) + "<" + m.group(2) + ">"
    return name.split("(")[0][:120]


def files(args):
    for a in args:
        if os.path.isdir(a):
            for root, _, fs in os.walk(a):
                for f in fs:
                    if f.endswith("counter_collection.csv"):
                        yield os.path.join(root, f)
        else:
            yield a


def main():
    out = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in files(sys.argv[2:]):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k.startswith("at::") or "rocclr" in k or "Cijk" in k or k.startswith("__amd"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        res[k]["dispatches_sampled"] = max(len(v) for v in cs.values())
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    keys = sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0) * res[k]["dispatches_sampled"])[:10]
    for k in keys:
        v = res[k]
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        line = f"{k[:60]:60s} n={v['dispatches_sampled']:4d}"
        for name, key in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"), ("valu", "SQ_ACTIVE_INST_VALU"),
                          ("lds", "SQ_ACTIVE_INST_LDS"), ("wait_lds", "SQ_WAIT_INST_LDS")):
            if key in v:
                line += f" {name} {v[key] / wc:5.2f}"
        if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
            line += f" | lds conflict {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:5.3f} of active"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
            line += f" | mfma busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):5.2f}"
        if "TCC_HIT_sum" in v:
            line += f" | L2 hit {v['TCC_HIT_sum'] / max(1.0, v['TCC_HIT_sum'] + v.get('TCC_MISS_sum', 0)):5.2f}"
        print(line)


if __name__ == "__main__":
    main()
