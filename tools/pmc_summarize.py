#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into per-kernel HBM-side bytes per launch.
usage: python tools/pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <key> [note]
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  FETCH counts L2-miss requests to the fabric,
Infinity-Cache hits included."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)E+v", name)
    if m:
        return m.group(1) + "<" + m.group(2) + ">"
    return name.split("(")[0][:120]


def load(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out, key = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    F, W = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        if k.startswith("at::") or "rocclr" in k or "Cijk" in k:
            continue
        rd = 2.0 * 1024.0 * sum(F.get(k, [0])) / max(1, len(F.get(k, [0])))
        wr = 1024.0 * sum(W.get(k, [0])) / max(1, len(W.get(k, [0])))
        res[k] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                  "launches_sampled": len(F.get(k, []))}
    try:
        with open(out) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {}
    doc[key] = res
    if note:
        doc["note_" + key] = note
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:14]:
        print(f"{k[:70]:70s} n={v['launches_sampled']:5d} read {v['hbm_read_bytes_per_launch']/1e6:9.1f} MB write {v['hbm_write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
