#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately:
TCC has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2).  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; FETCH_SIZE under-reports a wide coalesced
read stream by exactly 2x on gfx950, so the upper bound on read bytes is 2 x FETCH_SIZE.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            acc[name][1] += 1
    return acc


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    print("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch, reads x1 | reads x2 (gfx950 wide-stream correction) |")
    print("|---|---|---|---|---|---|")
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 1])[0] + w.get(k, [0, 1])[0])):
        fk, fn = f.get(k, [0.0, 1]); wk, wn = w.get(k, [0.0, 1])
        fpl, wpl = fk / max(fn, 1), wk / max(wn, 1)
        out[k] = {"launches": max(fn, wn), "fetch_kib_per_launch": fpl, "write_kib_per_launch": wpl,
                  "hbm_bytes_per_launch_x1": (fpl + wpl) * 1024, "hbm_bytes_per_launch_x2": (2 * fpl + wpl) * 1024}
        print(f"| `{k}` | {max(fn, wn)} | {fpl:.1f} | {wpl:.1f} | {(fpl + wpl) * 1024:.3e} | {(2 * fpl + wpl) * 1024:.3e} |")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
