#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> {kernel: {counter: per-launch value}} as JSON (stdout) + a table (stderr-free, after the JSON marker).

    python scripts/summarise_pmc.py <dir with pass directories> '<regex of pass directory names>' [out.json]

Per counter and kernel the value is the median of the upper half of its launches (warm-up launches of a kernel are the
low outliers of most counters, the cold ones the high outliers of the wait counters: the same rule as summarise_r03.py).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def counters(src, pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dirs = [d for d in os.listdir(src) if re.fullmatch(pattern, d) and os.path.isdir(os.path.join(src, d))]
    for f in [x for d in dirs for x in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)]:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            # a counter may come as several rows per dispatch (one per dimension instance): sum them
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for (did, cn), v in per_dispatch.items():
            acc[names[did]][cn].append(v)
    out = {}
    for k, cs in acc.items():
        if "pccm" not in k:
            continue
        out[k] = {}
        for c, v in cs.items():
            v = sorted(v)
            top = v[len(v) // 2:]
            out[k][c] = top[len(top) // 2]
        out[k]["launches_seen"] = max(len(v) for v in cs.values())
    return out


def main():
    src, pattern = sys.argv[1], sys.argv[2]
    c = counters(src, pattern)
    if len(sys.argv) > 3:
        json.dump(c, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k in sorted(c):
        print(k)
        for cn in sorted(c[k]):
            print("   %-32s %18.1f" % (cn, c[k][cn]))


if __name__ == "__main__":
    main()
