#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of a profiling call (gpurun_out/<dir>) into the summaries kept under profiles/.

    python scripts/summarise_profiles.py gpurun_out/r01b profiles/r01

Expects, as written by the commands listed in profiles/r01/README.md:
    <dir>/grid_stats/*_kernel_stats.csv     --kernel-trace --stats run of bench.py (grid engine, hipGraph)
    <dir>/brute_stats/*_kernel_stats.csv    the same for --engine brute
    <dir>/grid_fetch|grid_write/*_counter_collection.csv   separate --pmc FETCH_SIZE / WRITE_SIZE passes (eager)
    <dir>/bench_line.json                   the JSON line of a plain bench.py run
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def one(pattern):
    hits = glob.glob(pattern)
    return hits[0] if hits else None


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        v = sorted(v)
        top = v[len(v) // 2:]                     # first launches of a process see cold caches / short warm-up shapes
        out[k] = top[len(top) // 2]
    return out


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for eng in ("grid", "brute"):
        f = one(os.path.join(src, f"{eng}_stats", "*_kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(dst, f"{eng}_1M_kernel_stats.csv"))
    fetch = one(os.path.join(src, "grid_fetch", "*_counter_collection.csv"))
    write = one(os.path.join(src, "grid_write", "*_counter_collection.csv"))
    if fetch and write:
        f, w = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
        table = {k: {"FETCH_SIZE_KB_median_upper_half": f[k], "WRITE_SIZE_KB_median_upper_half": w.get(k, 0.0)} for k in f}
        json.dump(table, open(os.path.join(dst, "grid_1M_pmc_fetch_write.json"), "w"), indent=1)
        key = next(k for k in table if "k_grid_query_coop" in k)
        fe, wr = table[key]["FETCH_SIZE_KB_median_upper_half"] * 1024, table[key]["WRITE_SIZE_KB_median_upper_half"] * 1024
        # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950
        traffic = {"k_grid_query_coop": {
            "points": 1000000, "fetch_size_bytes": fe, "write_size_bytes": wr, "hbm_bytes_per_launch": 2 * fe + wr,
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --no-graph`, per launch "
                    "(both directions): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE; profiles/r01/grid_1M_pmc_fetch_write.json"}}
        json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
        print("k_grid_query_coop: FETCH %.1f MB (x2 = %.1f MB), WRITE %.1f MB per launch" % (fe / 1e6, 2 * fe / 1e6, wr / 1e6))
    line = os.path.join(src, "bench_line.json")
    if os.path.exists(line):
        text = [ln for ln in open(line).read().splitlines() if ln.startswith("{")][-1]
        json.dump(json.loads(text), open(os.path.join(dst, "grid_1M_bench_line.json"), "w"), indent=1)
    for name, log in (("grid_1M_bench_line_under_rocprof.json", "grid_stats.log"), ("brute_1M_bench_line_under_rocprof.json", "brute_stats.log")):
        p = os.path.join(src, log)
        if os.path.exists(p):
            lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(dst, name), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
