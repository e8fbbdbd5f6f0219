#!/usr/bin/env python3
"""Raw rocprofv3 output of scripts/prof_r03.sh (gpurun_out/r03prof) -> the summaries kept under profiles/r03/.

    python scripts/summarise_r03.py gpurun_out/r03prof profiles/r03
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(src, pattern):
    """{kernel: {counter: median of the upper half of its launches}} over the pmc passes whose directory matches `pattern`."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    import re
    dirs = [d for d in os.listdir(src) if re.fullmatch(pattern, d) and os.path.isdir(os.path.join(src, d))]
    for f in [x for d in dirs for x in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
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


def traffic(ctrs, points, what):
    out = {}
    for k, v in ctrs.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fe, wr = v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
            # MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of wide coalesced streaming reads on gfx950
            out[k.split("<")[0].split("::")[-1] + ("<" + k.split("<", 1)[1] if "<" in k else "")] = {
                "points": points, "fetch_size_bytes": fe, "write_size_bytes": wr, "fetch_plus_write_bytes": fe + wr,
                "hbm_bytes_per_launch": 2 * fe + wr,
                "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `{what}`, per launch; "
                        "hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 correction for wide streaming reads) + WRITE_SIZE"}
    return out


def stats(src, name, dst, label):
    hits = glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, f"{label}_kernel_stats.csv"))
    log = os.path.join(src, name + ".log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{label}_bench_line_under_rocprof.json"), "w"), indent=1)


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for name, label in (("stats_graph", "graph_1M"), ("stats_eager", "eager_1M"), ("stats_brute", "brute_1M"), ("stats_8M", "graph_8M"),
                        ("stats_content", "content_0.8M")):
        stats(src, name, dst, label)
    c1 = counters(src, r"pmc\d+")
    json.dump(c1, open(os.path.join(dst, "grid_1M_pmc_counters.json"), "w"), indent=1, sort_keys=True)
    t = traffic(c1, 1000000, "bench.py --no-graph --no-extras")
    # bench.py quotes the k_brick_query entry under this short key
    for k in list(t):
        if k.startswith("k_brick_query") and "k_brick_query" not in t:
            t["k_brick_query"] = t[k]
    json.dump(t, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    c8 = counters(src, r"pmc8M_.*")
    json.dump(traffic(c8, 8000000, "bench.py --points 8000000 --no-graph --no-extras"), open(os.path.join(dst, "pmc_traffic_8M.json"), "w"), indent=1, sort_keys=True)
    cc = counters(src, r"pmcC_.*")
    json.dump(cc, open(os.path.join(dst, "content_0.8M_pmc_counters.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic(cc, 800000, "bench.py --content-only --no-graph"), open(os.path.join(dst, "content_0.8M_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for fn, label in (("bench_line.json", "grid_1M_bench_line.json"), ("bench_8M.json", "grid_8M_bench_line.json"),
                      ("bench_32M.json", "grid_32M_bench_line.json")):
        path = os.path.join(src, fn)
        if os.path.exists(path):
            text = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
            if text:
                json.dump(json.loads(text[-1]), open(os.path.join(dst, label), "w"), indent=1)
    rp = os.path.join(src, "rank_profile.json")
    if os.path.exists(rp):
        shutil.copy(rp, os.path.join(dst, "rank_profile.json"))
    for k in sorted(c1):
        if any(s in k for s in ("k_brick_query", "k_bin_", "k_unit_lean", "k_grid_finish")):
            print(k)
            for cn in sorted(c1[k]):
                print("   %-32s %16.1f" % (cn, c1[k][cn]))
    for name, tt in (("1M", t), ("8M", traffic(c8, 8000000, "")), ("content", traffic(cc, 800000, ""))):
        for k, v in tt.items():
            print(name, k, "FETCH %.1f MB  WRITE %.1f MB" % (v["fetch_size_bytes"] / 1e6, v["write_size_bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
