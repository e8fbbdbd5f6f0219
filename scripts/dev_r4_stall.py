"""Round-4 developer scratch: what the host was doing while a kernel of the command-line flow took 15-23 ms (rocprofv3 --hip-trace
--kernel-trace CSVs): HIP API calls that overlap the slowest kernel."""
import csv
import glob
import sys

d = sys.argv[1]
k = [r for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]))]
slow = max(k, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
s, e = int(slow["Start_Timestamp"]), int(slow["End_Timestamp"])
print("slowest kernel", slow["Kernel_Name"][:70], (e - s) / 1e3, "us")
for pat in ("hip_api_trace", "hsa_api_trace"):
    hits = glob.glob(d + f"/**/*{pat}.csv", recursive=True)
    if not hits:
        continue
    print("==", pat)
    for r in csv.DictReader(open(hits[0])):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b > s - 300000 and a < e + 100000 and (b - a > 20000 or "memory" in r["Function"] or "lock" in r["Function"] or "free" in r["Function"].lower() or "alloc" in r["Function"].lower()):
            print(f"  {r['Function'][:48]:48s} start {(a - s) / 1e3:10.1f} us  dur {(b - a) / 1e3:10.1f} us")
