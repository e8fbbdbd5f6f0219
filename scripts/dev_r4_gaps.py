"""Round-4 developer scratch: gaps between the kernels of a replayed step, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 bench.py --steps 200 --no-extras --no-cpu-baseline
    python scripts/dev_r4_gaps.py OUT/t_kernel_trace.csv
"""
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void pccm::", "").split("<")[0] for r in rows]
# steps: sequences k_bin_count .. k_unit_lean
steps, cur = [], []
for r, nm in zip(rows, names):
    if nm == "k_bin_count" and cur:
        steps.append(cur)
        cur = []
    cur.append((nm, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
steps.append(cur)
full = [s for s in steps if [k[0] for k in s] == ["k_bin_count", "k_bin_scatter", "k_bin_sort", "k_brick_query", "k_grid_tail", "k_unit_lean"]]
full = full[len(full) // 4:]                          # the replayed ones
print(len(full), "steps")
for i in range(6):
    dur = statistics.median(s[i][2] - s[i][1] for s in full) / 1e3
    gap = statistics.median(s[i + 1][1] - s[i][2] for s in full) / 1e3 if i < 5 else float("nan")
    print(f"  {full[0][i][0]:16s} {dur:7.1f} us   gap to next {gap:6.1f}")
span = statistics.median(s[-1][2] - s[0][1] for s in full) / 1e3
period = statistics.median(b[0][1] - a[0][1] for a, b in zip(full, full[1:])) / 1e3
print(f"  first start -> last end {span:.1f} us; step period {period:.1f} us; idle between steps {period - span:.1f} us")
