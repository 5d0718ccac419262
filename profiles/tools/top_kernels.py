"""usage: top_kernels.py <kernel_stats.csv> [n]  -- the n largest kernels of a rocprofv3 --stats run"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {float(r["Percentage"]):6.2f}% n={r["Calls"]:>6} '
          f'avg={float(r["AverageNs"]) / 1e3:10.1f}us  {r["Name"][:90]}')
