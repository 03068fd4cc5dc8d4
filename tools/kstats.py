"""Prints a rocprofv3 kernel_stats.csv compactly: usage kstats.py file.csv [calls_divisor] [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6/div:.2f} ms, {sum(int(r['Calls']) for r in rows)/div:.0f} launches (per call)")
for r in rows[:top]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/div:7.1f} {float(r['TotalDurationNs'])/1e6/div:8.3f} ms {float(r['AverageNs'])/1e3:8.1f} us")
