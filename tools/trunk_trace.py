"""From a rocprofv3 kernel_trace.csv of tools/probe_one_trunk.py: the launches of the LAST call (the final 1/reps of the
dispatches) in start order with duration, gap to the previous kernel's end and grid size; then totals per kernel name.
usage: trunk_trace.py run_kernel_trace.csv <marker kernel substring> [all]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]     # substring of the first kernel of a call (flow: k_pair_sums)
start = max(i for i, r in enumerate(rows) if marker in r["Kernel_Name"])
fr = rows[start:]
n = len(fr)
t0 = int(fr[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr)
span = int(fr[-1]["End_Timestamp"]) - t0
print(f"{n} launches, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms")
prev_end = t0
agg = {}
for i, r in enumerate(fr):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    short = nm.split("(")[0][:46]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
    if len(sys.argv) > 3:
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  wgs {g:6d}  {short}")
    prev_end = max(prev_end, e)
    a = agg.setdefault(short, [0, 0])
    a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:48s} x{c:4d} {t / 1e3:9.1f} us  avg {t / c / 1e3:7.1f}")
