"""Prints a rocprofv3 kernel_trace.csv as runs of the same kernel in launch order: name, count, average duration, average gap to
the previous kernel's end.  usage: trace_seq.py run_kernel_trace.csv [min_count]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
runs, prev_end = [], None
for r in rows:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
    gap = s - prev_end if prev_end is not None else 0
    prev_end = e
    g = f'{r.get("Grid_Size_X", "?")}x{r.get("Grid_Size_Y", "?")}x{r.get("Grid_Size_Z", "?")}'
    if runs and runs[-1][0] == n and runs[-1][4] == g:
        runs[-1][1] += 1; runs[-1][2] += e - s; runs[-1][3] += gap
    else:
        runs.append([n, 1, e - s, gap, g])
for n, c, d, gp, g in runs:
    if c >= minc:
        print(f"{n[:70]:70s} grid {g:18s} x{c:4d}  {d / c / 1e3:8.1f} us  gap {gp / c / 1e3:6.1f} us")
