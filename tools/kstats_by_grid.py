"""Kernel statistics of a rocprofv3 --kernel-trace CSV grouped by (kernel name, grid size): launches of one kernel with different
geometries (k_utd3 on 5 planes vs on 3 planes beside the trunks; a convolution kernel across layers) stay separable.
usage: kstats_by_grid.py kernel_trace.csv [calls_divisor] [top] [name filter]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
flt = sys.argv[4] if len(sys.argv) > 4 else ""
g = defaultdict(list)
for r in rows:
    if flt and flt not in r["Kernel_Name"]:
        continue
    if "Grid_Size" in r:      # (--pmc counter_collection.csv)
        grid, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
    else:                      # (--kernel-trace kernel_trace.csv: per-dimension columns)
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    g[(r["Kernel_Name"], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in g.values())
print(f"total {tot / 1e6 / div:.2f} ms, {sum(len(v) for v in g.values()) / div:.0f} launches (per call); columns: kernel, grid threads, launches per call, ms per call, avg us, min us")
for (name, grid, wg), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{name[:78]:78s} {grid:9d} {len(v) / div:7.1f} {sum(v) / 1e6 / div:8.3f} ms {sum(v) / len(v) / 1e3:8.1f} us {min(v) / 1e3:8.1f} us")
