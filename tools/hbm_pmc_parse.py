"""Parses the two passes of tools/hbm_pmc.sh -> JSON on stdout: per-launch HBM traffic of the kernel whose name holds argv[2],
by grid size (a kernel launched with several geometries in one run is reported per geometry)."""
import csv, glob, json, os, sys
out, ksub, alg, what = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
by = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if ksub not in r["Kernel_Name"]:
                continue
            g = by.setdefault((r["Kernel_Name"][:80], int(r["Grid_Size"]), int(r["Workgroup_Size"])), {})
            g.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            g.setdefault("dur_" + r["Counter_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = []
for (name, grid, wg), g in sorted(by.items(), key=lambda kv: -kv[0][1]):
    med = lambda k: sorted(g[k])[len(g[k]) // 2]
    if "FETCH_SIZE" not in g or "WRITE_SIZE" not in g:
        continue
    fetch_kb, write_kb = med("FETCH_SIZE"), med("WRITE_SIZE")
    traffic = (2 * fetch_kb + write_kb) * 1024
    res.append(dict(kernel=name, grid_threads=grid, workgroup=wg, launches_seen=len(g["FETCH_SIZE"]), FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
                    traffic_bytes_per_launch=traffic, algorithmic_bytes_per_launch=alg, ratio=round(traffic / alg, 3) if alg else None,
                    dispatch_ns_under_pmc=med("dur_FETCH_SIZE")))
json.dump(dict(workload=what, correction="gfx950: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads -> doubled; WRITE_SIZE exact "
                                         "(MI355X_MICROARCH.md, HBM)", hbm=res[0] if res else None, all_geometries=res), sys.stdout, indent=1)
