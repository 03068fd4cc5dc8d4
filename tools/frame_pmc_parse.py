"""Per-kernel HBM traffic and rate from the passes of tools/frame_pmc.sh -> <outdir>/hbm_table.txt (+ .json).
traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies a 128-B streaming read request at 64 B; WRITE_SIZE is exact)."""
import csv, glob, json, os, re, sys
out = sys.argv[1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


cnt = {}
for name in ("fetch", "write"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
rows = []
for k, (calls, avg_ns, tot) in dur.items():
    c = cnt.get(k, {})
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fe = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024
    wr = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    traffic = 2 * fe + wr
    rows.append(dict(kernel=k, calls=calls, avg_us=avg_ns / 1e3, total_ms=tot / 1e6, fetch_bytes=fe, write_bytes=wr, traffic_bytes=traffic,
                     hbm_gbs=traffic / avg_ns))
rows.sort(key=lambda r: -r["total_ms"])
with open(os.path.join(out, "hbm_table.txt"), "w") as f:
    f.write(f"{'kernel':70s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'MB/launch':>10s} {'HBM GB/s':>9s} {'% of 8 TB/s':>11s}\n")
    for r in rows:
        f.write(f"{r['kernel']:70s} {r['calls']:6d} {r['avg_us']:9.1f} {r['total_ms']:9.3f} {r['traffic_bytes'] / 1e6:10.1f} {r['hbm_gbs']:9.0f} "
                f"{100 * r['hbm_gbs'] / 8000:10.1f}%\n")
json.dump(rows, open(os.path.join(out, "hbm_table.json"), "w"), indent=1)
print(open(os.path.join(out, "hbm_table.txt")).read())
