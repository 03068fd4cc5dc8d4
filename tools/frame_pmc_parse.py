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
    f.write("# FETCH / WRITE = the counters as read (MB per launch).  gfx950 tallies a 128-B request of a WIDE streaming read (16 B per\n"
            "# lane) at 64 B, so such kernels (NHWC fp16 maps) moved up to 2 x FETCH + WRITE (column 'corr'); kernels that read 4 B per\n"
            "# lane (planar float32 planes) are uncalibrated and closer to FETCH + WRITE (column 'raw').  GB/s = bytes / avg duration.\n")
    f.write(f"{'kernel':66s} {'calls':>6s} {'avg us':>8s} {'tot ms':>8s} {'FETCH':>8s} {'WRITE':>8s} {'raw GB/s':>9s} {'corr GB/s':>9s} {'corr % 8TB/s':>12s}\n")
    for r in rows:
        rawb = r['fetch_bytes'] + r['write_bytes']
        f.write(f"{r['kernel'][:66]:66s} {r['calls']:6d} {r['avg_us']:8.1f} {r['total_ms']:8.3f} {r['fetch_bytes'] / 1e6:8.1f} {r['write_bytes'] / 1e6:8.1f} "
                f"{rawb / (r['avg_us'] * 1e3):9.0f} {r['hbm_gbs']:9.0f} {100 * r['hbm_gbs'] / 8000:11.1f}%\n")
json.dump(rows, open(os.path.join(out, "hbm_table.json"), "w"), indent=1)
print(open(os.path.join(out, "hbm_table.txt")).read())
