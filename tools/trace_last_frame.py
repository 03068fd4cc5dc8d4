"""From a rocprofv3 kernel_trace.csv of tools/frame_workload.py: the launches of the LAST forward call in start order
(name, stream-less; duration, gap to the previous kernel's end).  A frame starts at the last k_resize_estimate launch.
usage: trace_last_frame.py run_kernel_trace.csv [substring filter]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
flt = sys.argv[2] if len(sys.argv) > 2 else None
start = max(i for i, r in enumerate(rows) if "k_resize_estimate" in r["Kernel_Name"])
fr = rows[start:]
print(f"{len(fr)} launches, {(int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e6:.2f} ms")
stock = [r for r in fr if "at::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"]]
print(f"stock launches: {len(stock)}, {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in stock) / 1e3:.1f} us")
for i, r in enumerate(fr):
    n = r["Kernel_Name"]
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:80]
    mark = "*" if ("at::" in n or "rocclr" in n or "rocprim" in n) else " "
    if flt is None or flt in n or mark == "*":
        print(f"{i:4d}{mark} {(int(r['Start_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3}  {short}")
