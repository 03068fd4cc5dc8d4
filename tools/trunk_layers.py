"""Per-LAYER device time of one trunk call from a rocprofv3 kernel trace: tools/probe_one_trunk.py writes the (label, route)
list of its last call in launch order (VSR_ROUTES_OUT); with every launch on ONE stream (VSR_FLOWSD_STREAM=0) the convolution
kernels of the trace's last call appear in that order, a split-K layer followed by its k_splitk_finish.
usage: trunk_layers.py kernel_trace.csv routes.txt -> "label<TAB>route<TAB>us" lines on stdout"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
labs = [l.rstrip("\n").split("\t") for l in open(sys.argv[2])]
def isconv(n): return any(k in n for k in ("k_conv_", "k_deconv4s2_patch", "k_stem7_rows", "k_conv1x1", "k_flow_head", "k_hg_front"))
conv = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if isconv(r["Kernel_Name"]) or "k_splitk_finish" in r["Kernel_Name"]]
# walk backwards: the last len(labs) layers
out, i = [], len(conv) - 1
for lab, route in reversed(labs):
    t = 0.0
    if "splitk" in route:
        assert "k_splitk_finish" in conv[i][0], (lab, route, conv[i][0]); t += conv[i][1]; i -= 1
    assert "k_splitk_finish" not in conv[i][0], (lab, route, conv[i][0])
    t += conv[i][1]; i -= 1
    out.append((lab, route, t))
for lab, route, t in reversed(out): print(f"{lab}\t{route}\t{t:.1f}")
