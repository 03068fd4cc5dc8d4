"""Aggregates tools/trunk_layers.sh outputs (label<TAB>route<TAB>us) into a ranked per-layer table with FLOPs and TFLOP/s.
usage: layer_table.py gpurun_out/tl_<tag>.txt [...] -> table on stdout (one section per file, then the routes' totals)"""
import re
import sys
from collections import OrderedDict, defaultdict


def flop(label):
    m = re.match(r"conv N(\d+) (\d+)x(\d+) c(\d+)->(\d+) k(\d+)x(\d+) s(\d+)", label)
    if m:
        n, h, w, ci, co, kh, kw, s = map(int, m.groups())
        if "stem" in label:
            ci = 3
        return 2.0 * n * (h // s) * (w // s) * ci * co * kh * kw
    m = re.match(r"deconv4s2 N(\d+) (\d+)x(\d+) c(\d+)->(\d+)", label)
    if m:
        n, h, w, ci, co = map(int, m.groups())
        return 2.0 * n * h * w * ci * co * 16
    m = re.match(r"hg_front N(\d+) (\d+)x(\d+) c4->128->(\d+)", label)
    if m:
        n, h, w, c2 = map(int, m.groups())
        return 2.0 * n * h * w * (3 * 128 * 49 + 128 * c2)
    m = re.match(r"flow_head N(\d+) (\d+)x(\d+) c(\d+)", label)
    if m:
        n, h, w, ci = map(int, m.groups())
        return 2.0 * n * h * w * ci * 18
    return 0.0


def main():
    grand = defaultdict(lambda: [0, 0.0, 0.0])
    for path in sys.argv[1:]:
        rows = OrderedDict()
        for line in open(path):
            parts = line.rstrip("\n").split("\t")
            if len(parts) != 3:
                continue
            lab, route, us = parts[0], parts[1], float(parts[2])
            r = rows.setdefault((lab, route), [0, 0.0])
            r[0] += 1
            r[1] += us
        tot = sum(v[1] for v in rows.values())
        totf = sum(flop(k[0]) * v[0] for k, v in rows.items())
        print(f"== {path}: {tot:.1f} us in {sum(v[0] for v in rows.values())} conv layers, {totf / 1e9:.1f} GFLOP, "
              f"{totf / tot / 1e6:.0f} TFLOP/s")
        for (lab, route), (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f = flop(lab) * cnt
            print(f"{lab:44s} x{cnt:<2d} {route:30s} {us:8.1f} us {100 * us / tot:5.1f} %  {f / us / 1e6 if us else 0:7.0f} TFLOP/s")
            g = grand[re.sub(r"\+splitk\d+", "+splitk", route)]
            g[0] += cnt
            g[1] += us
            g[2] += f
    print("== by route (all files)")
    for route, (cnt, us, f) in sorted(grand.items(), key=lambda kv: -kv[1][1]):
        print(f"{route:34s} x{cnt:<4d} {us:9.1f} us  {f / us / 1e6:7.0f} TFLOP/s")


if __name__ == "__main__":
    main()
