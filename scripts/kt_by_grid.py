#!/usr/bin/env python3
"""rocprofv3 --kernel-trace: durations per (kernel name, grid size).

`--stats` averages every launch of a kernel name; bench.py launches ea_eval_poses_kernel in several shapes (the timed call,
the 8-pose launches of the batch table, the call-cost probes), so its one average mixes them.  This reads the per-dispatch
table (`*_kernel_trace.csv`) and prints count / mean / min / max per grid, which is what `roofline.kernel_ms` is to be
compared with.  usage: kt_by_grid.py <dir or csv> [substring of the kernel name]
"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "ea_eval"
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name", "")
                if want not in name:
                    continue
                if "Grid_Size_X" in r:
                    grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                else:
                    grid = (int(r.get("Grid_Size", 0) or 0),)
                wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                short = name.split("(")[0].replace("void ea::", "")
                rows.setdefault((short, grid, wg), []).append(dur)
    print("%-62s %-22s %5s %7s %10s %10s %10s" % ("kernel", "grid (threads)", "wg", "calls", "mean_ns", "min_ns", "max_ns"))
    for (name, grid, wg), d in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        print("%-62s %-22s %5d %7d %10.0f %10d %10d" % (name[:62], "x".join(map(str, grid)), wg, len(d), sum(d) / len(d), min(d), max(d)))


if __name__ == "__main__":
    main()
