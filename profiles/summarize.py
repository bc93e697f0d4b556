#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small tables kept under profiles/.

    python profiles/summarize.py trace  <rocprof_dir> <out.csv>     per (kernel, grid size): calls, avg/min/max ms
    python profiles/summarize.py pmc    <rocprof_dir> <out.csv>     per (kernel, grid size, counter): launches, mean value
"""
import collections
import csv
import glob
import os
import sys


def trace(d, out):
    path = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        agg[(r["Kernel_Name"], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size_threads", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"])
        for (k, g), v in rows:
            w.writerow([k, g, len(v), f"{sum(v):.4f}", f"{sum(v) / len(v):.5f}", f"{min(v):.5f}", f"{max(v):.5f}"])


def pmc(d, out):
    path = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    rows = sorted(agg.items(), key=lambda kv: -max(kv[1]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size_threads", "counter", "launches", "mean", "max"])
        for (k, g, c), v in rows:
            w.writerow([k, g, c, len(v), f"{sum(v) / len(v):.3f}", f"{max(v):.3f}"])


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
