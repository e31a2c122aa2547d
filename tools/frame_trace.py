#!/usr/bin/env python3
"""rocprofv3 kernel trace of bench.py -> the kernels of the last frames in order: start offset inside the frame, duration, gap behind the
previous kernel (all us).  A frame starts at lz_k_get_rays."""
import csv, glob, os, sys
paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "lz_k_get_rays" in r[2]]
if len(starts) < 3:
    print("no frames found", len(rows)); sys.exit(0)
for fi in (-3, -2):
    a, b = starts[fi], starts[fi + 1]
    t0 = rows[a][0]
    print("frame", fi, "length %.1f us" % ((rows[b][0] - t0) / 1e3))
    prev_end = t0
    for s, e, n in rows[a:b]:
        print("  +%8.1f  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:90]))
        prev_end = e
