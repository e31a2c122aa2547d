#!/usr/bin/env python3
"""GPU idle time inside the cfg3 training step from a rocprofv3 kernel trace of `bench.py --train-only`: per step (a step starts at
lz_k_near_far) the span, the sum of kernel durations, the idle remainder and the number of launches.
    python tools/train_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("lz_k_near_far")]
for a, b in list(zip(starts, starts[1:]))[-6:]:
    span = rows[b][0] - rows[a][0]
    busy = 0
    end = rows[a][0]
    for s, e, _ in rows[a:b]:
        s2 = max(s, end)
        if e > s2:
            busy += e - s2
            end = e
    gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[a:i + 1]), rows[i][2][:50], rows[i + 1][2][:50]) for i in range(a, b - 1)), reverse=True)[:5]
    print("step: span %.1f us, busy %.1f us, idle %.1f us, %d launches; largest gaps: %s" % (span / 1e3, busy / 1e3, (span - busy) / 1e3, b - a,
          [(round(g / 1e3, 1), x, y) for g, x, y in gaps]))
