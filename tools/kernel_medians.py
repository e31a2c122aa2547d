#!/usr/bin/env python3
"""Per-kernel launch durations of a rocprofv3 --kernel-trace run WITHOUT the warm-up launches: the summary committed under profiles/.

    python tools/kernel_medians.py <dir with *kernel_trace.csv> [--skip W] > profiles/rN_<what>_kernel_stats.csv

rocprofv3's own `--stats` table averages every launch of the process, warm-ups and first (cold, lazily loaded) launches included: round 4's
summary read 9.26 ms for a kernel whose timed launches took 8.96 (Weak 5).  This table reports, per kernel, the number of
launches, the MEDIAN, the mean over all launches but the first W of that kernel (default 2 = bench.py's --warmup in tools/profile_bench.sh),
min and max -- `roofline.avg_launch_ms` of the bench line is the mean of the timed launches and has to agree with MeanSteadyNs to 2 %."""
import csv
import glob
import os
import statistics
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 2
    args = [a for a in args if not a.isdigit() or os.path.isdir(a)]
    per = {}
    for d in args:
        for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(p, newline="")):
                per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    total = sum(sum(d for _, d in v) for v in per.values()) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "MedianNs", "MeanSteadyNs", "SteadyCalls", "MinNs", "MaxNs", "Percentage"])
    for name, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        v.sort()
        dur = [d for _, d in v]
        steady = dur[skip:] if len(dur) > skip else dur
        w.writerow([name, len(dur), sum(dur), int(statistics.median(dur)), int(sum(steady) / len(steady)), len(steady), min(dur), max(dur),
                    round(100.0 * sum(dur) / total, 4)])


if __name__ == "__main__":
    main()
