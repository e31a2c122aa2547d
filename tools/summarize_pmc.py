#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (…_counter_collection.csv, one row per dispatch and counter) per kernel and counter.

    python tools/summarize_pmc.py <dir with csv files> [more dirs ...] > profiles/rN_pmc_summary.json

Values are reported raw (sum / average per launch / max over launches); unit conversion (e.g. FETCH_SIZE, WRITE_SIZE in
KiB on gfx950, MI355X_MICROARCH.md) is left to the reader and stated next to the numbers in DESIGN.md."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)            # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    return name.strip()


def main():
    acc = {}
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k, c, v = short(row["Kernel_Name"]), row["Counter_Name"], float(row["Counter_Value"])
                    a = acc.setdefault(c, {}).setdefault(k, [0, 0.0, 0.0])
                    a[0] += 1
                    a[1] += v
                    a[2] = max(a[2], v)
    out = {c: {k: dict(launches=a[0], sum=round(a[1], 1), avg_per_launch=round(a[1] / a[0], 2), max=round(a[2], 1))
               for k, a in sorted(ks.items(), key=lambda kv: -kv[1][1])} for c, ks in acc.items()}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
