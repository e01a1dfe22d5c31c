#!/usr/bin/env python3
"""Per-kernel launch count and average duration from a rocprofv3 --kernel-trace --stats CSV directory.
usage: python tools/kstats.py <dir> [name filter]"""
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void |\(anonymous namespace\)::|itgk::", "", r["Name"])
        name = re.sub(r"\(.*$", "", name)[:70]
        if flt and flt not in name:
            continue
        print("%-72s calls %5s  avg %9.1f us  total %9.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))


if __name__ == "__main__":
    main()
