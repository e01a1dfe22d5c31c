#!/usr/bin/env python3
"""In-situ kernel durations of one traced step (tools/step_trace.py --rows output) against the same kernels' averages when they
run ALONE (a rocprofv3 --stats CSV of the un-overlapped bench): which kernels pay for sharing the chip.
usage: python tools/situ_excess.py <step_trace.txt> <kernel_stats.csv> [rows]"""
import collections
import csv
import re
import sys


def norm(n):
    n = re.sub(r'^void ', '', n).replace('itgk::', '').replace('(anonymous namespace)::', '')
    depth, out = 0, []
    for ch in n:
        depth += ch == '<'
        depth -= ch == '>'
        if ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()[:60]


def main():
    alone = {norm(r['Name']): float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}
    situ = collections.defaultdict(list)
    for l in open(sys.argv[1]):
        m = re.match(r'q(\d)\s+([\d.]+) us\s+dur\s+([\d.]+)\s+gap\s+([\d.]+)\s+(\S.*)$', l)
        if m:
            situ[norm(m.group(5))].append(float(m.group(3)))
    rows = []
    for n, v in situ.items():
        a, tot = alone.get(n), sum(v)
        rows.append(((tot - a * len(v)) if a else 0.0, n, len(v), tot, a))
    rows.sort(reverse=True)
    print("%-62s %4s %9s %9s %9s" % ("kernel", "n", "in situ", "alone", "excess us"))
    for ex, n, c, tot, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
        print("%-62s %4d %9.1f %9.1f %9.1f" % (n, c, tot, a * c if a else -1, ex))
    print("total in situ %.1f us, excess %.1f us; not in the stats file: %s" % (
        sum(r[3] for r in rows), sum(r[0] for r in rows), [r[1] for r in rows if r[4] is None]))


if __name__ == "__main__":
    main()
