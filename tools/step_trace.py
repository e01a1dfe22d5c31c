#!/usr/bin/env python3
"""Launch-by-launch view of ONE train step from a rocprofv3 --kernel-trace CSV (overlapped or not): per launch the queue,
start relative to the step's first kernel, duration and the gap to the previous kernel of the same queue; then, per phase
of the main queue (G forward, D(fake) forward, D backward, G-step D forward, backward through D and G), the sum of kernel
time, of gaps, and the launches.  Answers "is the generator path kernel-bound or latency-bound" with numbers.
usage: python tools/step_trace.py <dir with *_kernel_trace.csv> [step index=3] [--rows]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"itgk::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    d = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 3
    show = "--rows" in sys.argv
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"])) for r in csv.DictReader(open(f))]
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    ends = adam[1::2]
    if step + 1 >= len(ends):
        step = len(ends) - 2
    lo, hi = ends[step] + 1, ends[step + 1]
    # the pack launch after Adam(G) belongs to the previous step
    while lo < hi and rows[lo][2].startswith("pack_multi"):
        lo += 1
    seg = rows[lo:hi + 1]
    t0 = seg[0][0]
    by_q = {}
    for r in seg:
        by_q.setdefault(r[3], []).append(r)
    main_q = max(by_q, key=lambda q: len(by_q[q]))
    print("step %d: %d launches on %d queues, wall %.3f ms; main queue %d has %d launches" % (
        step, len(seg), len(by_q), (seg[-1][1] - t0) / 1e6, main_q, len(by_q[main_q])))
    last_end = {}
    out = []
    for (a, b, n, q) in seg:
        gap = (a - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = b
        out.append((q, (a - t0) / 1e3, (b - a) / 1e3, gap, n))
    if show:
        for q, st, du, gap, n in out:
            print("q%-2d %9.1f us  dur %7.1f  gap %6.1f  %s%s" % (q, st, du, gap, "" if q == main_q else "      ", n))
    # phases of the main queue, split at the loss kernels (bce_fwd: D(fake) fwd end, ...) and adam
    mq = [(st, du, gap, n) for q, st, du, gap, n in out if q == main_q]
    marks = [i for i, r in enumerate(mq) if r[3].startswith(("bce_fwd", "hinge_fwd", "adam_kernel"))]
    prev = 0
    for k, i in enumerate(marks + [len(mq) - 1]):
        part = mq[prev:i + 1]
        if not part:
            continue
        ksum, gsum = sum(p[1] for p in part), sum(p[2] for p in part[1:])
        print("main-queue phase %d (ends with %-12s): %3d launches, kernel %.3f ms, gaps %.3f ms (max %.1f us), span %.3f ms" % (
            k, part[-1][3][:12], len(part), ksum / 1e3, gsum / 1e3, max([p[2] for p in part[1:]] or [0]),
            (part[-1][0] + part[-1][1] - part[0][0]) / 1e3))
        prev = i + 1
    for q, lst in sorted(by_q.items()):
        print("queue %d: %d launches, kernel time %.3f ms" % (q, len(lst), sum(b - a for a, b, _, _ in lst) / 1e6))


if __name__ == "__main__":
    main()
