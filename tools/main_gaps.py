#!/usr/bin/env python3
"""Where does the default stream wait?  From a rocprofv3 --kernel-trace CSV of an overlapped bench run: per stream the
busy time of one steady-state step, and the default stream's idle gaps > N us with the kernels before / after them.
usage: python tools/main_gaps.py <dir with *_kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("itgk::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n).replace("at::native::", "")[:56]


d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Stream_Id"]), int(r["Queue_Id"]))
        for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
a, b = adam[-5], adam[-3]                      # one whole step, well after warm-up
seg = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
print("step wall %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(seg)))
per = collections.defaultdict(list)
for r in seg:
    per[(r[3], r[4])].append(r)
main_key = max(per, key=lambda k: len(per[k]))
for k, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
    print("stream %d queue %d: %4d launches, busy %.3f ms%s" % (k[0], k[1], len(v), sum(e - s for s, e, *_ in v) / 1e6,
                                                               "  <- default stream" if k == main_key else ""))
prev = None
tot = 0.0
for r in per[main_key]:
    if prev is not None and (r[0] - prev[1]) / 1e3 >= min_gap:
        g = (r[0] - prev[1]) / 1e3
        tot += g
        others = [o for o in seg if (o[3], o[4]) != main_key and o[0] < r[0] and o[1] > prev[1]]
        print("%8.1f us into the step: idle %6.1f us  after %-40s before %-40s (other streams meanwhile: %s)" % (
            (prev[1] - t0) / 1e3, g, prev[2][:40], r[2][:40], ", ".join(sorted({o[2][:28] for o in others})[:3]) or "-"))
    prev = r
print("default stream idle in gaps >= %.0f us: %.3f ms" % (min_gap, tot / 1e3))
