#!/usr/bin/env python3
"""Timeline view of rocprofv3 --kernel-trace CSV of a bench run (stream overlap on): per train step the wall time, the
union of kernel intervals (device busy), the idle gaps, the sum of kernel durations (> busy when streams overlap), and
the launches.  usage: python tools/timeline.py <dir with *_kernel_trace.csv> [steps_to_show]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Stream_Id", 0) or 0),
             int(r["Queue_Id"])) for r in csv.DictReader(open(f))]
    rows.sort()
    # step boundaries: the Adam kernel runs twice per step (D then G); a step ends with the second one
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    bounds = adam[1::2]
    print("kernels %d, adam launches %d -> %d steps" % (len(rows), len(adam), len(bounds)))
    prev_end = rows[bounds[0]][1] if bounds else rows[0][0]
    for k in range(1, len(bounds)):
        seg = [r for r in rows if prev_end <= r[0] and r[1] <= rows[bounds[k]][1] + 1]
        if not seg:
            continue
        t0, t1 = prev_end, rows[bounds[k]][1]
        ivs = sorted((a, b) for a, b, *_ in seg)
        busy, cur_a, cur_b = 0, ivs[0][0], ivs[0][1]
        gaps = []
        for a, b in ivs[1:]:
            if a > cur_b:
                busy += cur_b - cur_a
                gaps.append(a - cur_b)
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        busy += cur_b - cur_a
        tot = sum(b - a for a, b in ivs)
        queues = len(set(r[4] for r in seg))
        print("step %2d: wall %.3f ms | device busy %.3f ms | idle %.3f ms in %d gaps (max %.1f us) | sum of kernel time %.3f ms | %d launches on %d queues" % (
            k, (t1 - t0) / 1e6, busy / 1e6, ((t1 - t0) - busy) / 1e6, len(gaps), max(gaps) / 1e3 if gaps else 0, tot / 1e6, len(seg), queues))
        prev_end = t1


if __name__ == "__main__":
    main()
