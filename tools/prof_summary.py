#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the committed summaries under profiles/.

  python tools/prof_summary.py --stats DIR --iters N [--fetch DIR --write DIR] --round r01 --cmd "..."

--stats : directory of `rocprofv3 --kernel-trace --stats --output-format csv` (kernel_stats.csv)
--fetch / --write : directories of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
                    (counter_collection.csv); FETCH_SIZE is doubled and both are KiB -> bytes, as
                    MI355X_MICROARCH.md prescribes for gfx950.
Writes profiles/<round>_bench_kernel_stats.md (+ .csv copy) and profiles/<round>_hbm_traffic.json
(kernel -> average HBM bytes per launch), which bench.py reads for roofline.traffic."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("itgk::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return max(hits, key=os.path.getmtime)       # a re-used tag keeps older runs beside the new one: take the newest


def pmc(d, counter):
    rows = csv.DictReader(open(find(d, "*counter_collection.csv")))
    tot, cnt = collections.Counter(), collections.Counter()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--iters", type=float, required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--round", default="r01")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--note", default="")
    ap.add_argument("--suffix", default="", help="e.g. _config3: files profiles/<round><suffix>_bench_kernel_stats.md / _hbm_traffic.json")
    a = ap.parse_args()
    stats = find(a.stats, "*kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats)) if "stream_spin_kernel" not in r["Name"]]   # start-up stream placement probe
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = os.path.join(ROOT, "profiles", "%s%s_bench_kernel_stats" % (a.round, a.suffix))
    shutil.copy(stats, out + ".csv")
    with open(out + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s  (MI355X)\n" % a.cmd)
        if a.note:
            f.write(a.note + "\n")
        f.write("total kernel time %.2f ms per iteration over %g iterations, %.0f launches per iteration\n\n" % (
            tot / 1e6 / a.iters, a.iters, sum(int(r["Calls"]) for r in rows) / a.iters))
        f.write("| kernel | launches/iter | ms/iter | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            if float(r["TotalDurationNs"]) / tot < 0.001:
                continue
            f.write("| `%s` | %.1f | %.3f | %.1f | %.1f |\n" % (
                short(r["Name"])[:70], int(r["Calls"]) / a.iters, float(r["TotalDurationNs"]) / 1e6 / a.iters,
                float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
        if a.fetch and a.write:
            fe, wr = pmc(a.fetch, "FETCH_SIZE"), pmc(a.write, "WRITE_SIZE")
            avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in rows}       # the --stats pass: the kernel ALONE (un-overlapped step)
            traffic = {}
            f.write("\n## HBM traffic per launch (separate --pmc passes; FETCH_SIZE x2 gfx950 correction, KiB -> bytes)\n\n")
            f.write("| kernel | launches | avg fetch MB | avg write MB |\n|---|---|---|---|\n")
            for k in sorted(fe, key=lambda k: -fe[k][0] * fe[k][1]):
                fb = fe[k][0] * 2 * 1024
                wb = wr.get(k, (0, 0))[0] * 1024
                traffic[k] = {"fetch_bytes": fb, "write_bytes": wb, "launches": fe[k][1]}
                if k in avg_ns:          # HBM-side rate of the kernel: bytes the memory moved / its average duration
                    traffic[k]["avg_us"] = round(avg_ns[k] / 1e3, 2)
                    traffic[k]["hbm_gbps"] = round((fb + wb) / avg_ns[k], 1)
                if fb * fe[k][1] > 50e6:
                    f.write("| `%s` | %d | %.1f | %.1f |\n" % (k[:70], fe[k][1], fb / 1e6, wb / 1e6))
            import subprocess, sys
            sys.path.insert(0, ROOT)
            import bench
            try:
                head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
            except Exception:      # noqa: BLE001  (the GPU box has no .git)
                head = os.environ.get("ITG_GIT_HEAD", "unknown")
            traffic["_meta"] = {"kernel_source_sha16": bench.kernel_source_hash(), "schedule_source_sha16": bench.schedule_source_hash(),
                                "git_head": head}
            json.dump(traffic, open(os.path.join(ROOT, "profiles", "%s%s_hbm_traffic.json" % (a.round, a.suffix)), "w"), indent=0)
    print("wrote", out + ".md")


if __name__ == "__main__":
    main()
