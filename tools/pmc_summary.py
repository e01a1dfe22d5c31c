#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_run.sh into profiles/<round>_pmc_summary.md: per kernel
(top N by time) the MFMA-pipe busy share, issue / wait shares, occupancy proxy, VALU : MFMA instruction mix and
LDS bank-conflict share.

  python tools/pmc_summary.py gpurun_out/<tag> --round r02 [--top 12]

Definitions (gfx950; SQ counters are sums over the chip, SQ_*_CYCLES in quad-cycles as MI355X_MICROARCH.md notes):
  mfma busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = duration x clock, where
                 clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (the guide's effective-clock quotient)
                 [= rocprofv3's MfmaUtil expression with SIMD_NUM = 1024]
  wait % / issue-stall % / active %  = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  waves/SIMD   = SQ_WAVE_CYCLES x 4 / (kernel cycles x 1024)      (average resident waves per SIMD)
  VALU : MFMA  = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs)
  LDS conflict % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import argparse
import collections
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("itgk::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)


def load(d):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run of the tag
    per = collections.defaultdict(dict)             # dispatch -> counter -> value (+ name, duration)
    for r in csv.DictReader(open(f)):
        e = per[int(r["Dispatch_Id"])]
        e["name"] = short(r["Kernel_Name"])
        e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.Counter())
    for e in per.values():
        a = agg[e["name"]]
        a["launches"] += 1
        for k, v in e.items():
            if k != "name":
                a[k] += v
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix", help="gpurun_out/<tag>: directories <tag>_A and <tag>_B")
    ap.add_argument("--round", default="r02")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--steps", type=float, default=4.0, help="train steps the profiled program ran")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    A, B = load(a.prefix + "_A"), load(a.prefix + "_B")
    # stream_spin_kernel: the start-up stream placement probe (ops.concurrent_streams), not part of a step
    names = [k for k in sorted(A, key=lambda k: -A[k]["ns"]) if "stream_spin" not in k][:a.top]
    out = os.path.join(ROOT, "profiles", "%s_pmc_summary.md" % a.round)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc passes over tools/step_profile.py (one un-overlapped train step x %g, MI355X)\n" % a.steps)
        f.write("Two counter passes (tools/pmc_run.sh), each with --kernel-trace only beside --pmc.  Profiled dispatches are "
                "serialised and run at a lower clock than un-profiled ones: read shares, not durations.\n")
        if a.note:
            f.write(a.note + "\n")
        f.write("\n| kernel | launches/step | avg us (profiled) | clock GHz | MFMA busy % | active % | issue-stall % | wait % | "
                "waves/SIMD | VALU:MFMA | LDS conflict % |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in names:
            x, y = A[k], B.get(k, collections.Counter())
            ns = x["ns"]
            cyc = x["GRBM_GUI_ACTIVE"] / 8.0                     # chip cycles summed over the kernel's launches
            clock = cyc / ns if ns else 0.0
            mfma = 100.0 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0.0
            wc = x["SQ_WAVE_CYCLES"] or 1.0
            f.write("| `%s` | %.1f | %.1f | %.2f | %.1f | %.1f | %.1f | %.1f | %.2f | %s | %.1f |\n" % (
                k[:64], x["launches"] / a.steps, ns / x["launches"] / 1e3, clock, mfma,
                100 * x["SQ_ACTIVE_INST_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc, 100 * x["SQ_WAIT_ANY"] / wc,
                wc * 4 / (cyc * 1024) if cyc else 0.0,
                ("%.1f" % (y["SQ_INSTS_VALU"] / y["SQ_INSTS_MFMA"])) if y.get("SQ_INSTS_MFMA") else
                ("%.1f" % (y["SQ_INSTS_VALU"] / x["SQ_INSTS_MFMA"]) if x.get("SQ_INSTS_MFMA") else "-"),
                100.0 * y["SQ_LDS_BANK_CONFLICT"] / y["SQ_LDS_IDX_ACTIVE"] if y.get("SQ_LDS_IDX_ACTIVE") else 0.0))
    print(open(out).read())


if __name__ == "__main__":
    main()
