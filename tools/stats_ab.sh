#!/bin/bash
# Per-kernel time of one un-overlapped bench under two environments, side by side (same box):
#   tools/stats_ab.sh <tag> "<env A>" "<env B>"   -> gpurun_out/<tag>_{A,B}_stats + a merged table on stdout
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ITG_OVERLAP=0
for V in A B; do
  if [ $V = A ]; then E="$2"; else E="$3"; fi
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${V}_stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_${V}_stats.log 2>&1
done
python3 - "$OUT/${TAG}_A_stats" "$OUT/${TAG}_B_stats" <<'PY'
import csv, glob, sys, re, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    t = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Name"].replace("(anonymous namespace)::", "").replace("itgk::", "").replace("void ", ""))
        t[n] = (int(r["Calls"]) / 8.0, float(r["TotalDurationNs"]) / 8e3)
    return t
a, b = load(sys.argv[1]), load(sys.argv[2])
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))
print("%-72s %6s %9s | %6s %9s" % ("kernel (per iteration)", "nA", "usA", "nB", "usB"))
for k in keys[:70]:
    print("%-72s %6.1f %9.1f | %6.1f %9.1f" % (k[:72], *a.get(k, (0, 0)), *b.get(k, (0, 0))))
print("total us: A %.0f  B %.0f   launches: A %.0f  B %.0f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values()),
      sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
PY
