#!/bin/bash
# Interleaved N-way A/B of bench.py on ONE box: tools/probes/multi_ab.sh ROUNDS STEPS "<env 1>" "<env 2>" ... [-- bench args]
# (an empty string is the default environment).  Prints every run and the median crops/s per variant.
R=$1; S=$2; shift 2
V=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do V+=("$1"); shift; done
[ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=/tmp/mab_$$.txt; : > $T
for i in $(seq $R); do
  for k in "${!V[@]}"; do
    v=$(env ${V[$k]} python3 $ROOT/bench.py --steps $S --warmup 5 --no-cpu-baseline --no-direct --no-membound "$@" 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
    echo "$k $v" | tee -a $T
  done
done
python3 - $T "${V[@]}" <<'PY'
import sys, statistics
rows = [l.split() for l in open(sys.argv[1])]
names = sys.argv[2:]
base = None
for k, n in enumerate(names):
    vals = [float(v) for i, v in rows if int(i) == k]
    m = statistics.median(vals)
    base = base or m
    print("median [%s] %.1f  (%.4f of the first)  runs %s" % (n or "default", m, m / base, " ".join("%.0f" % v for v in vals)))
PY
