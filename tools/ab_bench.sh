#!/bin/bash
# Alternating A/B of bench.py on ONE box: tools/ab_bench.sh "<env A>" "<env B>" [rounds] [steps] [extra bench args]
# Prints every run and the median crops/s of each side (boxes differ by up to 10 %, runs on one box by ~0.5 %).
A="$1"; B="$2"; R=${3:-4}; S=${4:-60}; shift 4 2>/dev/null
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq $R); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    v=$(env $E python3 $ROOT/bench.py --steps $S --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
    echo "$side $v"
  done
done | tee /tmp/ab_$$.txt
python3 - /tmp/ab_$$.txt "$A" "$B" <<'PY'
import sys, statistics
a = [float(l.split()[1]) for l in open(sys.argv[1]) if l.startswith("A ")]
b = [float(l.split()[1]) for l in open(sys.argv[1]) if l.startswith("B ")]
print("median A [%s] %.1f   B [%s] %.1f   B/A %.4f" % (sys.argv[2], statistics.median(a), sys.argv[3], statistics.median(b), statistics.median(b) / statistics.median(a)))
PY
