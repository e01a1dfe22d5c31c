#!/bin/bash
# runs graph_branch_start.py for several chain lengths under rocprofv3 --kernel-trace and prints, per replay, when each queue's first kernel started
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export SPIN_A=40 SPIN_B=60
for cfg in "30 30 afirst" "30 30" "30 30 split" "30 30 split picked"; do
  rm -rf /tmp/gbs; rocprofv3 --kernel-trace --output-format csv -d /tmp/gbs -- python3 $ROOT/tools/probes/graph_branch_start.py $cfg > /tmp/gbs.log 2>&1
  python3 - "$cfg" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/gbs/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'spin' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
na, nb = map(int, sys.argv[1].split()[:2])
per = na + nb
rows = rows[2:]            # the two warm-up spins
out = []
for k in range(len(rows) // per):
    seg = rows[k * per:(k + 1) * per]
    t0 = int(seg[0]['Start_Timestamp'])
    first, cnt, dur = {}, collections.Counter(), collections.Counter()
    for r in seg:
        first.setdefault(r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3)
        cnt[r['Queue_Id']] += 1
        dur[r['Queue_Id']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    end = (max(int(r['End_Timestamp']) for r in seg) - t0) / 1e3
    out.append("replay %d: first kernel per queue %s, kernels per queue %s, avg us %s, all done at %.0f us" % (
        k, {q: round(v, 1) for q, v in first.items()}, dict(cnt), {q: round(dur[q] / cnt[q], 1) for q in cnt}, end))
print("cfg", sys.argv[1]); print("\n".join(out[-2:]))
PY
done > $OUT/r6_graph_branch_start.txt 2>&1
cat $OUT/r6_graph_branch_start.txt
