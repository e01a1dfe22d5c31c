#!/bin/bash
# pixel-tile width of the 49-class Winograd GEMMs (conv_nt_kernel<128, X, ...>): per-call times and the train step
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for V in 0 64 96 128; do
  echo "== ITG_WINO_BPIX=$V"
  ITG_WINO_BPIX=$V python3 $ROOT/tools/step_profile.py config1 2>/dev/null | awk '$1==3||$1==7||$1==36||$1==40||$1==50||$1==53 {print}'
  ITG_WINO_BPIX=$V python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])"
done
