# usage (GPU box): bash tools/probes/cu_mask_sweep.sh - config-1 bench with CU-masked background streams
# needs the ITG_SIDE_MASK / ITG_W_MASK hooks of the experiment ("tried and dropped"; not in the tree)
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o "\"value\": [0-9.]*"; }
echo "== no masks"; run ITG_X=0
for m in 7/8 3/4 1/2; do echo "== side $m"; run ITG_SIDE_MASK=$m; done
for m in 7/8 3/4 1/2; do echo "== W $m"; run ITG_W_MASK=$m; done
for m in 7/8 3/4 1/2; do echo "== side + W $m"; run ITG_SIDE_MASK=$m ITG_W_MASK=$m; done
echo "== no masks"; run ITG_X=0
