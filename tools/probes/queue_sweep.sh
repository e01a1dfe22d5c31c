# usage (GPU box): bash tools/probes/queue_sweep.sh   - config-1 bench under a one-rank RCCL group, by stream count
run() { env "$@" RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o "\"value\": [0-9.]*"; }
for w in 2 1; do
 echo "== plain, $w weight-gradient streams"; run ITG_WGRAD_STREAMS=$w
 echo "== RCCL initialised only, $w W streams"; run ITG_WGRAD_STREAMS=$w ITG_FORCE_COLLECTIVES=1 ITG_RT_INIT_ONLY=1
 echo "== RCCL, one all-reduce per model, $w W streams"; run ITG_WGRAD_STREAMS=$w ITG_FORCE_COLLECTIVES=1 ITG_BUCKETS=0
 echo "== RCCL, buckets, $w W streams"; run ITG_WGRAD_STREAMS=$w ITG_FORCE_COLLECTIVES=1 ITG_BUCKETS=1
done
echo "== RCCL initialised only, no overlap at all"; run ITG_OVERLAP=0 ITG_FORCE_COLLECTIVES=1 ITG_RT_INIT_ONLY=1
