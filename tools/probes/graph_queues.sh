# usage (GPU box): bash tools/probes/graph_queues.sh  - hipGraph replay of the config-1 step by number of graph queues
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o "\"value\": [0-9.]*"; }
echo "== eager"; run ITG_GRAPH=0
for q in 1 2 3 4; do echo "== graph, DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; run ITG_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; done
echo "== graph, default queues"; run ITG_GRAPH=1
echo "== graph, queues 4, GPU_MAX_HW_QUEUES=3"; run ITG_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 GPU_MAX_HW_QUEUES=3
echo "== graph, queues 3, GPU_MAX_HW_QUEUES=3"; run ITG_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=3 GPU_MAX_HW_QUEUES=3
