#!/bin/bash
# second sweep of HIP runtime knobs against the c2 step (product build): fence scopes, graph queues, stream-op waits
cd "$GRAFT_REPO_ROOT"
run() { v=$(env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3200 --warmup 3200 --replay-stats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step']*1e3,1))" 2>/dev/null); echo "[$*]: ${v:-FAILED}"; }
run X=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
run GPU_STREAMOPS_CP_WAIT=0
run GPU_STREAMOPS_CP_WAIT=1
run GPU_FLUSH_ON_EXECUTION=1
run DEBUG_HIP_FORCE_ASYNC_QUEUE=1
run GPU_NUM_MEM_DEPENDENCY=0
run AMD_DIRECT_DISPATCH=0
run X=1
