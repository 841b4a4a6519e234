#!/bin/bash
# runtime environment knobs of the HIP runtime vs the c2 step (product build)
cd "$GRAFT_REPO_ROOT"
run() { v=$(env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3200 --warmup 3200 --replay-stats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'])"); echo "[$*]: $v"; }
run X=0
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=1
run ROC_ACTIVE_WAIT_TIMEOUT=0
run AMD_SERIALIZE_KERNEL=0 HSA_ENABLE_INTERRUPT=0
run X=1
