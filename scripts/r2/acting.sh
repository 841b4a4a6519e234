#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_agent.py -x -q 2>&1 | tail -5 &&
timeout -k 10 300 python scripts/r2/acting_bench.py > gpurun_out/acting_bench.json 2> gpurun_out/acting_bench.err; tail -3 gpurun_out/acting_bench.err; cat gpurun_out/acting_bench.json
