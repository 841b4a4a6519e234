#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== driver-style run"; timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids || exit 1
echo "== default run"; timeout -k 10 900 python bench.py 2>&1 | grep -v amdgpu.ids || exit 1
echo "== c3"; timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids || exit 1
echo "== c5"; timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 2000 --warmup 2000 2>&1 | grep -v amdgpu.ids || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
