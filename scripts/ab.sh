#!/bin/bash
# A/B two builds of libisdqn_hip.so inside one GPU box: kernel-trace both under identical conditions.
# usage: scripts/ab.sh <libA> <libB>   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in A B; do
  if [ $tag = A ]; then lib=$1; else lib=$2; fi
  export ISDQN_HIP_LIB=$PWD/$lib
  timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$tag -- python bench.py --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/ab_$tag.log 2>&1 || exit 1
done
python scripts/kernel_means.py gpurun_out/ab_A gpurun_out/ab_B
