#!/bin/bash
# A/B of two builds: alternating bench runs, then kernel traces of both and a per-kernel comparison
# usage: scripts/r2/ab.sh <variantA or ""> <variantB> [workload]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
a=is-dqn_amd/lib/libisdqn_hip${1:+_$1}.so; b=is-dqn_amd/lib/libisdqn_hip${2:+_$2}.so; wl=${3:-c2}
for r in 1 2; do for tag in A B; do
  if [ $tag = A ]; then lib=$a; else lib=$b; fi
  v=$(ISDQN_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps 2400 --warmup 800 --replay-stats 0 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "round $r $tag ($lib): $v"
done; done
for tag in A B; do
  if [ $tag = A ]; then lib=$a; else lib=$b; fi
  ISDQN_HIP_LIB=$PWD/$lib timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$tag -- python3 bench.py --workload $wl --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > gpurun_out/ab_$tag.log 2>&1 || exit 1
done
python scripts/kernel_means.py gpurun_out/ab_A gpurun_out/ab_B
rm -rf gpurun_out/ab_A gpurun_out/ab_B
