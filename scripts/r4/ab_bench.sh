#!/bin/bash
# bench-only A/B of two builds (alternating runs, no traces).  usage: ab_bench.sh <variantA or ""> <variantB or ""> [workload] [rounds] [extra bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
wl=${3:-c2}; n=${4:-2}; extra=${5:-}
lib() { echo "$PWD/is-dqn_amd/lib/libisdqn_hip${1:+_$1}.so"; }
for r in $(seq $n); do for tag in A B; do
  if [ $tag = A ]; then l=$(lib "$1"); else l=$(lib "$2"); fi
  v=$(ISDQN_HIP_LIB=$l timeout -k 10 240 python bench.py --workload $wl --no-cpu-baseline --steps 2400 --warmup 800 --replay-stats 0 $extra 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "round $r $tag ($(basename $l)) $wl $extra: $v"
done; done
