#!/bin/bash
# kernel means of one build: scripts/r3/kt.sh <variant or ""> [grep pattern]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
lib=is-dqn_amd/lib/libisdqn_hip${1:+_$1}.so
out=gpurun_out/kt_${1:-prod}; rm -rf $out; mkdir -p $out
ISDQN_HIP_LIB=$PWD/$lib timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --workload c2 --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep '"metric"' $out.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${1:-prod}: %.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"
python3 scripts/kernel_means.py $out | grep -E "${2:-conv_fwd}"
rm -rf $out
