#!/bin/bash
# isolated kernel durations: development build, one stream (ISDQN_SINGLE_STREAM), eager launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
lib=is-dqn_amd/lib/libisdqn_hip_$v.so
out=gpurun_out/kts_$v; rm -rf $out; mkdir -p $out
ISDQN_SINGLE_STREAM=1 ISDQN_HIP_LIB=$PWD/$lib timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --workload c2 --no-cpu-baseline --steps 400 --warmup 200 --settle 0 --replay-stats 0 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep '"metric"' $out.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v single stream: %.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"
python3 scripts/kernel_means.py $out | grep -E "conv|gemm|head|adam"
rm -rf $out
done
