#!/bin/bash
# duration of the weights-stationary forward kernel with pieces ablated (development build; results are wrong by construction)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so
for ab in ${@:-0 1 2 4 8 3 10 11 15}; do
  out=gpurun_out/tl_ab$ab; mkdir -p $out
  ISDQN_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload c2 --no-cpu-baseline --steps 400 --warmup 200 --settle 0 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "== ISDQN_ABLATE=$ab (1 no weight fill, 2 no fragment loads, 4 no stores, 8 no MFMA / LDS reads)"
  python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) | grep -E "period|conv_fwd_ws"
  rm -rf $out/kt
done
